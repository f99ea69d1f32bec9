#!/bin/bash
# round 5 call 39: configs[3] bench line with its PMC child passes (the default flags), JSON validity
O=gpurun_out/r5_call39; mkdir -p $O
timeout 500 python bench.py --config c4 --no-cpu-baseline > $O/bench_c4_pmc.json 2> $O/bench_c4_pmc.err || tail -5 $O/bench_c4_pmc.err
python - $O/bench_c4_pmc.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["ms_per_step_min_median_max"])
for k in ("roofline", "roofline_gather_kernel", "roofline_tower"):
    r = d.get(k) or {}
    print(k, r.get("kernel", "")[:90], r.get("avg_launch_us"), r.get("frac"), r.get("traffic"))
PY
