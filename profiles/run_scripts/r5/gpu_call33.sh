#!/bin/bash
# round 5 call 33: configs[3] bench line again (roofline of the input layer by every column's own width / kernel names as launched)
O=gpurun_out/r5_call33; mkdir -p $O
timeout 400 python bench.py --config c4 --no-cpu-baseline --no-pmc --steps 60 > $O/bench_c4.json 2> $O/bench_c4.err || tail -5 $O/bench_c4.err
python - $O/bench_c4.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], json.dumps(d["roofline"])[:700])
PY
