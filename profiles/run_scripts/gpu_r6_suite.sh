#!/bin/bash
# round 6: the whole GPU suite + smoke, then the gather's in-step duration two ways (clock stamps dumped by bench.py, rocprofv3 kernel trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r6suite}; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -n 4 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --dump-stamps $OUT/gather_instep_stamps.txt > $OUT/bench_c2_driver_args.json 2> $OUT/bench.err
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o trace -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/gather_instep_from_trace.py $T $OUT/gather_instep_rocprof.json
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/c2_step_timeline.txt
rm -rf $OUT/prof
python - $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_c2_driver_args.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("repeats_ms_per_step"), {k: d["roofline"].get(k) for k in ("frac", "avg_launch_us", "rocprof_instep_us", "frac_rocprof_instep", "traffic")})
PY
head -5 $OUT/gather_instep_stamps.txt; tail -n 3 $OUT/bench.err
