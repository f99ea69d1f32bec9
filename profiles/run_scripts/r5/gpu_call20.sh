#!/bin/bash
# round 5, call 19: the gather of batch t+1 BEFORE the sort beside tower(t) (it lands on the x tile / F0 instead of the head): stage cycles + step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call20; mkdir -p $OUT
for v in "WD_PIPE_SIDE=gather" "WD_PIPE_SIDE=bucket"; do
  env $v python scripts/sclk_probe.py 2>&1 | grep "in the step" | sed "s/^/[$v] /" | cut -c1-250
done | tee $OUT/stages.txt
B="--no-cpu-baseline --no-pmc --no-parity --steps 20 --warmup 5 --repeats 9"
for v in "WD_PIPE_SIDE=gather" "WD_PIPE_SIDE=bucket" "WD_PIPE_SIDE=gather" "WD_PIPE_SIDE=bucket" "WD_PIPE_SIDE=sort"; do
  env $v timeout 200 python bench.py $B > $OUT/b.json 2>> $OUT/bench.err
  python - "$v" $OUT/b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("[%s] %.4f ms/step %s roofline %s us" % (sys.argv[1], d["ms_per_step"], d.get("ms_per_step_min_median_max"), (d.get("roofline") or {}).get("avg_launch_us")))
PY
done | tee $OUT/step_ab.txt
