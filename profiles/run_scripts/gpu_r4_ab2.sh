#!/bin/bash
# A/B of env switches on the bench step incl. the in-step gather roofline: gpu_r4_ab2.sh <tag> "ENV.." ...   ("-" = defaults)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=$1; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
B="--no-cpu-baseline --no-pmc --no-parity ${BENCH_ARGS:---steps 20 --warmup 5}"
i=0
for E in "$@"; do
  i=$((i+1)); [ "$E" = "-" ] && E=""
  env $E timeout 300 python bench.py $B > $OUT/bench_$i.json 2>> $OUT/bench.err
  python - $OUT/bench_$i.json "${E:-defaults}" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r, k, t = d.get("roofline", {}), d.get("roofline_gather_kernel", {}), d.get("roofline_tower", {})
    print("%-40s %.4f ms/step %s | gather in step %.2f us = %.3f, alone %.2f us = %.3f | tower %.1f us" % (sys.argv[2], d["ms_per_step"], d.get("repeats_ms_per_step"), r.get("avg_launch_us", 0), r.get("frac", 0), k.get("avg_launch_us", 0), k.get("frac", 0), t.get("avg_launch_us", 0)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
tail -n 3 $OUT/bench.err
