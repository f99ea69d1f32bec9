#!/bin/bash
# A/B of env switches on the bench step, same box: usage gpu_r4_ab.sh <tag> "ENV1=.. ENV2=.." "ENV..." ...   ("-" = defaults)
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
    print("%-44s %.4f ms/step  %.1f M ex/s  %s tower %s us" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d.get("repeats_ms_per_step"), d.get("roofline_tower", {}).get("avg_launch_us")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
tail -n 3 $OUT/bench.err
