#!/bin/bash
# the sharded step on a one-rank RCCL group: gpu_r4_shard_bench.sh <tag> "ENV.." ...  ("-" = defaults); runs uniform and zipf
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=$1; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
B="--no-cpu-baseline --no-pmc --no-parity --steps 20 --warmup 5"
P=29600
for E in "$@"; do
  [ "$E" = "-" ] && E=""
  for D in uniform zipf; do
    P=$((P+1))
    env $E MASTER_PORT=$P timeout 300 python bench.py $B --force-sharded --dist $D > $OUT/sh_$P.json 2>> $OUT/sh.err
    python - $OUT/sh_$P.json "$D ${E:-defaults}" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("sharded one rank %-40s %.4f ms/step %s" % (sys.argv[2], d["ms_per_step"], d["repeats_ms_per_step"]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
  done
done
tail -n 2 $OUT/sh.err
