#!/bin/bash
# the tower alone (stage cycles) under different environments: gpu_r4_chain_env.sh <tag> "ENV.." ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=$1; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; i=0
for E in "$@"; do i=$((i+1)); [ "$E" = "-" ] && E=""
  env $E timeout 200 python scripts/bench_chain.py > $OUT/chain_$i.txt 2>&1
  echo "== ${E:-defaults}: $(grep '^chain B' $OUT/chain_$i.txt | sed 's/.*list) //;s/(forward.*//')"; grep "^workgroup 0" $OUT/chain_$i.txt
done
