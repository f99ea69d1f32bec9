#!/bin/bash
# round 4, call 3: 8-wavefront tower with its split scratch in dead LDS regions; write-through vs plain stores, alone and in the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_prefetch.py -q -m gpu -x > $OUT/pytest_chain.txt 2>&1; tail -n 2 $OUT/pytest_chain.txt
show() { grep -v "^tile stamps\|amdgpu.ids\|wave0 F0\|row tile" $1 | tail -n 3; }
for W in 4 8; do for F in 0 4; do
  echo "== waves $W flags $F"; WD_CHAIN_FLAGS=$F WD_CHAIN_WAVES=$W timeout 200 python scripts/bench_chain.py > $OUT/chain_w${W}_f$F.txt 2>&1; show $OUT/chain_w${W}_f$F.txt
done; done
B="--no-cpu-baseline --no-pmc --no-parity"
for W in 4 8; do for WT in 1 0; do
  WD_WT=$WT WD_CHAIN_WAVES=$W timeout 300 python bench.py --steps 20 --warmup 5 $B > $OUT/bench_w${W}_wt$WT.json 2>> $OUT/bench.err
  python - $OUT/bench_w${W}_wt$WT.json w${W}_wt$WT <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-10s %.4f ms/step  %.1f M ex/s  %s tower %s us" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d.get("repeats_ms_per_step"), d.get("roofline_tower", {}).get("avg_launch_us")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done; done
