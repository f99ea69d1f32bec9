#!/bin/bash
# round 4, call 1: the two-wavefronts-per-SIMD tower (mlp_chain8.hip) -- parity tests, then A/B against the round-3 kernel
# (WD_CHAIN_WAVES=4) on the same box: kernel alone with stage cycles, and the bench step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_prefetch.py tests/test_gpu_step.py -q -m gpu -x > $OUT/pytest_chain.txt 2>&1; tail -n 5 $OUT/pytest_chain.txt
for W in 4 8; do
  echo "== WD_CHAIN_WAVES=$W"
  WD_CHAIN_WAVES=$W timeout 200 python scripts/bench_chain.py > $OUT/chain_w$W.txt 2>&1; grep -v "^tile stamps\|amdgpu.ids" $OUT/chain_w$W.txt | tail -n 6
done
B="--no-cpu-baseline --no-pmc"
for W in 4 8; do
  WD_CHAIN_WAVES=$W timeout 300 python bench.py --steps 20 --warmup 5 $B > $OUT/bench_w$W.json 2>> $OUT/bench.err
  python - $OUT/bench_w$W.json w$W <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-10s %.4f ms/step  %.1f M ex/s  %s tower %s parity %s" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d.get("repeats_ms_per_step"), d.get("roofline_tower"), d.get("parity")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
tail -n 5 $OUT/bench.err
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c2 or C2" > $OUT/pytest_fullsize.txt 2>&1; tail -n 5 $OUT/pytest_fullsize.txt
