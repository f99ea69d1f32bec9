#!/bin/bash
# round 4, call 2: the 8-wavefront tower with GLOBAL (not flat) weight loads + its experiment builds (stage cycles), same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_prefetch.py -q -m gpu -x > $OUT/pytest_chain.txt 2>&1; tail -n 2 $OUT/pytest_chain.txt
show() { grep -v "^tile stamps\|amdgpu.ids\|wave0 F0\|row tile" $1 | tail -n 3; }
echo "== 4 waves (round 3)"; WD_CHAIN_WAVES=4 timeout 200 python scripts/bench_chain.py > $OUT/chain_w4.txt 2>&1; show $OUT/chain_w4.txt
echo "== 8 waves"; timeout 200 python scripts/bench_chain.py > $OUT/chain_w8.txt 2>&1; show $OUT/chain_w8.txt
for E in 1 2 4 6 8 16 14; do
  echo "== 8 waves, EXP $E"; WD_HIP_LIB=$PWD/wide_deep_amd/_lib/libwd_hip_exp8_$E.so timeout 200 python scripts/bench_chain.py > $OUT/chain_w8_exp$E.txt 2>&1; show $OUT/chain_w8_exp$E.txt
done
B="--no-cpu-baseline --no-pmc --no-parity"
for W in 4 8; do
  WD_CHAIN_WAVES=$W timeout 300 python bench.py --steps 20 --warmup 5 $B > $OUT/bench_w$W.json 2>> $OUT/bench.err
  python - $OUT/bench_w$W.json w$W <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-10s %.4f ms/step  %.1f M ex/s  %s tower %s us" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d.get("repeats_ms_per_step"), d.get("roofline_tower", {}).get("avg_launch_us")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
