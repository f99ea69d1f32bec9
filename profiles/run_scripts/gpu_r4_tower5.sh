#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4h}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_prefetch.py -q -m gpu -x > $OUT/pytest_chain.txt 2>&1; tail -n 2 $OUT/pytest_chain.txt
show() { grep "^chain B\|^workgroup 0" $1; }
echo "== default"; timeout 200 python scripts/bench_chain.py > $OUT/chain_w8.txt 2>&1; show $OUT/chain_w8.txt
for E in ${EXPS:-32 16 48}; do
  echo "== EXP $E"; WD_HIP_LIB=$PWD/wide_deep_amd/_lib/libwd_hip_exp8_$E.so timeout 200 python scripts/bench_chain.py > $OUT/chain_w8_exp$E.txt 2>&1; show $OUT/chain_w8_exp$E.txt
done
B="--no-cpu-baseline --no-pmc --no-parity --steps 20 --warmup 5"
for E in 0 ${EXPS:-32 16 48}; do
  L=$PWD/wide_deep_amd/_lib/libwd_hip_exp8_$E.so; [ $E = 0 ] && L=$PWD/wide_deep_amd/_lib/libwd_hip.so
  WD_HIP_LIB=$L timeout 300 python bench.py $B > $OUT/bench_e$E.json 2>> $OUT/bench.err
  python - $OUT/bench_e$E.json exp$E <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-10s %.4f ms/step  %.1f M ex/s  %s tower %s us" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d.get("repeats_ms_per_step"), d.get("roofline_tower", {}).get("avg_launch_us")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
