#!/bin/bash
# A/B of step variants on one box (same clocks): C2 uniform
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ab; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_step.py tests/test_gpu_c1.py -m gpu -x -q 2>&1 | tail -2

for V in none bucket none bucket; do
  echo "WD_OVERLAP=$V"; WD_OVERLAP=$V timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | cut -c1-150
done
