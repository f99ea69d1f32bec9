#!/bin/bash
# f4 rows: optimizers, first_dense, eval/pred CLIs -- then the full suite + quick bench (regression check of the default path)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/f4; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_optimizers.py tests/test_gpu_step.py tests/test_gpu_c1.py -m gpu -x -q 2>&1 | tail -25 > $OUT/pytest_new.log
cat $OUT/pytest_new.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest_all.log
cat $OUT/pytest_all.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2> $OUT/c2.err | cut -c1-170
