#!/bin/bash
# quick: chain parity + microbench + C2 bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/chainq; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -4
python scripts/bench_chain.py 2>&1 | tail -1
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2> $OUT/c2.err | cut -c1-160
