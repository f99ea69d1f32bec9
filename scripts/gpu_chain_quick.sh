#!/bin/bash
# quick: chain parity + microbench + C2 bench (+ kernel stats of the top kernels)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/chainq; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_step.py -m gpu -x -q 2>&1 | tail -4
python scripts/bench_chain.py 2>&1 | tail -4
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2> $OUT/c2.err | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --no-cpu-baseline > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace*.csv" -delete
python scripts/summarize_stats.py $OUT/kernel_stats.csv 72 | head -14
