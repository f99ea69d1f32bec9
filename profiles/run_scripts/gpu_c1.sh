#!/bin/bash
# sparse-backward tests + C1 end-to-end at batch 512 and 8192 + a quick C2 bench (regression check) + C1 kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/c1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_c1.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest.log
timeout 300 python scripts/bench_c1.py > $OUT/b512.json 2> $OUT/b512.err
C1_BATCH=8192 C1_REPEAT=60 timeout 300 python scripts/bench_c1.py > $OUT/b8192.json 2> $OUT/b8192.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2> $OUT/c2.err | cut -c1-200 > $OUT/c2.json
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dist zipf 2> $OUT/c2z.err | cut -c1-200 > $OUT/c2zipf.json
C1_BATCH=8192 C1_REPEAT=30 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python scripts/bench_c1.py > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace*.csv" -delete
cat $OUT/pytest.log $OUT/b512.json $OUT/b8192.json $OUT/c2.json $OUT/c2zipf.json
head -25 $OUT/kernel_stats.csv | cut -c1-150
