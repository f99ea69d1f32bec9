#!/bin/bash
# the sharded step on ONE rank over a one-rank RCCL group: exchange-kernel cost + graph segments vs eager launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/shard1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -3
MASTER_PORT=29561 timeout 300 python bench.py --force-sharded --steps 100 --warmup 10 --pool 8 --no-cpu-baseline 2> $OUT/g.err | cut -c1-200
MASTER_PORT=29562 timeout 300 python bench.py --force-sharded --steps 100 --warmup 10 --pool 8 --no-cpu-baseline --no-graph 2> $OUT/e.err | cut -c1-200
tail -2 $OUT/e.err
MASTER_PORT=29563 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --force-sharded --steps 40 --warmup 5 --pool 4 --no-cpu-baseline > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace*.csv" -delete
python scripts/summarize_stats.py $OUT/kernel_stats.csv 53 | head -12
