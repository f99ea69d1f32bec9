#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the bench command -> gpurun_out/prof_*/ ; copy summaries to profiles/ afterwards.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --no-cpu-baseline > $OUT/bench.log 2>&1
find $OUT -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace*.csv" -delete
tail -2 $OUT/bench.log
head -40 $OUT/kernel_stats.csv
