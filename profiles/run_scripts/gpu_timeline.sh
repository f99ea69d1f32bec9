#!/bin/bash
# step timeline of the bench command: rocprofv3 kernel trace -> scripts/timeline.py.  usage: gpu_timeline.sh <tag> [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-tl}; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 64 --warmup 16 --pool 16 --no-cpu-baseline --no-parity "$@" > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/timeline.py $T > $OUT/timeline.txt; cat $OUT/timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python scripts/summarize_stats.py $OUT/kernel_stats.csv 80 > $OUT/kernel_stats.md
cp $T $OUT/kernel_trace.csv; rm -rf $OUT/prof
