#!/bin/bash
# kernel-trace timeline of the bench command (two steps inside a graph replay) + kernel stats; env passes through (WD_CHAIN_WAVES ...)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4trace}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
B="--no-cpu-baseline --no-pmc --no-parity"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --repeats 1 $B ${BENCH_ARGS} > $OUT/prof.log 2>&1
tail -n 1 $OUT/prof.log | cut -c 1-300
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/step_timeline.txt; cat $OUT/step_timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python scripts/summarize_stats.py $OUT/kernel_stats.csv 70 > $OUT/kernel_stats.md; head -16 $OUT/kernel_stats.md
rm -rf $OUT/prof
