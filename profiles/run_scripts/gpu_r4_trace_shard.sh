#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4shard}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
B="--no-cpu-baseline --no-pmc --no-parity"
WD_DIST_TEARDOWN=skip MASTER_PORT=29563 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --force-sharded --steps 40 --warmup 10 --pool 8 --repeats 2 $B ${BENCH_ARGS} > $OUT/prof.log 2>&1
tail -n 1 $OUT/prof.log | cut -c 1-200
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 64 1 > $OUT/step_timeline.txt; cat $OUT/step_timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python scripts/summarize_stats.py $OUT/kernel_stats.csv 104 > $OUT/kernel_stats.md; head -24 $OUT/kernel_stats.md
rm -rf $OUT/prof
