#!/bin/bash
# rocprofv3 kernel stats of a bench command.  usage: gpu_stats.sh <tag> <steps-in-trace> [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-st}; N=${2:-40}; shift; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --no-cpu-baseline --no-parity --no-pmc "$@" > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python scripts/summarize_stats.py $OUT/kernel_stats.csv $N > $OUT/kernel_stats.md; head -28 $OUT/kernel_stats.md
tail -c 400 $OUT/prof.log | grep -o '"ms_per_step": [0-9.]*'
rm -rf $OUT/prof
