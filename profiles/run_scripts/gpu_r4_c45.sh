#!/bin/bash
# connection-list tests + kernel-trace timelines / stats of C4 and C5 (one step inside a graph replay each)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4c45}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_step.py -q -m gpu -x > $OUT/pytest_step.txt 2>&1; tail -n 3 $OUT/pytest_step.txt
B="--no-cpu-baseline --no-pmc --no-parity"
for c in c4 c5; do
  anchor=k_input_layer; [ $c = c5 ] && anchor=k_fold_affine_all
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --config $c --steps 40 --warmup 10 --repeats 1 $B > $OUT/prof_$c.log 2>&1
  tail -n 1 $OUT/prof_$c.log | cut -c 1-200
  T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
  python scripts/trace_window.py $T $anchor 30 1 > $OUT/${c}_step_timeline.txt; cat $OUT/${c}_step_timeline.txt | cut -c1-110
  find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/${c}_kernel_stats.csv
  python scripts/summarize_stats.py $OUT/${c}_kernel_stats.csv 50 > $OUT/${c}_kernel_stats.md; head -30 $OUT/${c}_kernel_stats.md | cut -c1-160
  rm -rf $OUT/prof
done
