#!/bin/bash
# quick A/B on the GPU box: bench at several WD_BUCKET_TARGET values + kernel stats of the last
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-q}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
for T in 64 128 256; do
  echo "== target $T"; WD_BUCKET_TARGET=$T python bench.py --no-cpu-baseline --steps 100 2>/dev/null | cut -c1-140
  WD_BUCKET_TARGET=$T python bench.py --no-cpu-baseline --steps 100 --dist zipf 2>/dev/null | cut -c1-140
done
WD_BUCKET_TARGET=128 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --no-cpu-baseline > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace*.csv" -delete
python scripts/summarize_stats.py $OUT/kernel_stats.csv 72 | grep -i "bucket\|head" 
