#!/bin/bash
# round-3 baseline on today's box: bench line, step timeline of the single-GPU step, one-rank sharded step + its timeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3base; mkdir -p $OUT
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/bench_c2.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench_c2.json
MASTER_PORT=29561 timeout 300 python bench.py --force-sharded --steps 100 --warmup 10 --pool 8 --no-cpu-baseline --no-pmc 2> $OUT/g.err > $OUT/bench_shard1.json; cut -c1-200 $OUT/bench_shard1.json
MASTER_PORT=29563 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --force-sharded --steps 40 --warmup 5 --pool 4 --no-cpu-baseline --no-pmc > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/timeline.py $T k_tower_chain 10 > $OUT/shard_timeline.txt; cat $OUT/shard_timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/shard_kernel_stats.csv
python scripts/summarize_stats.py $OUT/shard_kernel_stats.csv 45 > $OUT/shard_kernel_stats.md
rm -rf $OUT/prof
