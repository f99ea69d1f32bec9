#!/bin/bash
# round 3, sharded engine: parity tests (world 2 / 4 on one GPU, gloo staging), the one-rank RCCL step (collectives captured in
# the multi-step hipGraph) + its kernel timeline, and the N=2 bench launched from a bare `python bench.py --gpus 2`
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r3shard}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ "${TESTS:-1}" = "1" ]; then timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -15; fi
MASTER_PORT=29561 timeout 200 python bench.py --force-sharded --steps 100 --warmup 10 --pool 8 --no-cpu-baseline --no-pmc 2> $OUT/g.err > $OUT/bench_shard1.json; cut -c1-260 $OUT/bench_shard1.json; tail -3 $OUT/g.err
MASTER_PORT=29562 timeout 200 python bench.py --force-sharded --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2> $OUT/g2.err > $OUT/bench_shard1_driver_args.json; cut -c1-260 $OUT/bench_shard1_driver_args.json; tail -3 $OUT/g2.err
WD_DIST_TEARDOWN=skip MASTER_PORT=29563 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --force-sharded --steps 40 --warmup 10 --pool 8 --repeats 2 --no-cpu-baseline --no-pmc > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/timeline.py $T k_tower_chain 30 > $OUT/shard_timeline.txt; cat $OUT/shard_timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/shard_kernel_stats.csv
python scripts/summarize_stats.py $OUT/shard_kernel_stats.csv 104 > $OUT/shard_kernel_stats.md
rm -rf $OUT/prof
if [ "${GLOO2:-1}" = "1" ]; then
WD_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --pool 4 --repeats 2 --no-cpu-baseline --no-pmc 2> $OUT/gloo2.err > $OUT/bench_gloo2.json; echo "gloo2 rc=$?"; cut -c1-300 $OUT/bench_gloo2.json; tail -3 $OUT/gloo2.err
fi
