#!/bin/bash
# round 6: the sharded owner's update on the flat row-update kernel -- the sharded tests, the one-rank lines A/B (WD_OWNER_FLAT), timeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r6shard2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
B="--no-cpu-baseline --no-pmc"
timeout 1800 python -m pytest tests/test_gpu_dist.py tests/test_gpu_c4.py -q -m gpu -x > $OUT/pytest_dist.txt 2>&1; tail -n 5 $OUT/pytest_dist.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print("%-40s %.4f ms/step %s parity %s" % (sys.argv[1], d["ms_per_step"], d.get("repeats_ms_per_step"), (d.get("parity") or {})))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
P=29580
for ENVS in "WD_OWNER_FLAT=1" "WD_OWNER_FLAT=0"; do
  P=$((P+1)); env $ENVS MASTER_PORT=$P timeout 400 python bench.py --steps 20 --warmup 5 --force-sharded $B > $OUT/bench_c2_sharded_$ENVS.json 2>> $OUT/bench.err; line "c2 sharded one rank [$ENVS]" $OUT/bench_c2_sharded_$ENVS.json
  P=$((P+1)); env $ENVS MASTER_PORT=$P timeout 400 python bench.py --steps 20 --warmup 5 --force-sharded --dist zipf $B --no-parity > $OUT/bench_c2_zipf_sharded_$ENVS.json 2>> $OUT/bench.err; line "c2 zipf sharded one rank [$ENVS]" $OUT/bench_c2_zipf_sharded_$ENVS.json
done
tail -n 3 $OUT/bench.err
WD_DIST_TEARDOWN=skip MASTER_PORT=29590 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 40 --warmup 5 --repeats 1 --force-sharded $B --no-parity > $OUT/prof_shard.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/sharded_one_rank_kernel_stats.csv
python scripts/summarize_stats.py $OUT/sharded_one_rank_kernel_stats.csv 40 > $OUT/sharded_one_rank_kernel_stats.md
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 30 1 > $OUT/sharded_one_rank_step_timeline.txt; cat $OUT/sharded_one_rank_step_timeline.txt
rm -rf $OUT/prof
