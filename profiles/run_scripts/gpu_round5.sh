#!/bin/bash
# round-5 evidence of the final tree: bench lines of every configuration, A/B lines on the same box, kernel stats + timelines.
# Output -> gpurun_out/<tag>/ ; what is judged is copied into profiles/ (r5_*).  PYTEST=1 also runs the GPU suite + smoke first.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r5}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ "${PYTEST:-0}" = 1 ]; then
  timeout 1700 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -n 3 $OUT/pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
fi
B="--no-cpu-baseline --no-pmc"
r() { name=$1; shift; "$@" > $OUT/bench_$name.json 2>> $OUT/bench.err; python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ro, tw = d.get("roofline") or {}, d.get("roofline_tower") or {}
    print("%-36s %.4f ms/step  %.1f M ex/s  %s | roofline %s us = %s | tower %s us = %s" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d.get("repeats_ms_per_step"), ro.get("avg_launch_us"), ro.get("frac"), tw.get("avg_launch_us"), tw.get("frac")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
r c2_uniform_driver_args timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5
r c2_uniform timeout 600 python bench.py --no-cpu-baseline
r c2_zipf timeout 300 python bench.py --dist zipf --steps 20 --warmup 5 $B
r c3 timeout 300 python bench.py --config c3 $B --steps 100
r c4 timeout 400 python bench.py --config c4 $B --steps 60
r c4_nocross timeout 300 python bench.py --config c4-nocross $B --steps 100
WD_SMALL_TABLES=0 r c4_general_path timeout 400 python bench.py --config c4 $B --steps 40 --no-parity
r c5_fp16 timeout 300 python bench.py --config c5 $B --steps 60
WD_TN_STREAM=1 r c2_streamed_products timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --no-parity
WD_PIPE_SIDE=sort r c2_sort_before_gather timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --no-parity
r c2_eager_launches timeout 300 python bench.py --no-graph $B --no-parity --steps 100
MASTER_PORT=29561 r c2_sharded_one_rank timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded $B
MASTER_PORT=29562 r c2_zipf_sharded_one_rank timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded --dist zipf $B --no-parity
WD_DIST_BACKEND=gloo r c2_gloo_two_ranks_one_gpu timeout 400 python bench.py --gpus 2 --steps 10 --warmup 2 --pool 4 --repeats 2 --no-pmc --cpu-steps 5
# kernel stats + two-step timeline of the bench command
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --repeats 1 $B --no-parity > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/c2_step_timeline.txt; grep -v hash_bucket $OUT/c2_step_timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c2_uniform_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c2_uniform_kernel_stats.csv 70 > $OUT/c2_uniform_kernel_stats.md; head -12 $OUT/c2_uniform_kernel_stats.md
rm -rf $OUT/prof
# the sharded step on a one-rank RCCL group
WD_DIST_TEARDOWN=skip MASTER_PORT=29563 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --force-sharded --steps 40 --warmup 10 --pool 8 --repeats 2 $B --no-parity > $OUT/prof_shard.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 64 1 > $OUT/sharded_one_rank_step_timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/sharded_one_rank_kernel_stats.csv
python scripts/summarize_stats.py $OUT/sharded_one_rank_kernel_stats.csv 104 > $OUT/sharded_one_rank_kernel_stats.md; head -14 $OUT/sharded_one_rank_kernel_stats.md
rm -rf $OUT/prof
# configs[3] with its crossed columns: kernel stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --config c4 --steps 30 --warmup 5 --pool 8 --repeats 1 $B --no-parity > $OUT/prof_c4.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c4_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c4_kernel_stats.csv 35 > $OUT/c4_kernel_stats.md; head -14 $OUT/c4_kernel_stats.md
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 20 1 > $OUT/c4_step_timeline.txt; cat $OUT/c4_step_timeline.txt
rm -rf $OUT/prof
# the tower alone: stage cycles
timeout 200 python scripts/bench_chain.py > $OUT/tower_chain8_stage_cycles.txt 2>&1; grep "^chain B\|^workgroup 0" $OUT/tower_chain8_stage_cycles.txt
CHAIN_MODE=resnet timeout 200 python scripts/bench_chain.py > $OUT/tower_chain8_resnet_stage_cycles.txt 2>&1; grep "^chain B\|^workgroup 0" $OUT/tower_chain8_resnet_stage_cycles.txt
tail -n 5 $OUT/bench.err
