#!/bin/bash
# round-6 evidence of the final tree, ONE box: the GPU suite + smoke, the bench line of every configuration, A/B lines, kernel stats
# + step timelines, the gather's in-step duration two ways.  Output -> gpurun_out/<tag>/ ; what is judged is copied into profiles/ (r6_*).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r6}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ "${PYTEST:-1}" = 1 ]; then
  timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -n 3 $OUT/pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
fi
B="--no-cpu-baseline --no-pmc"
r() { name=$1; shift; "$@" > $OUT/bench_$name.json 2>> $OUT/bench.err; python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ro, tw = d.get("roofline") or {}, d.get("roofline_tower") or {}
    print("%-36s %.4f ms/step  %.1f M ex/s  %s | roofline %s us = %s (rocprof %s) | tower %s us = %s" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d.get("repeats_ms_per_step"), ro.get("avg_launch_us"), ro.get("frac"), ro.get("frac_rocprof_instep"), tw.get("avg_launch_us"), tw.get("frac")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
r c2_uniform_driver_args timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-stamps $OUT/gather_instep_stamps.txt
r c2_uniform timeout 600 python bench.py --no-cpu-baseline
r c2_zipf timeout 300 python bench.py --dist zipf --steps 20 --warmup 5 $B
r c3 timeout 300 python bench.py --config c3 $B --steps 100
r c4_tokens timeout 400 python bench.py --config c4 $B --steps 60
r c4_ids_resident timeout 400 python bench.py --config c4 $B --steps 60 --ids-input --no-parity
r c4_nocross timeout 300 python bench.py --config c4-nocross $B --steps 100
r c5_fp16 timeout 300 python bench.py --config c5 $B --steps 60
r c2_eager_launches timeout 300 python bench.py --no-graph $B --no-parity --steps 100
MASTER_PORT=29561 r c2_sharded_one_rank timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded $B
MASTER_PORT=29562 r c2_zipf_sharded_one_rank timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded --dist zipf $B --no-parity
MASTER_PORT=29564 r c4_sharded_one_rank timeout 400 python bench.py --config c4 --steps 20 --warmup 5 --force-sharded $B
WD_DIST_BACKEND=gloo r c2_gloo_two_ranks_one_gpu timeout 400 python bench.py --gpus 2 --steps 10 --warmup 2 --pool 4 --repeats 2 --no-pmc --cpu-steps 5
# kernel stats + two-step timeline of the bench command; the gather's in-step kernel-trace duration
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --gpus 1 --steps 20 --warmup 5 $B --no-parity > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/gather_instep_from_trace.py $T $OUT/gather_instep_rocprof.json
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/c2_step_timeline.txt; grep -v hash_bucket $OUT/c2_step_timeline.txt | head -20
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c2_uniform_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c2_uniform_kernel_stats.csv 230 > $OUT/c2_uniform_kernel_stats.md; head -12 $OUT/c2_uniform_kernel_stats.md
rm -rf $OUT/prof
# the sharded step on a one-rank RCCL group
WD_DIST_TEARDOWN=skip MASTER_PORT=29563 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --force-sharded --steps 40 --warmup 10 --pool 8 --repeats 2 $B --no-parity > $OUT/prof_shard.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 64 1 > $OUT/sharded_one_rank_step_timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/sharded_one_rank_kernel_stats.csv
python scripts/summarize_stats.py $OUT/sharded_one_rank_kernel_stats.csv 104 > $OUT/sharded_one_rank_kernel_stats.md; head -14 $OUT/sharded_one_rank_kernel_stats.md
rm -rf $OUT/prof
# configs[3] from tokens: kernel stats + timeline
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --config c4 --steps 60 --warmup 10 --pool 16 --repeats 1 $B --no-parity > $OUT/prof_c4.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c4_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c4_kernel_stats.csv 75 > $OUT/c4_kernel_stats.md; head -22 $OUT/c4_kernel_stats.md
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 40 1 > $OUT/c4_step_timeline.txt; cat $OUT/c4_step_timeline.txt
rm -rf $OUT/prof
# the featurizer alone
timeout 200 python scripts/bench_featurizer.py --check > $OUT/featurizer.json 2>> $OUT/bench.err; cat $OUT/featurizer.json
tail -n 5 $OUT/bench.err
