#!/bin/bash
# round-3 evidence of the FINAL tree (after gpu_round3.sh: write-through tower stores, chained graphs, sender-side unique):
# GPU suite + smoke, bench lines, A/B lines on the same box, kernel stats + timelines, tower PMC.  Output -> gpurun_out/<tag>/ ;
# what is judged is copied into profiles/ (r3b_*).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r3b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ "${PYTEST:-1}" = 1 ]; then
  timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -n 3 $OUT/pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
fi
B="--no-cpu-baseline --no-pmc"
r() { name=$1; shift; "$@" > $OUT/bench_$name.json 2>> $OUT/bench.err; python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-34s %.4f ms/step  %.1f M ex/s  %s" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d.get("repeats_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
r c2_uniform timeout 600 python bench.py
r c2_uniform_driver_args timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
r c2_uniform_plain_tower_stores env WD_WT=0 timeout 300 python bench.py --steps 20 --warmup 5 $B --no-parity
r c2_uniform_round2_step env WD_INPUT_AHEAD=0 WD_WT=0 timeout 300 python bench.py --steps 20 --warmup 5 $B --no-parity
r c2_zipf timeout 300 python bench.py --dist zipf --steps 20 --warmup 5 $B
r c2_zipf_round2_step env WD_INPUT_AHEAD=0 WD_WT=0 timeout 300 python bench.py --steps 20 --warmup 5 --dist zipf $B --no-parity
r c3 timeout 300 python bench.py --config c3 $B --steps 100
r c4 timeout 300 python bench.py --config c4 $B --steps 100
r c5_fp16 timeout 300 python bench.py --config c5 $B --steps 60
r c2_eager_launches timeout 300 python bench.py --no-graph $B --no-parity --steps 100
MASTER_PORT=29561 r c2_sharded_one_rank timeout 200 python bench.py --steps 20 --warmup 5 --force-sharded $B --no-parity
MASTER_PORT=29562 r c2_zipf_sharded_one_rank timeout 200 python bench.py --steps 20 --warmup 5 --force-sharded --dist zipf $B --no-parity
MASTER_PORT=29564 r c2_zipf_sharded_one_rank_per_occurrence env WD_SHARD_DEDUP=0 timeout 200 python bench.py --steps 20 --warmup 5 --force-sharded --dist zipf $B --no-parity
WD_DIST_BACKEND=gloo r c2_gloo_two_ranks_one_gpu timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --pool 4 --repeats 2 $B
# kernel stats + two-step timeline of the bench command
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --repeats 1 $B --no-parity > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/c2_step_timeline.txt; cat $OUT/c2_step_timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c2_uniform_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c2_uniform_kernel_stats.csv 70 > $OUT/c2_uniform_kernel_stats.md; head -14 $OUT/c2_uniform_kernel_stats.md
rm -rf $OUT/prof
# the sharded step on a one-rank RCCL group, Zipf ids, sender-side unique
WD_DIST_TEARDOWN=skip MASTER_PORT=29563 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --force-sharded --dist zipf --steps 40 --warmup 10 --pool 8 --repeats 2 $B > $OUT/prof_shard.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 64 1 > $OUT/sharded_zipf_unique_step_timeline.txt; cat $OUT/sharded_zipf_unique_step_timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/sharded_zipf_unique_kernel_stats.csv
python scripts/summarize_stats.py $OUT/sharded_zipf_unique_kernel_stats.csv 104 > $OUT/sharded_zipf_unique_kernel_stats.md
rm -rf $OUT/prof
# tower MFMA PMC
CHAIN_ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pmc -- python scripts/bench_chain.py > $OUT/pmc.log 2>&1
python - <<'PY' $OUT
import csv, glob, sys, collections, json
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); sys.exit()
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_tower_chain" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {c: round(sum(v[1:9]) / 8) for c, v in agg.items()}
res["note"] = "k_tower_chain<32>, C2 tower, batch 8192, x from HBM, write-through output stores, mean of 8 full launches; SQ_* summed over all waves / SIMDs"
if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "SQ_WAVE_CYCLES" in res:
    res["mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES)"] = round(res["SQ_VALU_MFMA_BUSY_CYCLES"] / max(res["SQ_WAVE_CYCLES"], 1) / 4, 4)
json.dump(res, open(out + "/tower_chain_pmc.json", "w"), indent=1); print(res)
PY
rm -rf $OUT/pmc
