#!/bin/bash
# round evidence: full GPU suite, bench lines (uniform with cpu_baseline, zipf, other configs), rocprofv3 kernel stats of the
# bench command, PMC of the tower kernel (MFMA busy).  Output -> gpurun_out/<tag>/ ; copy what is judged into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-final}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_c2_uniform.json 2> $OUT/bench.err; cut -c1-260 $OUT/bench_c2_uniform.json
timeout 300 python bench.py --dist zipf --no-cpu-baseline > $OUT/bench_c2_zipf.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c2_zipf.json
timeout 300 python bench.py --config c3 --no-cpu-baseline --steps 100 > $OUT/bench_c3.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c3.json
timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 100 > $OUT/bench_c4.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c4.json
timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 50 > $OUT/bench_c5_fp16.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c5_fp16.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --no-cpu-baseline > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace*.csv" -delete; find $OUT/prof -name "*.db" -delete
python scripts/summarize_stats.py $OUT/kernel_stats.csv 72 > $OUT/kernel_stats.md; head -14 $OUT/kernel_stats.md
python scripts/bench_chain.py > $OUT/chain_stamps.txt 2>&1; tail -4 $OUT/chain_stamps.txt
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
# bench_chain.py launches: train step, 5+3 full, 5+3 no-dx, 5+3 forward-only, 1 stamped  -> the first 9 are full launches
res = {c: round(sum(v[1:9]) / 8) for c, v in agg.items()}
res["note"] = "k_tower_chain, C2 tower, batch 8192, mean of 8 full launches (forward + head + gradient chain + dx); SQ_* summed over all waves / SIMDs"
if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "SQ_BUSY_CYCLES" in res:
    res["mfma_busy_over_sq_busy"] = round(res["SQ_VALU_MFMA_BUSY_CYCLES"] / max(res["SQ_BUSY_CYCLES"], 1) / 4, 4)
json.dump(res, open(out + "/chain_pmc.json", "w"), indent=1); print(res)
PY
