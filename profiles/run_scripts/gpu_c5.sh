#!/bin/bash
# C5 (BASELINE configs[4]) on the GPU box: fp16 vs fp32 tower step time, kernel stats, MFMA-busy counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-c5}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python bench.py --config c5 --no-cpu-baseline --steps 30 --warmup 5 --pool 8 2>/dev/null | tail -1 > $OUT/bench_c5_fp16.json; cut -c1-330 $OUT/bench_c5_fp16.json
python bench.py --config c5 --tower-dtype fp32 --no-cpu-baseline --steps 20 --warmup 5 --pool 8 2>/dev/null | tail -1 > $OUT/bench_c5_fp32.json; cut -c1-200 $OUT/bench_c5_fp32.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --config c5 --steps 20 --warmup 5 --pool 4 --no-cpu-baseline > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; find $OUT/prof -name "*kernel_trace*.csv" -delete
python scripts/summarize_stats.py $OUT/kernel_stats.csv 27 | head -16
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pmc -- python bench.py --config c5 --steps 3 --warmup 1 --pool 2 --no-cpu-baseline --no-graph > $OUT/pmc.log 2>&1
python - <<'PY' $OUT
import csv, glob, sys, collections, json
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "k_hgemm" not in r["Kernel_Name"]: continue
    key = r["Kernel_Name"].split("k_hgemm")[1][:12] + " grid " + r["Grid_Size"]
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    # MFMA busy fraction = MFMA-busy SIMD cycles / (kernel cycles x 1024 SIMDs); GRBM_GUI_ACTIVE = kernel cycles
    if m.get("GRBM_GUI_ACTIVE"): m["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] * 1024)
    res[k] = {c: round(v, 4) if c == "mfma_busy_frac" else round(v) for c, v in m.items()}
    print(k, res[k])
json.dump(res, open(out + "/mfma_pmc.json", "w"), indent=1)
PY
