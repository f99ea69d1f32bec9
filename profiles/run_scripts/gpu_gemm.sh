#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-g1}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python scripts/bench_gemm.py > $OUT/gemm.txt 2>&1; cat $OUT/gemm.txt
GEMM_ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc -o pmc -- python scripts/bench_gemm.py > $OUT/pmc.log 2>&1
python - <<'PY' $OUT
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "k_gemm" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"].split("k_gemm")[1][:22], r["Grid_Size"])
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
