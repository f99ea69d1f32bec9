#!/bin/bash
# PMC passes over the stand-alone fp16 GEMMs (scripts/bench_hgemm.py): MFMA busy / waits, L2 hit rate, LDS conflicts
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-hg}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
run() { ITERS=3 WHICH=${WHICH:-nn} timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/p -o pmc -- python scripts/bench_hgemm.py > $OUT/log 2>&1
  python - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "k_hgemm" not in r["Kernel_Name"]: continue
    agg[(r["Kernel_Name"].split("k_hgemm")[1][:14], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
  rm -rf $OUT/p; }
run SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
