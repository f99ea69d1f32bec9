#!/bin/bash
# Is the one-launch tower (223 KB of code against a 64 KB instruction cache per CU pair) waiting for instruction fetch?
# One --pmc pass over scripts/bench_chain.py: I-cache requests / hits / misses of k_tower_chain<32> beside its wave cycles.
# PMC_COUNTERS / PMC_NAME: another counter set through the same script (r3b_tower_issue_pmc.json: VALU / SALU / LDS issue cycles).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3icache}; mkdir -p $OUT
CHAIN_ITERS=3 timeout 150 rocprofv3 --kernel-trace --pmc ${PMC_COUNTERS:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} --output-format csv -d $OUT/pmc -o pmc -- python scripts/bench_chain.py > $OUT/pmc.log 2>&1
tail -3 $OUT/pmc.log
PMC_NAME=${PMC_NAME:-icache} python - <<'PY' $OUT
import csv, glob, sys, collections, json
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    name = "k_tower_chain" if "k_tower_chain" in k else ("k_gemm_tn_group" if "k_gemm_tn_group" in k else None)
    if name: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: round(sum(v) / len(v)) for c, v in d.items()} for k, d in agg.items()}
import os
json.dump(res, open(out + "/%s_pmc.json" % os.environ.get("PMC_NAME", "icache"), "w"), indent=1); print(json.dumps(res, indent=1))
PY
rm -rf $OUT/pmc
