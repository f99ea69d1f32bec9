#!/bin/bash
# MFMA-busy / wait counters of the one-launch tower (with and without the fused gather phase); separate --pmc passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-chainpmc2}; mkdir -p $OUT
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  CHAIN_ITERS=3 timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o pmc -- python scripts/bench_chain.py > $OUT/p$i.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for sub in sorted(glob.glob(out + "/p[0-9]")):
    f = glob.glob(sub + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_tower_chain" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        res[c] = {"per_launch_mean": sum(v) / len(v), "launches": len(v), "min": min(v), "max": max(v)}
json.dump(res, open(out + "/tower_chain_pmc.json", "w"), indent=1)
g = lambda k: res.get(k, {}).get("per_launch_mean", float("nan"))
print("MFMA busy / (4 x wave cycles):", g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_WAVE_CYCLES")))
print("MFMA busy / (GUI_ACTIVE x 128):", g("SQ_VALU_MFMA_BUSY_CYCLES") / (128 * g("GRBM_GUI_ACTIVE")))
print("WAIT_ANY / wave cycles:", g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"))
print(json.dumps({k: round(v["per_launch_mean"]) for k, v in res.items()}))
PY
rm -rf $OUT/p[0-9]
