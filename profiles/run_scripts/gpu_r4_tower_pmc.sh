#!/bin/bash
# MFMA-busy / wait / issue counters of the one-launch tower (k_tower_chain8) as the bench command launches it (the full training
# launch: x from HBM, wide weight list, forward + head + gradient chain + dx); separate --pmc passes (counter collection serialises kernels)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r4_tower_pmc}; mkdir -p $OUT
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCC_HIT TCC_MISS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o pmc -- python bench.py --steps 20 --warmup 5 --repeats 1 --pool 16 --no-cpu-baseline --no-pmc --no-parity > $OUT/p$i.log 2>&1
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
g = lambda k: res.get(k, {}).get("per_launch_mean", float("nan"))
summary = {"kernel": "k_tower_chain8 (C2 tower, batch 8192: every launch of `python bench.py --steps 20 --warmup 5` under rocprofv3 --pmc, mean per launch)",
           "note": "SQ_WAVE_CYCLES counts quad-cycles summed over the launch's 2048 wavefronts (two per SIMD): SIMD-cycles = 4 * SQ_WAVE_CYCLES / 2; the round-3 kernel ran one wavefront per SIMD (busy / (4 * SQ_WAVE_CYCLES))",
           "mfma_busy_over_simd_cycles": g("SQ_VALU_MFMA_BUSY_CYCLES") / (2 * g("SQ_WAVE_CYCLES")),
           "mfma_busy_cycles_floor_per_launch": 77.8e3 * 1024,
           "mfma_busy_over_gui_active_x_1024_simds": g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("GRBM_GUI_ACTIVE")),
           "wait_any_over_wave_cycles": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
           "l2_read_latency_cycles": g("TCP_TCC_READ_REQ_LATENCY") / g("TCP_TCC_READ_REQ"),
           "l2_hit_rate": g("TCC_HIT") / (g("TCC_HIT") + g("TCC_MISS")),
           "counters": {k: round(v["per_launch_mean"]) for k, v in res.items()}}
json.dump(summary, open(out + "/tower_chain8_pmc.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
rm -rf $OUT/p[0-9]
