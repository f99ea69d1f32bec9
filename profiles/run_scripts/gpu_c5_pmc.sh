#!/bin/bash
# MFMA-busy counters of the C5 bench command itself (fp16-operand tower): one --pmc pass, aggregated per kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-c5pmc}; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/p -o pmc -- \
  python bench.py --config c5 --steps 8 --warmup 2 --pool 4 --no-graph --no-cpu-baseline --no-parity --no-pmc > $OUT/bench.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
f = glob.glob(out + "/p/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
rows = []
for k, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0 or "hgemm" not in k and "tower" not in k and "gemm" not in k: continue
    rows.append((gui, k, cnt[k], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (128.0 * gui), c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
rows.sort(reverse=True)
tot_g = sum(r[0] for r in rows); tot_b = sum(r[3] * r[0] for r in rows)
with open(out + "/c5_mfma_pmc.md", "w") as fo:
    fo.write("| kernel | launches | GRBM_GUI_ACTIVE cycles | MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GUI_ACTIVE) | SQ_WAIT_ANY / SQ_WAVE_CYCLES |\n|---|---:|---:|---:|---:|\n")
    for g, k, n, b, w in rows:
        fo.write("| `%s` | %d | %.0f | %.1f %% | %.0f %% |\n" % (k, n, g, 100 * b, 100 * w))
    fo.write("\nGEMM launches together: MFMA busy %.1f %% (cycle-weighted)\n" % (100 * tot_b / max(tot_g, 1)))
print(open(out + "/c5_mfma_pmc.md").read())
PY
rm -rf $OUT/p
