#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-ga}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python scripts/bench_gather.py > $OUT/gather.txt 2>&1; cat $OUT/gather.txt | tail -2
GATHER_DIST=zipf python scripts/bench_gather.py 2>&1 | tail -1 | tee -a $OUT/gather.txt
for C in FETCH_SIZE WRITE_SIZE; do
  GATHER_ITERS=20 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python scripts/bench_gather.py > $OUT/pmc_$C.log 2>&1
done
python - <<'PY' $OUT
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % C, recursive=True)
    if not f: print("no csv for", C); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != C: continue
        n = r["Kernel_Name"]
        key = "gather" if "k_embag_fwd" in n else ("copy" if "copy" in n.lower() or "direct_copy" in n else None)
        if key: agg[key].append(float(r["Counter_Value"]))
    res[C] = {k: {"n": len(v), "mean": sum(v) / len(v), "last": v[-1]} for k, v in agg.items()}
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
PY
