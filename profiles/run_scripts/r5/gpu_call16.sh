#!/bin/bash
# round 5, call 16: what do the L2 -> memory read counters charge per random row of the gather?  (calibration on its own access pattern)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call17; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_R[A-Z0-9_]*\|TCC_EA0_W[A-Z0-9_]*\|TCC_BUBBLE[A-Z_]*\|TCC_REQ[A-Z_]*\|TCC_READ[A-Z_]*\|TCC_MISS[A-Z_]*\|TCC_HIT[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/tcc_counters.txt; cat $OUT/tcc_counters.txt; echo
for ctrs in "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "WRITE_SIZE" "TCC_READ_SECTORS_sum TCC_READ_sum"; do
  for m in 6 2 1 g; do
    tag=$(echo $ctrs | tr ' ' '+')
    GC_MODE=$m timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/p -o c -- python scripts/bench_gather_calib.py > $OUT/log.txt 2>&1
    f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
    python - "$f" "$m" "$tag" <<'PY'
import csv, sys, collections
f, m, tag = sys.argv[1:4]
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "gather" if ("diag_gather" in k or "prefetch_onehot" in k) else ("copy" if "copy" in k.lower() else None)
        if name:
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, c), v in sorted(acc.items()):
        v = v[-8:] if name == "gather" else v
        print("mode %s  %-7s %-28s per launch %.1f  (n=%d)%s" % (m, name, c, sum(v) / len(v), len(v),
              "  per row %.2f" % (sum(v) / len(v) / 212992) if name == "gather" else "  per 256 MiB read: x%.4f of bytes/64" % (sum(v) / len(v) / (2**28 / 64))))
except Exception as e:
    print("mode", m, tag, "FAILED", e)
PY
    rm -rf $OUT/p
  done
done 2>&1 | tee $OUT/calib.txt
