#!/bin/bash
# round 4: the gather's ceiling with counters -- request order and row layout; address-translation and memory-side counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4gather}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ "${SKIP_TIMING:-0}" != 1 ]; then timeout 600 python scripts/bench_gather_ceiling.py > $OUT/gather_ceiling.txt 2> $OUT/gather_ceiling.err; cat $OUT/gather_ceiling.txt; tail -n 2 $OUT/gather_ceiling.err; fi
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TCP_UTCL1[A-Z_0-9]*\|UTCL2[A-Z_0-9]*\|TCP_TCC_READ_REQ[A-Z_0-9]*\|TCC_EA0\?_RDREQ[A-Z_0-9]*\|TCC_HIT[A-Z_0-9]*\|TCC_MISS[A-Z_0-9]*\|TCC_REQ[A-Z_0-9]*\|TCP_TA_DATA_STALL[A-Z_0-9]*\|TCP_PENDING_STALL[A-Z_0-9]*\|TCC_EA0\?_RD_UNCACHED[A-Z_0-9]*\|TCC_TAG_STALL[A-Z_0-9]*\|TCC_BUBBLE[A-Z_0-9]*\|FETCH_SIZE\|TCP_TOTAL_CACHE_ACCESSES[A-Z_0-9]*\|TCP_TCC_NC_READ_REQ[A-Z_0-9]*\|TCP_READ_TAGCONFLICT[A-Z_0-9]*\)\b" | sort -u > $OUT/counters_avail.txt
echo "available counters of interest:"; tr '\n' ' ' < $OUT/counters_avail.txt; echo
have() { grep -qx "$1" $OUT/counters_avail.txt; }
SETS=()
s=""; for c in TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS; do have $c && s="$s $c"; done; [ -n "$s" ] && SETS+=("$s")
s=""; for c in TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ; do have $c && s="$s $c"; done; [ -n "$s" ] && SETS+=("$s")
s=""; for c in TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_UTCL1_STALL_INFLIGHT_MAX; do have $c && s="$s $c"; done; [ -n "$s" ] && SETS+=("$s")
s=""; for c in TCC_EA0_RDREQ_DRAM TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL; do have $c && s="$s $c"; done; [ -n "$s" ] && SETS+=("$s")
s=""; for c in TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_THRASHING_STALL; do have $c && s="$s $c"; done; [ -n "$s" ] && SETS+=("$s")
for CELL in 32,random,1,6 32,sorted,1,6 16,random,1,6 16,sorted,1,6 32,random,16,6 32,zipf_sorted_unique,1,6; do
  i=0
  for SET in "${SETS[@]}"; do
    i=$((i+1))
    GC_ONLY=$CELL GC_ITERS=6 timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p -o pmc -- python scripts/bench_gather_ceiling.py > $OUT/pmc_${CELL}_$i.log 2>&1
    python - $OUT/p "$CELL" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if f:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gather_modes" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("  %-28s" % sys.argv[2], "  ".join("%s %.4g" % (c, sum(v[-6:]) / len(v[-6:])) for c, v in sorted(agg.items())), flush=True)
PY
    rm -rf $OUT/p
  done
done | tee $OUT/gather_pmc.txt
