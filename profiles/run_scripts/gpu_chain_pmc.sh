#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/chainpmc; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_raw.txt 2>&1
grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|SQ_[A-Z0-9_]*" $OUT/counters_raw.txt | sort -u > $OUT/counter_names.txt
wc -l $OUT/counter_names.txt
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  CHAIN_ITERS=3 timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o pmc -- python scripts/bench_chain.py > $OUT/p$i.log 2>&1
done
python - <<'PY' $OUT
import csv, glob, sys, collections
out = sys.argv[1]
for sub in sorted(glob.glob(out + "/p[0-9]")):
    f = glob.glob(sub + "/**/*counter_collection.csv", recursive=True)
    if not f: print(sub, "no counter csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_tower_chain" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(sub.split("/")[-1], {c: [round(x) for x in v[:2]] for c, v in agg.items()})
PY
