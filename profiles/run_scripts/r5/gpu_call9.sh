#!/bin/bash
# round 5, call 9: how much of the products || row-update segment is the update's VALU work (optimizer arithmetic compiled to one fma)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call9; mkdir -p $OUT
i=0
for v in "WD_TN_SPLIT=13" "WD_TN_SPLIT=13 WD_UPD_EXP=1" "WD_TN_STREAM=0 WD_TN_SPLIT=16" "WD_TN_STREAM=0 WD_TN_SPLIT=16 WD_UPD_EXP=1" "WD_TN_SPLIT=13 WD_TNS_EXP=3" "WD_TN_SPLIT=13 WD_TNS_EXP=3 WD_UPD_EXP=1"; do
  i=$((i+1))
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 40 --warmup 10 --pool 16 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof$i.log 2>&1
  T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
  echo "== $v"; grep -o '"ms_per_step": [0-9.]*' $OUT/prof$i.log | head -1
  python scripts/trace_window.py $T k_tower_chain 30 1 > $OUT/timeline$i.txt; grep -v "hash_bucket\|bucket_\|prefetch" $OUT/timeline$i.txt | head -6 | tail -4
  rm -rf $OUT/prof
done 2>&1 | tee $OUT/coupling.txt
