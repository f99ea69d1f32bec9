#!/bin/bash
# round 5, call 10: the row update against its own occupancy (workgroups per CU limited by LDS padding), products compiled empty
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call10; mkdir -p $OUT
i=0
for v in "WD_UPD_LDS=0" "WD_UPD_LDS=18000" "WD_UPD_LDS=32000" "WD_UPD_LDS=45000" "WD_UPD_LDS=72000"; do
  i=$((i+1))
  env WD_TN_SPLIT=13 WD_TNS_EXP=3 $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 40 --warmup 10 --pool 16 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof$i.log 2>&1
  T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
  echo "== $v (workgroups per CU <= $(( 163840 / (7760 + ${v#*=}) )))"; grep -o '"ms_per_step": [0-9.]*' $OUT/prof$i.log | head -1
  python scripts/trace_window.py $T k_tower_chain 30 1 > $OUT/timeline$i.txt; grep "row_update" $OUT/timeline$i.txt | head -2
  rm -rf $OUT/prof
done 2>&1 | tee $OUT/occupancy.txt
