#!/bin/bash
# round 5, call 4: pure kernel durations (rocprofv3 kernel trace) of the streamed products and their parts
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call4; mkdir -p $OUT
i=0
for v in "WD_TN_SPLIT=6 WD_TNS_LDS=51200" "WD_TN_SPLIT=6 WD_TNS_LDS=51200 WD_TNS_EXP=2" "WD_TN_SPLIT=6 WD_TNS_LDS=51200 WD_TNS_EXP=3" "WD_TN_SPLIT=6 WD_TNS_LDS=51200 WD_TNS_EXP=7" \
         "WD_TN_SPLIT=6 WD_TNS_LDS=51200 WD_TNS_EXP=15" "WD_TN_SPLIT=6 WD_TNS_LDS=51200 WD_TNS_EXP=4" "WD_TN_SPLIT=13 WD_TNS_LDS=24576" "WD_TN_SPLIT=13 WD_TNS_LDS=24576 WD_TNS_EXP=4" "WD_TN_STREAM=0 WD_TN_SPLIT=16"; do
  i=$((i+1))
  env $v CHAIN_ITERS=50 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$i -o t -- python scripts/bench_tn.py > $OUT/log$i.txt 2>&1
  grep "^products" $OUT/log$i.txt | sed "s/^/[$v] /"
  S=$(find $OUT/p$i -name "*kernel_stats*.csv" | head -1)
  grep -i "k_tn_stream\|k_gemm_tn_group" $S | cut -c1-160
  rm -rf $OUT/p$i
done | tee $OUT/tn_rocprof.txt
