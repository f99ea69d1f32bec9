#!/bin/bash
# round 5, call 2: why the streamed products are slow: workgroups per CU (LDS padding), loads only, MFMAs only
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call3; mkdir -p $OUT
for v in "WD_TN_SPLIT=6" "WD_TN_SPLIT=6 WD_TNS_LDS=51200" "WD_TN_SPLIT=13" "WD_TN_SPLIT=13 WD_TNS_LDS=24576" "WD_TN_SPLIT=13 WD_TNS_LDS=51200" \
         "WD_TN_SPLIT=6 WD_TNS_LDS=51200 WD_TNS_EXP=1" "WD_TN_SPLIT=6 WD_TNS_LDS=51200 WD_TNS_EXP=2" "WD_TN_SPLIT=6 WD_TNS_LDS=51200 WD_TNS_EXP=3" \
         "WD_TN_SPLIT=13 WD_TNS_LDS=24576 WD_TNS_EXP=1" "WD_TN_SPLIT=13 WD_TNS_LDS=24576 WD_TNS_EXP=2" \
         "WD_TN_SPLIT=8 WD_TNS_LDS=51200" "WD_TN_SPLIT=7 WD_TNS_LDS=51200" "WD_TN_SPLIT=12 WD_TNS_LDS=24576" "WD_TN_STREAM=0 WD_TN_SPLIT=16"; do
  env $v timeout 120 python scripts/bench_tn.py 2>&1 | grep "^products" | sed "s/^/[$v] /"
done | tee $OUT/tn_alone.txt
