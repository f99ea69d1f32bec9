#!/bin/bash
# round 5, call 6: what couples the products and the row update in the step: timelines with the products' MFMAs / loads compiled out
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call6; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k grouped > $OUT/pytest.txt 2>&1; tail -n 25 $OUT/pytest.txt
i=0
for v in "WD_TN_SPLIT=13" "WD_TN_SPLIT=13 WD_TNS_EXP=1" "WD_TN_SPLIT=13 WD_TNS_EXP=2" "WD_TN_SPLIT=13 WD_TNS_EXP=3" "WD_TN_SPLIT=6 WD_TNS_EXP=2" "WD_TN_SPLIT=6 WD_TNS_EXP=1"; do
  i=$((i+1))
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 40 --warmup 10 --pool 16 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof$i.log 2>&1
  T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
  echo "== $v"; grep -o '"ms_per_step": [0-9.]*' $OUT/prof$i.log | head -1
  python scripts/trace_window.py $T k_tower_chain 30 1 > $OUT/timeline$i.txt; grep -v "hash_bucket\|bucket_\|prefetch" $OUT/timeline$i.txt | head -8
  rm -rf $OUT/prof
done 2>&1 | tee $OUT/coupling.txt
