#!/bin/bash
# round 5, call 7: products v2 fixed (operand halves): test; in-step timelines by slices x issue priority
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call7; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_chain.py -q -m gpu -x > $OUT/pytest.txt 2>&1; tail -n 3 $OUT/pytest.txt
for v in "WD_TN_SPLIT=6" "WD_TN_SPLIT=13" "WD_TN_STREAM=0 WD_TN_SPLIT=16"; do
  env $v timeout 120 python scripts/bench_tn.py 2>&1 | grep "^products" | sed "s/^/[$v] /"
done | tee $OUT/tn_alone.txt
i=0
for v in "WD_TN_SPLIT=13" "WD_TN_SPLIT=13 WD_TNS_PRIO=3" "WD_TN_SPLIT=6" "WD_TN_SPLIT=6 WD_TNS_PRIO=3" "WD_TN_SPLIT=6 WD_TNS_PRIO=1" "WD_TN_STREAM=0 WD_TN_SPLIT=16"; do
  i=$((i+1))
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 40 --warmup 10 --pool 16 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof$i.log 2>&1
  T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
  echo "== $v"; grep -o '"ms_per_step": [0-9.]*' $OUT/prof$i.log | head -1
  python scripts/trace_window.py $T k_tower_chain 30 1 > $OUT/timeline$i.txt; grep -v "hash_bucket\|bucket_\|prefetch" $OUT/timeline$i.txt | head -8
  rm -rf $OUT/prof
done 2>&1 | tee $OUT/coupling.txt
B="--no-cpu-baseline --no-pmc --no-parity --steps 20 --warmup 5 --repeats 9"
for v in "WD_TN_SPLIT=13 WD_TNS_PRIO=3" "WD_TN_SPLIT=6 WD_TNS_PRIO=3" "WD_TN_STREAM=0 WD_TN_SPLIT=16" "WD_TN_SPLIT=13" ; do
  env $v timeout 200 python bench.py $B > $OUT/b.json 2>> $OUT/bench.err
  python - "$v" $OUT/b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("[%s] %.4f ms/step %s" % (sys.argv[1], d["ms_per_step"], d.get("repeats_ms_per_step")))
PY
done | tee $OUT/step_ab.txt
