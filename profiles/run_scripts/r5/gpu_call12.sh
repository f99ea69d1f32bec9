#!/bin/bash
# round 5, call 12: the LDS-tiled products with 35 KB of LDS per workgroup (one instantiation of the body) against the 70 KB of rounds 2-4
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call12; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_chain.py -q -m gpu -x > $OUT/pytest.txt 2>&1; tail -n 3 $OUT/pytest.txt
for v in "WD_TN_LDS=0" "WD_TN_LDS=34816" "WD_TN_LDS=0 WD_TN_SPLIT_CAP=12" "WD_TN_LDS=0 WD_TN_SPLIT_CAP=24"; do
  env $v timeout 120 python scripts/bench_tn.py 2>&1 | grep "^products" | sed "s/^/[$v] /"
done | tee $OUT/tn_alone.txt
i=0
for v in "WD_TN_LDS=0" "WD_TN_LDS=34816" "WD_TN_LDS=16000"; do
  i=$((i+1))
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 40 --warmup 10 --pool 16 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof$i.log 2>&1
  T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
  echo "== $v"; grep -o '"ms_per_step": [0-9.]*' $OUT/prof$i.log | head -1
  python scripts/trace_window.py $T k_tower_chain 30 1 > $OUT/timeline$i.txt; grep -v "hash_bucket\|bucket_\|prefetch" $OUT/timeline$i.txt | head -6 | tail -4
  rm -rf $OUT/prof
done 2>&1 | tee $OUT/lds.txt
B="--no-cpu-baseline --no-pmc --no-parity --steps 20 --warmup 5 --repeats 9"
for v in "WD_TN_LDS=0" "WD_TN_LDS=34816" "WD_TN_LDS=0" "WD_TN_LDS=34816" "WD_TN_LDS=0 WD_TN_SPLIT_CAP=12" "WD_TN_LDS=0 WD_TN_SPLIT_CAP=20"; do
  env $v timeout 200 python bench.py $B > $OUT/b.json 2>> $OUT/bench.err
  python - "$v" $OUT/b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("[%s] %.4f ms/step %s" % (sys.argv[1], d["ms_per_step"], d.get("repeats_ms_per_step")))
PY
done | tee $OUT/step_ab.txt
