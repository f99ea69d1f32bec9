#!/bin/bash
# round 5, call 5: products kernel v2 (buffer loads, compact arguments, three register sets): test, alone, in the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call5; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_chain.py -q -m gpu -x > $OUT/pytest.txt 2>&1; tail -n 5 $OUT/pytest.txt
for v in "WD_TN_SPLIT=6" "WD_TN_SPLIT=13" "WD_TN_SPLIT=12" "WD_TN_SPLIT=7" "WD_TN_SPLIT=5" "WD_TN_SPLIT=6 WD_TNS_LDS=51200" "WD_TN_STREAM=0 WD_TN_SPLIT=16"; do
  env $v timeout 120 python scripts/bench_tn.py 2>&1 | grep "^products" | sed "s/^/[$v] /"
done | tee $OUT/tn_alone.txt
B="--no-cpu-baseline --no-pmc --no-parity --steps 20 --warmup 5 --repeats 9"
for v in "WD_TN_SPLIT=6" "WD_TN_SPLIT=13" "WD_TN_STREAM=0 WD_TN_SPLIT=16" "WD_TN_SPLIT=6" "WD_TN_STREAM=0 WD_TN_SPLIT=16"; do
  env $v timeout 200 python bench.py $B > $OUT/b.json 2>> $OUT/bench.err
  python - "$v" $OUT/b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("[%s] %.4f ms/step %s" % (sys.argv[1], d["ms_per_step"], d.get("repeats_ms_per_step")))
PY
done | tee $OUT/step_ab.txt
WD_TN_SPLIT=6 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/c2_step_timeline.txt; grep -v hash_bucket $OUT/c2_step_timeline.txt | head -24
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c2_uniform_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c2_uniform_kernel_stats.csv 70 > $OUT/c2_uniform_kernel_stats.md; head -12 $OUT/c2_uniform_kernel_stats.md
rm -rf $OUT/prof
tail -n 3 $OUT/bench.err
