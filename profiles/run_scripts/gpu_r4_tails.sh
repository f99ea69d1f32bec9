#!/bin/bash
# serialised tails of the critical-path kernels: bias_weights workgroup of k_row_update (first in the grid, 8 loads in flight),
# column-sum jobs first in the grouped products launch (16 loads in flight); A/B against the previous kernels (libwd_hip_prev.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4tails}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_prefetch.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/pytest.txt 2>&1; tail -n 2 $OUT/pytest.txt
B="--no-cpu-baseline --no-pmc --no-parity"
PREV=$PWD/wide_deep_amd/_lib/libwd_hip_prev.so
line() { python - "$@" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.4f ms/step %s" % (sys.argv[2], d["ms_per_step"], d.get("repeats_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for i in 1 2 3; do
  WD_HIP_LIB=$PREV timeout 300 python bench.py --steps 20 --warmup 5 $B > $OUT/c2_prev_$i.json 2>> $OUT/err.txt; line $OUT/c2_prev_$i.json "C2 previous kernels run $i"
  timeout 300 python bench.py --steps 20 --warmup 5 $B > $OUT/c2_new_$i.json 2>> $OUT/err.txt; line $OUT/c2_new_$i.json "C2 new run $i"
done
WD_HIP_LIB=$PREV timeout 300 python bench.py --dist zipf --steps 20 --warmup 5 $B > $OUT/c2z_prev.json 2>> $OUT/err.txt; line $OUT/c2z_prev.json "C2 zipf previous"
timeout 300 python bench.py --dist zipf --steps 20 --warmup 5 $B > $OUT/c2z_new.json 2>> $OUT/err.txt; line $OUT/c2z_new.json "C2 zipf new"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --repeats 1 $B > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/step_timeline.txt; grep -v hash_bucket $OUT/step_timeline.txt | cut -c1-110
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python scripts/summarize_stats.py $OUT/kernel_stats.csv 70 | head -10 | cut -c1-120
rm -rf $OUT/prof
tail -n 3 $OUT/err.txt
