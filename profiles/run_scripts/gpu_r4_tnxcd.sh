#!/bin/bash
# weight-gradient products: all tiles of a split on one XCD (WD_TN_SPLIT_XCD=1) vs tiles spread (0); same box, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4tnxcd}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_kernels.py -q -m gpu -x > $OUT/pytest.txt 2>&1; tail -n 2 $OUT/pytest.txt
B="--no-cpu-baseline --no-pmc --no-parity"
line() { python - "$@" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.4f ms/step %s" % (sys.argv[2], d["ms_per_step"], d.get("repeats_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for i in 1 2 3; do for m in 0 1; do
  WD_TN_SPLIT_XCD=$m timeout 300 python bench.py --steps 20 --warmup 5 $B > $OUT/c2_xcd${m}_$i.json 2>> $OUT/err.txt; line $OUT/c2_xcd${m}_$i.json "C2 split_xcd=$m run $i"
done; done
for m in 0 1; do
  WD_TN_SPLIT_XCD=$m timeout 300 python bench.py --dist zipf --steps 20 --warmup 5 $B > $OUT/c2z_xcd${m}.json 2>> $OUT/err.txt; line $OUT/c2z_xcd${m}.json "C2 zipf split_xcd=$m"
  WD_TN_SPLIT_XCD=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$m -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --repeats 1 $B > $OUT/prof$m.log 2>&1
  find $OUT/prof$m -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_xcd$m.csv
  python scripts/summarize_stats.py $OUT/kernel_stats_xcd$m.csv 70 | head -9 | cut -c1-120
  rm -rf $OUT/prof$m
done
tail -n 3 $OUT/err.txt
