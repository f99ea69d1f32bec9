#!/bin/bash
# logits-layer partials summed by a column-sum job of the products launch (WD_LOGITS_COLSUM=1) vs by the dense tail (0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4logsum}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_prefetch.py tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/pytest.txt 2>&1; tail -n 2 $OUT/pytest.txt
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
  WD_LOGITS_COLSUM=$m timeout 300 python bench.py --steps 20 --warmup 5 $B > $OUT/c2_ls${m}_$i.json 2>> $OUT/err.txt; line $OUT/c2_ls${m}_$i.json "C2 logits colsum=$m run $i"
done; done
for m in 0 1; do
  WD_LOGITS_COLSUM=$m MASTER_PORT=2957$m timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded $B > $OUT/sh_ls${m}.json 2>> $OUT/err.txt; line $OUT/sh_ls${m}.json "sharded one rank colsum=$m"
  WD_LOGITS_COLSUM=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$m -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --repeats 1 $B > $OUT/prof$m.log 2>&1
  find $OUT/prof$m -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_ls$m.csv
  python scripts/summarize_stats.py $OUT/kernel_stats_ls$m.csv 70 | head -9 | cut -c1-120
  rm -rf $OUT/prof$m
done
tail -n 3 $OUT/err.txt
