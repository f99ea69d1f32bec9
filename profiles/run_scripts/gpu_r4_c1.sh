#!/bin/bash
# C1 (repo-default conf, real rows) end to end: batch t+1 featurized on a worker thread while batch t steps (WD_FEATURIZE_AHEAD)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4c1}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_c1.py -q -m gpu -x > $OUT/pytest_c1.txt 2>&1; tail -n 3 $OUT/pytest_c1.txt
for bs in 512 64; do for a in 0 1 0 1; do
  C1_BATCH=$bs C1_REPEAT=20 WD_FEATURIZE_AHEAD=$a timeout 300 python scripts/bench_c1.py > $OUT/c1_b${bs}_ahead$a.json 2>> $OUT/err.txt
  python - $OUT/c1_b${bs}_ahead$a.json $bs $a <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("batch %s ahead=%s: loop %.0f ex/s | parse %.0f featurize %.0f step-only %.0f" % (sys.argv[2], sys.argv[3], d["train_loop_examples_per_sec"], d["host_parse_rows_per_sec"], d["gpu_featurize_rows_per_sec"], d["train_step_only_examples_per_sec"]))
PY
done; done
tail -n 3 $OUT/err.txt
