#!/bin/bash
# split-K count of the weight-gradient products (WD_TN_SPLIT_CAP) on the C2 step; WD_OVERLAP on C4 / C5; connection-list tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4split}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_step.py -q -m gpu -x > $OUT/pytest_step.txt 2>&1; tail -n 3 $OUT/pytest_step.txt
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
for cap in 16 13 12 10 8 20 16; do
  WD_TN_SPLIT_CAP=$cap timeout 300 python bench.py --steps 20 --warmup 5 $B > $OUT/c2_cap$cap.json 2>> $OUT/err.txt; line $OUT/c2_cap$cap.json "C2 split cap $cap"
done
for ov in none bucket both; do
  WD_OVERLAP=$ov timeout 300 python bench.py --config c4 --steps 60 $B > $OUT/c4_$ov.json 2>> $OUT/err.txt; line $OUT/c4_$ov.json "C4 WD_OVERLAP=$ov"
done
for ov in tn both; do
  WD_OVERLAP=$ov timeout 300 python bench.py --config c5 --steps 40 $B > $OUT/c5_$ov.json 2>> $OUT/err.txt; line $OUT/c5_$ov.json "C5 WD_OVERLAP=$ov"
done
tail -n 5 $OUT/err.txt
