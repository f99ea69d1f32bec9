#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4n}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_prefetch.py -q -m gpu -x > $OUT/pytest.txt 2>&1; tail -n 2 $OUT/pytest.txt
bash scripts/gpu_r4_ab2.sh $TAG - -
B="--no-cpu-baseline --no-pmc --no-parity --steps 20 --warmup 5"
for D in uniform zipf; do
MASTER_PORT=2957$RANDOM timeout 300 python bench.py $B --force-sharded --dist $D > $OUT/sh_$D.json 2>> $OUT/sh.err
python - $OUT/sh_$D.json $D <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("sharded one rank %-8s %.4f ms/step %s" % (sys.argv[2], d["ms_per_step"], d["repeats_ms_per_step"]))
except Exception as e: print("FAILED", e)
PY
done
bash scripts/gpu_r4_ab2.sh ${TAG}_z "-" 
