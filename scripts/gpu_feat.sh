#!/bin/bash
# featurizer check: every-column bit-exact ids on real rows + C1 end-to-end throughput at batch 512 / 8192
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/feat; mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_c1.py -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest.log
timeout 100 python scripts/bench_c1.py > $OUT/b512.json 2> $OUT/b512.err
C1_BATCH=8192 C1_REPEAT=60 timeout 100 python scripts/bench_c1.py > $OUT/b8192.json 2> $OUT/b8192.err
cat $OUT/pytest.log $OUT/b512.json $OUT/b8192.json; tail -3 $OUT/b512.err
