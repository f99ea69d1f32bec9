#!/bin/bash
# featurizer / input pipeline check: distributed estimator test, C1 end-to-end throughput at batch 8192 / 512 with and
# without the parse-ahead thread (files long enough that the loop time is not dominated by the first batches)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/feat; mkdir -p $OUT
timeout 100 python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k estimator 2>&1 | tail -4 > $OUT/pytest.log
C1_BATCH=8192 C1_REPEAT=400 timeout 60 python scripts/bench_c1.py > $OUT/b8192.json 2> $OUT/b8192.err
C1_BATCH=8192 C1_REPEAT=400 WD_PREFETCH=0 timeout 60 python scripts/bench_c1.py > $OUT/b8192_nopf.json 2> $OUT/b8192_nopf.err
C1_REPEAT=100 timeout 60 python scripts/bench_c1.py > $OUT/b512.json 2> $OUT/b512.err
C1_REPEAT=100 WD_PREFETCH=0 timeout 60 python scripts/bench_c1.py > $OUT/b512_nopf.json 2> $OUT/b512_nopf.err
cat $OUT/pytest.log; for f in b8192 b8192_nopf b512 b512_nopf; do echo $f; cut -c1-330 $OUT/$f.json; done
