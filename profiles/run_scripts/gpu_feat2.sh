#!/bin/bash
# device featurizer: id parity tests (device vs host vs oracle), C1 conf tests, C1 loop throughput device vs host featurizer
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-feat2}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_c1.py tests/test_gpu_c4.py -x -q > $OUT/pytest.log 2>&1; tail -n 15 $OUT/pytest.log
for mode in device host; do
  WD_FEATURIZER=$mode C1_BATCH=8192 C1_REPEAT=400 timeout 90 python scripts/bench_c1.py > $OUT/b8192_$mode.json 2> $OUT/b8192_$mode.err
  WD_FEATURIZER=$mode C1_REPEAT=100 timeout 90 python scripts/bench_c1.py > $OUT/b512_$mode.json 2> $OUT/b512_$mode.err
done
for f in b8192_device b8192_host b512_device b512_host; do echo $f; cut -c1-400 $OUT/$f.json; tail -n 2 $OUT/$f.err; done
