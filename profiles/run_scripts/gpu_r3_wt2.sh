#!/bin/bash
# round 3: which kernels gain from write-through stores (WD_WT mask: 1 tower, 2 products, 4 row update, 8 prefetch); same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r3wt2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_prefetch.py -m gpu -x -q 2>&1 | tail -3
WD_WT=15 timeout 300 python -m pytest tests/test_gpu_prefetch.py tests/test_gpu_fused_tail.py -m gpu -x -q 2>&1 | tail -3
b() { name=$1; shift; env "$@" timeout 150 python bench.py --no-cpu-baseline --no-pmc --no-parity ${ARGS:---steps 100 --warmup 10} 2> $OUT/$name.err > $OUT/bench_$name.json; python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.4f ms/step  %s" % (sys.argv[2], d["ms_per_step"], d.get("repeats_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for m in 0 1 3 5 9 7 15 1; do b wt_$m WD_WT=$m; done
ARGS="--steps 100 --warmup 10 --dist zipf" b zipf_wt_1 WD_WT=1
ARGS="--steps 100 --warmup 10 --dist zipf" b zipf_wt_15 WD_WT=15
ARGS="--steps 100 --warmup 10 --dist zipf" b zipf_wt_5 WD_WT=5
