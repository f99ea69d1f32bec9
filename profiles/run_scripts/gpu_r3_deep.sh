#!/bin/bash
# round 3: weight-gradient products with the global loads two slabs ahead (WD_TN_DEEP), tower wavefront priority (WD_CHAIN_FLAGS=8)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r3deep}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_fused_tail.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -3
b() { name=$1; shift; env "$@" timeout 150 python bench.py --no-cpu-baseline --no-pmc --no-parity ${ARGS:---steps 100 --warmup 10} 2> $OUT/$name.err > $OUT/bench_$name.json; python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.4f ms/step  %s" % (sys.argv[2], d["ms_per_step"], d.get("repeats_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
b deep1 WD_TN_DEEP=1
b deep0 WD_TN_DEEP=0
b deep1_prio WD_TN_DEEP=1 WD_CHAIN_FLAGS=8
b deep1_again WD_TN_DEEP=1
b deep0_again WD_TN_DEEP=0
ARGS="--steps 100 --warmup 10 --dist zipf" b zipf_deep1 WD_TN_DEEP=1
ARGS="--steps 100 --warmup 10 --dist zipf" b zipf_deep0 WD_TN_DEEP=0
ARGS="--steps 40 --warmup 5 --config c4" b c4 X=1
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/c2_step_timeline.txt; cat $OUT/c2_step_timeline.txt
rm -rf $OUT/prof
