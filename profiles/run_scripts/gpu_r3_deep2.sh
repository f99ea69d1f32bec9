#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r3deep2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
WD_TN_DEEP=1 timeout 300 python -m pytest tests/test_gpu_fused_tail.py -m gpu -x -q 2>&1 | tail -3
b() { name=$1; shift; env "$@" timeout 150 python bench.py --no-cpu-baseline --no-pmc --no-parity ${ARGS:---steps 100 --warmup 10} 2> $OUT/$name.err > $OUT/bench_$name.json; python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.4f ms/step  %s" % (sys.argv[2], d["ms_per_step"], d.get("repeats_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
b base X=1
b fuse WD_FUSE_TAIL=1
b fuse_deep WD_FUSE_TAIL=1 WD_TN_DEEP=1
b base_again X=1
b fuse_deep_again WD_FUSE_TAIL=1 WD_TN_DEEP=1
WD_FUSE_TAIL=1 WD_TN_DEEP=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/c2_step_timeline.txt; cat $OUT/c2_step_timeline.txt
rm -rf $OUT/prof
