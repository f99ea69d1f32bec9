#!/bin/bash
# round 3: pipelined tile tail of wd_gemm_tn_group_tail, graph-launch cost probe, gaps at graph boundaries
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r3fuse2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_fused_tail.py -m gpu -x -q 2>&1 | tail -5
b() { name=$1; shift; env "$@" timeout 150 python bench.py --no-cpu-baseline --no-pmc ${ARGS:---steps 20 --warmup 5} 2> $OUT/$name.err > $OUT/bench_$name.json; python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.4f ms/step  %s" % (sys.argv[2], d["ms_per_step"], d.get("repeats_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
b default X=1
b no_fuse WD_FUSE_TAIL=0
ARGS="--steps 200 --warmup 20" b default_200 X=1
timeout 200 python scripts/bench_graph_launch.py 2> $OUT/launch.err | tee $OUT/graph_launch.txt; tail -2 $OUT/launch.err
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/graph_gaps.py $T 64 > $OUT/graph_gaps.txt; head -c 6000 $OUT/graph_gaps.txt
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/c2_step_timeline.txt; cat $OUT/c2_step_timeline.txt
rm -rf $OUT/prof
