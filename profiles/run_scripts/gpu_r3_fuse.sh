#!/bin/bash
# round 3, after the evidence run: (1) dense tail inside the weight-gradient launch (wd_gemm_tn_group_tail) and (2) multi-step
# graphs chained through lookahead / primed -- parity tests, then A/B bench lines on ONE box and a step timeline of the default
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r3fuse}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 420 python -m pytest tests/test_gpu_fused_tail.py tests/test_gpu_prefetch.py -m gpu -x -q 2>&1 | tail -15
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
b no_chain WD_GRAPH_CHAIN=0
b no_fuse WD_FUSE_TAIL=0
ARGS="--steps 200 --warmup 20" b default_200 X=1
ARGS="--steps 20 --warmup 5 --dist zipf" b zipf X=1
ARGS="--steps 20 --warmup 5 --dist zipf" b zipf_round2_step WD_INPUT_AHEAD=0
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/c2_step_timeline.txt; cat $OUT/c2_step_timeline.txt
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c2_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c2_kernel_stats.csv 104 > $OUT/c2_kernel_stats.md 2>/dev/null; head -30 $OUT/c2_kernel_stats.md
rm -rf $OUT/prof
