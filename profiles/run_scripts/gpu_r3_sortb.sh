#!/bin/bash
# round 3: bucket + sort of the next batch on the ids branch (WD_SORT_BRANCH) instead of in front of its gather
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r3sortb}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
b() { name=$1; shift; env "$@" timeout 150 python bench.py --no-cpu-baseline --no-pmc --no-parity ${ARGS:---steps 100 --warmup 10} 2> $OUT/$name.err > $OUT/bench_$name.json; python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.4f ms/step  %s" % (sys.argv[2], d["ms_per_step"], d.get("repeats_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
tail -2 $OUT/$name.err | cut -c1-200
}
b branch_gate_tower WD_SORT_GATE=tower
ARGS="--steps 100 --warmup 10 --dist zipf" b zipf_branch_gate_tower WD_SORT_GATE=tower
WD_SORT_GATE=tower timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --dist zipf --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 45 3 > $OUT/c2_zipf_step_timeline.txt; cat $OUT/c2_zipf_step_timeline.txt
rm -rf $OUT/prof
