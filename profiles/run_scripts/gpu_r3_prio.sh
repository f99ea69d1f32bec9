#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r3prio}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
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
b base X=1
b sort_prio WD_SORT_PRIO=1
ARGS="--steps 100 --warmup 10 --dist zipf" b zipf_base X=1
ARGS="--steps 100 --warmup 10 --dist zipf" b zipf_sort_prio WD_SORT_PRIO=1
