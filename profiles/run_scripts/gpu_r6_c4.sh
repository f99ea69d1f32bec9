#!/bin/bash
# round 6: configs[3] work loop -- the tests that touch its kernels, the line from tokens (and optionally A/B environments), kernel table + timeline
# usage: gpu_r6_c4.sh TAG [notest] ["ENV=.. ENV=.." ...]   (every extra argument = one more bench line under that environment)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r6c4}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; shift
B="--no-cpu-baseline --no-pmc"
if [ "$1" = "notest" ]; then shift; else
timeout 1500 python -m pytest tests/test_gpu_c4.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -q -m gpu -x -k "c4 or small or cross or ragged" > $OUT/pytest_c4.txt 2>&1; tail -n 5 $OUT/pytest_c4.txt
fi
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print("%-40s %.4f ms/step %s parity %s" % (sys.argv[1], d["ms_per_step"], d.get("repeats_ms_per_step"), (d.get("parity") or {}).get("max_abs_dlogit")))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
timeout 400 python bench.py --config c4 $B --steps 60 > $OUT/bench_c4_tokens.json 2> $OUT/bench.err; line c4_tokens $OUT/bench_c4_tokens.json; tail -n 2 $OUT/bench.err
k=0
for ENVS in "$@"; do k=$((k+1))
  env $ENVS timeout 400 python bench.py --config c4 $B --steps 60 --no-parity > $OUT/bench_c4_ab$k.json 2>> $OUT/bench.err; line "c4 [$ENVS]" $OUT/bench_c4_ab$k.json
done
timeout 300 python bench.py --config c4-nocross $B --steps 100 --no-parity > $OUT/bench_c4_nocross.json 2>> $OUT/bench.err; line c4_nocross $OUT/bench_c4_nocross.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --config c4 --steps 60 --warmup 10 --pool 16 --repeats 1 $B --no-parity > $OUT/prof_c4.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c4_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c4_kernel_stats.csv 75 > $OUT/c4_kernel_stats.md; head -24 $OUT/c4_kernel_stats.md | cut -c1-110
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 40 1 > $OUT/c4_step_timeline.txt; cat $OUT/c4_step_timeline.txt
rm -rf $OUT/prof
