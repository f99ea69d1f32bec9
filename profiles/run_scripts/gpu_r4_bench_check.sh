#!/bin/bash
# round 4: the restructured bench (pipeline.StepRunner, in-step roofline on the timed graphs, N > 1 line + launcher fallback)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4i}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "graphs_bench_times" > $OUT/pytest_runner.txt 2>&1; tail -n 4 $OUT/pytest_runner.txt
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu -x -k "bench_two_ranks or dedup or baseline_size" > $OUT/pytest_dist.txt 2>&1; tail -n 4 $OUT/pytest_dist.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-pmc --cpu-steps 10 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 2500 $OUT/bench_default.json; tail -n 3 $OUT/bench_default.err
MASTER_PORT=29571 timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded --no-pmc --no-cpu-baseline > $OUT/bench_sharded1.json 2> $OUT/bench_sharded1.err; python - $OUT/bench_sharded1.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("sharded one rank: %.4f ms/step" % d["ms_per_step"], json.dumps(d["config"]["exchange"])[:1500]); print("parity", d.get("parity")); print("roofline", json.dumps(d.get("roofline"))[:600])
except Exception as e:
    print("FAILED", e)
PY
tail -n 3 $OUT/bench_sharded1.err
