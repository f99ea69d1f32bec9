#!/bin/bash
# round 3: sender-side unique of the sharded row exchange -- parity (world 2 / 4, graph segments, BASELINE size with Zipf ids),
# then the one-rank RCCL step with Zipf ids with / without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r3dedup}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "dedup or chain_pack or (world2_equals and chain)" 2>&1 | tail -15
b() { name=$1; shift; env "$@" timeout 200 python bench.py --force-sharded --no-cpu-baseline --no-pmc ${ARGS:---steps 100 --warmup 10 --pool 8} 2> $OUT/$name.err > $OUT/bench_$name.json; python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    x = d["config"]["exchange"]
    print("%-28s %.4f ms/step  %s  cap %d unique %s overflow %s" % (sys.argv[2], d["ms_per_step"], d.get("repeats_ms_per_step"), x["segment_capacity"], x.get("sender_side_unique"), x["check_overflow"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
tail -2 $OUT/$name.err | cut -c1-300
}
ARGS="--steps 100 --warmup 10 --pool 8 --dist zipf" MASTER_PORT=29571 b zipf_dedup X=1
ARGS="--steps 100 --warmup 10 --pool 8 --dist zipf" MASTER_PORT=29572 b zipf_plain WD_SHARD_DEDUP=0
MASTER_PORT=29573 b uniform_auto X=1
MASTER_PORT=29574 b uniform_dedup WD_SHARD_DEDUP=1
