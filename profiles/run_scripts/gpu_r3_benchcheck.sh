#!/bin/bash
# the three graph paths of bench.py after the guarded fallback: single GPU, captured collectives (one-rank RCCL), graph segments (gloo, 2 ranks)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3check; mkdir -p $OUT
B="--no-cpu-baseline --no-pmc --no-parity"
timeout 120 python bench.py --steps 20 --warmup 5 $B 2> $OUT/a.err | cut -c1-160; tail -1 $OUT/a.err | cut -c1-200
MASTER_PORT=29581 timeout 120 python bench.py --steps 20 --warmup 5 --force-sharded $B 2> $OUT/b.err | cut -c1-160; tail -1 $OUT/b.err | cut -c1-200
WD_DIST_BACKEND=gloo timeout 200 python bench.py --gpus 2 --steps 4 --warmup 1 --pool 2 --repeats 1 $B 2> $OUT/c.err | cut -c1-160; tail -1 $OUT/c.err | cut -c1-200
