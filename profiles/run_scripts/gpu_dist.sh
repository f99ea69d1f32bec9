#!/bin/bash
# sharded engine: world-2 parity tests (gloo staging on one GPU) + the N=2 bench smoke in that mode, eager vs segments
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/dist; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -15
for G in "" "--no-graph"; do
  WD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 30 --warmup 5 --pool 4 $G 2> $OUT/err$G.log | cut -c1-250
done
tail -3 $OUT/err.log
