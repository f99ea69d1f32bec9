#!/bin/bash
# round 5 call 35: GPU_MAX_HW_QUEUES 2 / 3 / 4 (default) / 6 under the chained C2 graphs (8 made them 2x slower, call 26)
O=gpurun_out/r5_call35; mkdir -p $O
B="--no-cpu-baseline --no-pmc --no-parity"
for q in 4 2 3 6 4; do
  echo "== C2 driver args, $q hardware queues"; GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $O/c2_q$q.json 2> $O/c2_q$q.err || tail -3 $O/c2_q$q.err; cut -c1-330 $O/c2_q$q.json | cut -c150-330
done
for q in 2 3; do
  echo "== c4, $q hardware queues"; GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --config c4 $B > $O/c4_q$q.json 2> $O/c4_q$q.err || tail -3 $O/c4_q$q.err; cut -c150-330 $O/c4_q$q.json
done
