#!/bin/bash
# round 5 call 36: GPU_MAX_HW_QUEUES 3 vs 4, alternating, C2 driver arguments / 200 steps / Zipf / C3 / sharded one-rank
O=gpurun_out/r5_call36; mkdir -p $O
B="--no-cpu-baseline --no-pmc --no-parity"
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("   ", d["ms_per_step"], d["ms_per_step_min_median_max"])
except Exception as e:
    print("    FAILED", e)
PY
}
for q in 3 4 3 4 3 4; do
  echo "== C2 driver args, $q queues"; GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > $O/c2_q${q}_$RANDOM.json 2> $O/err.txt || tail -3 $O/err.txt; show $(ls -t $O/c2_q${q}_*.json | head -1)
done
for q in 3 4; do
  echo "== C2 200 steps, $q queues"; GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py $B > $O/c2long_q$q.json 2> $O/err.txt || tail -3 $O/err.txt; show $O/c2long_q$q.json
  echo "== C2 zipf, $q queues"; GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --dist zipf --steps 20 --warmup 5 $B > $O/zipf_q$q.json 2> $O/err.txt || tail -3 $O/err.txt; show $O/zipf_q$q.json
  echo "== sharded one-rank, $q queues"; MASTER_PORT=2959$q GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded $B > $O/sh_q$q.json 2> $O/err.txt || tail -3 $O/err.txt; show $O/sh_q$q.json
done
