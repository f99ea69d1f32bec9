#!/bin/bash
# round 5 call 34: chunk count of the ragged bucketing by occurrences -- sharded owner list (213 k) at 128 vs 512 chunks, configs[3] unchanged
O=gpurun_out/r5_call34; mkdir -p $O
B="--no-cpu-baseline --no-pmc --no-parity"
p=29570
for c in 128 512 128 512; do
  p=$((p+1)); echo "== sharded one-rank, chunks $c"; MASTER_PORT=$p WD_BUCKET_CHUNKS=$c timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded $B > $O/sh_$c.json 2> $O/sh_$c.err || tail -3 $O/sh_$c.err; cut -c1-330 $O/sh_$c.json
done
echo "== sharded default"; MASTER_PORT=29580 timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded $B > $O/sh_default.json 2> $O/sh_default.err; cut -c1-200 $O/sh_default.json
echo "== c4 default"; timeout 300 python bench.py --config c4 $B > $O/c4.json 2> $O/c4.err; cut -c1-200 $O/c4.json
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_c4.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
