#!/bin/bash
# round 5 call 32: up to 16384 row-range buckets on the ragged path
O=gpurun_out/r5_call32; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_c4.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu -k "c4 or small or multi or fuzz" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="--no-cpu-baseline --no-pmc --no-parity"
for tgt in 64 32; do
  echo "== nocross target $tgt"; WD_BUCKET_TARGET=$tgt timeout 300 python bench.py --config c4-nocross $B > $O/c4nc_t$tgt.json 2> $O/c4nc_t$tgt.err || tail -3 $O/c4nc_t$tgt.err; cut -c1-200 $O/c4nc_t$tgt.json
  echo "== c4 target $tgt"; WD_BUCKET_TARGET=$tgt timeout 300 python bench.py --config c4 $B > $O/c4_t$tgt.json 2> $O/c4_t$tgt.err || tail -3 $O/c4_t$tgt.err; cut -c1-200 $O/c4_t$tgt.json
done
