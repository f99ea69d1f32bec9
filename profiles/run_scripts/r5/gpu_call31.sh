#!/bin/bash
# round 5 call 31: occurrences per row-range bucket (WD_BUCKET_TARGET, default 64) for the ragged update of configs[3]
O=gpurun_out/r5_call31; mkdir -p $O
B="--no-cpu-baseline --no-pmc --no-parity"
for tgt in 32 64 128 256; do
  echo "== target $tgt"; WD_BUCKET_TARGET=$tgt timeout 300 python bench.py --config c4-nocross $B > $O/c4nc_t$tgt.json 2> $O/c4nc_t$tgt.err || tail -3 $O/c4nc_t$tgt.err; cut -c1-200 $O/c4nc_t$tgt.json
done
for tgt in 32 128; do
  echo "== c4 target $tgt"; WD_BUCKET_TARGET=$tgt timeout 300 python bench.py --config c4 $B > $O/c4_t$tgt.json 2> $O/c4_t$tgt.err || tail -3 $O/c4_t$tgt.err; cut -c1-200 $O/c4_t$tgt.json
done
