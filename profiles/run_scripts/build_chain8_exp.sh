#!/bin/sh
# diagnostics: experiment builds of the two-wavefronts-per-SIMD tower kernel (mlp_chain8.hip, -DWD_CHAIN8_EXP=<bits>: 1 no MFMAs,
# 2 no weight loads in the loop, 4 A fragments read once, 8 no HBM stores, 16 two alternating accumulators) as
# wide_deep_amd/_lib/libwd_hip_exp8_<bits>.so; run with WD_HIP_LIB=<that file> python scripts/bench_chain.py
set -e
HERE="$(cd "$(dirname "$0")/../wide_deep_amd/csrc" && pwd)"
OUT="$(cd "$(dirname "$0")/.." && pwd)/wide_deep_amd/_lib"
pids=""
for e in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $WD_EXP_FLAGS -DWD_CHAIN8_EXP=$e -c "$HERE/mlp_chain8.hip" -o "$HERE/_obj/mlp_chain8_exp$e.o" &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
for e in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC "$HERE"/_obj/common.o "$HERE"/_obj/hash.o "$HERE"/_obj/embag.o \
      "$HERE"/_obj/sparse_update.o "$HERE"/_obj/sparse_fused.o "$HERE"/_obj/onehot_path.o "$HERE"/_obj/dist_exchange.o "$HERE"/_obj/mlp.o "$HERE"/_obj/mlp_half.o \
      "$HERE"/_obj/mlp_chain.o "$HERE"/_obj/mlp_chain8_exp$e.o -o "$OUT/libwd_hip_exp8_$e.so"
  echo "built $OUT/libwd_hip_exp8_$e.so"
done
