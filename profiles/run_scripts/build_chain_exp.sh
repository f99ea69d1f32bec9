#!/bin/sh
# diagnostics: experiment builds of the tower kernel (WD_CHAIN_EXP = 1 no MFMAs / 2 weights loaded once / 3 A fragments read
# once) as gpurun_out/libwd_hip_exp<N>.so; run with WD_HIP_LIB=<that file> python scripts/bench_chain.py
set -e
HERE="$(cd "$(dirname "$0")/../wide_deep_amd/csrc" && pwd)"
OUT="$(cd "$(dirname "$0")/.." && pwd)/wide_deep_amd/_lib"
for e in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DWD_CHAIN_EXP=$e -c "$HERE/mlp_chain.hip" -o "$HERE/_obj/mlp_chain_exp$e.o"
  hipcc --offload-arch=gfx950 -shared -fPIC "$HERE"/_obj/common.o "$HERE"/_obj/hash.o "$HERE"/_obj/embag.o \
      "$HERE"/_obj/sparse_update.o "$HERE"/_obj/sparse_fused.o "$HERE"/_obj/dist_exchange.o "$HERE"/_obj/mlp.o "$HERE"/_obj/mlp_half.o \
      "$HERE"/_obj/mlp_chain_exp$e.o -o "$OUT/libwd_hip_exp$e.so"
  echo "built $OUT/libwd_hip_exp$e.so"
done
