#!/bin/sh
# diagnostics: ablation builds of the featurizer's emit kernel (hash.hip, -DFE_EXP=<bits>: 1 stop behind the LDS tables, 2 stop behind
# the prefix table, 4 ids without the hashing) as wide_deep_amd/_lib/libwd_hip_fe_<bits>.so; run with
# WD_HIP_LIB=<that file> python scripts/bench_featurizer.py
set -e
HERE="$(cd "$(dirname "$0")/../../wide_deep_amd/csrc" && pwd)"
OUT="$HERE/../_lib"
for e in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFE_EXP=$e -c "$HERE/hash.hip" -o "$HERE/_obj/hash_fe$e.o"
  objs=""
  for f in build_stamp common embag sparse_update sparse_fused small_tables onehot_path dist_exchange mlp mlp_half mlp_chain mlp_chain8; do objs="$objs $HERE/_obj/$f.o"; done
  hipcc --offload-arch=gfx950 -shared -fPIC $objs "$HERE/_obj/hash_fe$e.o" -o "$OUT/libwd_hip_fe_$e.so"
  echo "built $OUT/libwd_hip_fe_$e.so"
done
