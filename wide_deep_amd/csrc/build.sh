#!/bin/sh
# Build the gfx950 C-ABI library of the hot path: wide_deep_amd/_lib/libwd_hip.so
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_lib"
mkdir -p "$OUT" "$HERE/_obj"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=""
for f in common hash embag sparse_update sparse_fused small_tables onehot_path dist_exchange mlp mlp_half mlp_chain mlp_chain8; do
  if [ ! -f "$HERE/_obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/_obj/$f.o" ] || [ "$HERE/common.h" -nt "$HERE/_obj/$f.o" ] || [ "$HERE/mlp_chain.h" -nt "$HERE/_obj/$f.o" ] \
     || [ "$HERE/../../include/wd_hip.h" -nt "$HERE/_obj/$f.o" ]; then
    hipcc $FLAGS -c "$HERE/$f.hip" -o "$HERE/_obj/$f.o" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
# source stamp: sha256 over the sources this library is built from (sorted by name; capi.load() recomputes it and refuses a
# library that was built from other sources -- the prebuilt .so travels to the GPU box, a stale one would otherwise load silently)
STAMP=$(cd "$HERE" && cat $(ls *.hip *.h ../../include/wd_hip.h | LC_ALL=C sort) | sha256sum | cut -c1-64)
printf 'extern "C" const char *wd_build_stamp(void) { return "%s"; }\n' "$STAMP" > "$HERE/_obj/build_stamp.cpp"
g++ -O1 -fPIC -c "$HERE/_obj/build_stamp.cpp" -o "$HERE/_obj/build_stamp.o"
hipcc --offload-arch=gfx950 -shared -fPIC "$HERE"/_obj/build_stamp.o "$HERE"/_obj/common.o "$HERE"/_obj/hash.o "$HERE"/_obj/embag.o \
      "$HERE"/_obj/sparse_update.o "$HERE"/_obj/sparse_fused.o "$HERE"/_obj/small_tables.o "$HERE"/_obj/onehot_path.o "$HERE"/_obj/dist_exchange.o "$HERE"/_obj/mlp.o "$HERE"/_obj/mlp_half.o "$HERE"/_obj/mlp_chain.o "$HERE"/_obj/mlp_chain8.o -o "$OUT/libwd_hip.so"
echo "built $OUT/libwd_hip.so"
# host-side TSV ingest (plain C, no GPU code)
gcc -O3 -fPIC -shared -std=c99 -Wall "$HERE/tsv_ingest.c" -o "$OUT/libwd_ingest.so"
echo "built $OUT/libwd_ingest.so"
