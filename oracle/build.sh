#!/bin/sh
# Build the CPU oracle (test infrastructure only). Output: oracle/_build/libwd_oracle.so
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
mkdir -p "$HERE/_build"
gcc -O2 -fopenmp -fno-fast-math -ffp-contract=off -shared -fPIC -std=c99 \
    "$HERE/wd_oracle.c" -o "$HERE/_build/libwd_oracle.so" -lm
echo "built $HERE/_build/libwd_oracle.so"
