#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 200 python scripts/bench_chain.py > $OUT/chain_w8.txt 2>&1; grep -v "^tile stamps\|amdgpu.ids\|wave0 F0\|row tile" $OUT/chain_w8.txt
