#!/bin/bash
# round 5, call 8: is the chip power / clock limited during the step?  sclk + socket power sampled while the step loops
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call8; mkdir -p $OUT
rocm-smi --showpower --showclocks --showperflevel > $OUT/idle.txt 2>&1; grep -i "sclk\|power\|mclk\|fclk" $OUT/idle.txt | head
rocm-smi --showmaxpower 2>&1 | grep -i "max" | head -3
sample() { for k in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Socket Power\|mclk" | tr '\n' ' ' ; echo; sleep 0.3; done; }
echo "== full step loop"
WD_TN_STREAM=0 WD_TN_SPLIT=16 timeout 120 python bench.py --steps 30000 --warmup 5 --repeats 1 --pool 16 --no-cpu-baseline --no-pmc --no-parity > $OUT/loop.json 2>&1 &
P=$!; sleep 14; sample; wait $P; grep -o '"ms_per_step": [0-9.]*' $OUT/loop.json
echo "== products only loop (streamed, 13 slices)"
WD_TN_SPLIT=13 CHAIN_ITERS=150000 timeout 120 python scripts/bench_tn.py > $OUT/tn.txt 2>&1 &
P=$!; sleep 11; sample; wait $P; cat $OUT/tn.txt | grep products
echo "== tower only loop"
CHAIN_ITERS=30000 timeout 120 python scripts/bench_chain.py > $OUT/ch.txt 2>&1 &
P=$!; sleep 11; sample; wait $P; grep "^chain B" $OUT/ch.txt
