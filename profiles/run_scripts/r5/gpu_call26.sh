#!/bin/bash
# round 5 call 26: hardware queues -- GPU_MAX_HW_QUEUES 4 (default) vs 8 for the eager pipeline and the replayed graphs
O=gpurun_out/r5_call26; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-pmc"
r() { n=$1; shift; echo "== $n"; "$@" > $O/$n.json 2> $O/$n.err || tail -3 $O/$n.err; cut -c1-260 $O/$n.json; }
r c4_graph timeout 300 python bench.py --config c4 $B
r c4_graph_again timeout 300 python bench.py --config c4 $B --no-parity
GPU_MAX_HW_QUEUES=8 r c4_graph_q8 timeout 300 python bench.py --config c4 $B --no-parity
GPU_MAX_HW_QUEUES=8 r c4_eager_q8 timeout 300 python bench.py --config c4 $B --no-graph --no-parity
GPU_MAX_HW_QUEUES=8 r c2_graph_q8 timeout 300 python bench.py $B --no-parity
r c2_graph timeout 300 python bench.py $B --no-parity
GPU_MAX_HW_QUEUES=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --config c4 --steps 32 --warmup 32 --pool 8 --repeats 1 $B --no-parity --no-graph > $O/prof_c4.log 2>&1
T=$(find $O/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 40 1 > $O/c4_step_timeline_q8.txt; grep -v hash_bucket $O/c4_step_timeline_q8.txt
rm -rf $O/prof
