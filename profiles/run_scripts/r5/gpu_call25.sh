#!/bin/bash
# round 5 call 25: configs[3] -- plain capture with the forked input layer vs the pipelined schedule on eager streams
O=gpurun_out/r5_call25; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-pmc"
echo "== graph"; timeout 300 python bench.py --config c4 $B > $O/bench_c4_graph.json 2> $O/bench_c4_graph.err; cut -c1-330 $O/bench_c4_graph.json
echo "== eager pipeline"; timeout 300 python bench.py --config c4 $B --no-graph > $O/bench_c4_eager.json 2> $O/bench_c4_eager.err; cut -c1-330 $O/bench_c4_eager.json; tail -3 $O/bench_c4_eager.err
echo "== eager pipeline, no fork"; WD_FORK=0 timeout 300 python bench.py --config c4 $B --no-graph > $O/bench_c4_eager_nofork.json 2> $O/bench_c4_eager_nofork.err; cut -c1-330 $O/bench_c4_eager_nofork.json
echo "== eager step by step"; WD_EAGER_PIPELINE=0 timeout 300 python bench.py --config c4 $B --no-graph > $O/bench_c4_eager_plain.json 2> $O/bench_c4_eager_plain.err; cut -c1-330 $O/bench_c4_eager_plain.json
echo "== nocross eager pipeline"; timeout 300 python bench.py --config c4-nocross $B --no-graph > $O/bench_c4nc_eager.json 2> $O/bench_c4nc_eager.err; cut -c1-330 $O/bench_c4nc_eager.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --config c4 --steps 32 --warmup 32 --pool 8 --repeats 1 $B --no-parity --no-graph > $O/prof_c4.log 2>&1
T=$(find $O/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 40 1 > $O/c4_step_timeline.txt; grep -v hash_bucket $O/c4_step_timeline.txt
rm -rf $O/prof
