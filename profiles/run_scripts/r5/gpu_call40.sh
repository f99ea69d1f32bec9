#!/bin/bash
# round 5 call 40: configs[3] without its crosses on the final tree -- kernel stats + a steady-state step of the multi-step graph
O=gpurun_out/r5_call40; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-pmc --no-parity"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --config c4-nocross --steps 32 --warmup 16 --pool 16 --repeats 1 $B > $O/prof.log 2>&1
T=$(find $O/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 37 1 > $O/c4nc_step_timeline.txt; grep -v "hash_bucket" $O/c4nc_step_timeline.txt
find $O/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $O/c4nc_kernel_stats.csv
python scripts/summarize_stats.py $O/c4nc_kernel_stats.csv 48 > $O/c4nc_kernel_stats.md; head -14 $O/c4nc_kernel_stats.md
rm -rf $O/prof
