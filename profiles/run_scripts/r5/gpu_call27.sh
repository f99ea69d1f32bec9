#!/bin/bash
# round 5 call 27: configs[3] without its crosses (row records) -- kernel stats + timeline, to price records for the crossed model
O=gpurun_out/r5_call27; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --config c4-nocross --steps 30 --warmup 5 --pool 8 --repeats 1 $B --no-parity > $O/prof.log 2>&1
T=$(find $O/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 20 1 > $O/c4nc_step_timeline.txt; grep -v hash_bucket $O/c4nc_step_timeline.txt
find $O/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $O/c4nc_kernel_stats.csv
python scripts/summarize_stats.py $O/c4nc_kernel_stats.csv 35 > $O/c4nc_kernel_stats.md; head -16 $O/c4nc_kernel_stats.md
rm -rf $O/prof
WD_ROW_RECORDS=0 timeout 300 python bench.py --config c4-nocross $B --no-parity > $O/bench_c4nc_norec.json 2> $O/bench_c4nc_norec.err; cut -c1-200 $O/bench_c4nc_norec.json
timeout 300 python bench.py --config c4-nocross $B --no-parity > $O/bench_c4nc.json 2> $O/bench_c4nc.err; cut -c1-200 $O/bench_c4nc.json
