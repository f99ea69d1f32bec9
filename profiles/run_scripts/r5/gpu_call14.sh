#!/bin/bash
# round 5, call 14: BASELINE configs[3] with its crossed columns at size: parity test + bench lines (with / without the crosses)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call14; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c4" > $OUT/pytest.txt 2>&1; tail -n 30 $OUT/pytest.txt
timeout 600 python bench.py --config c4 --steps 40 --warmup 5 --repeats 5 --no-pmc --cpu-steps 4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; tail -c 3000 $OUT/bench_c4.json; tail -n 5 $OUT/bench_c4.err
timeout 600 python bench.py --config c4-nocross --steps 40 --warmup 5 --repeats 5 --no-pmc --no-cpu-baseline > $OUT/bench_c4_nocross.json 2> $OUT/bench_c4nx.err; python -c "
import json,sys
for f in ('bench_c4','bench_c4_nocross'):
    try:
        d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['repeats_ms_per_step'], d.get('parity'))
    except Exception as e: print(f,'FAILED',e)
"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --config c4 --steps 30 --warmup 5 --pool 8 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c4_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c4_kernel_stats.csv 35 > $OUT/c4_kernel_stats.md; head -22 $OUT/c4_kernel_stats.md
rm -rf $OUT/prof
