#!/bin/bash
# round 5, call 15: small-table path for crossed columns: tests + C4 (with crosses) bench + kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call15; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_c4.py tests/test_gpu_c1.py -q -m gpu -x > $OUT/pytest.txt 2>&1; tail -n 30 $OUT/pytest.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c4" > $OUT/pytest2.txt 2>&1; tail -n 30 $OUT/pytest2.txt
for v in "WD_SMALL_TABLES=cross" "WD_SMALL_TABLES=0"; do
env $v timeout 600 python bench.py --config c4 --steps 40 --warmup 5 --repeats 5 --no-pmc --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; python -c "
import json,sys
try:
    d=json.loads(open('$OUT/bench_c4.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['repeats_ms_per_step'], d['parity']['hash_ids_bit_exact'], d['parity']['max_abs_dlogit'])
except Exception as e: print('FAILED',e); print(open('$OUT/bench_c4.err').read()[-2000:])
"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --config c4 --steps 30 --warmup 5 --pool 8 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c4_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c4_kernel_stats.csv 35 > $OUT/c4_kernel_stats.md; head -22 $OUT/c4_kernel_stats.md
rm -rf $OUT/prof
