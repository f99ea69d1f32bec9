#!/bin/bash
# round 5, call 21: concatenating towers in the one launch (ResDnn of configs[3]): full-size parity + C4 bench lines + kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call21; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_c4.py tests/test_gpu_step.py -q -m gpu -x -k "c4 or resnet or dense or mode or connect" > $OUT/pytest.txt 2>&1; tail -n 8 $OUT/pytest.txt
for v in "WD_CHAIN_WINDOWS=1" "WD_CHAIN_WINDOWS=0"; do
for c in c4 c4-nocross; do
env $v timeout 600 python bench.py --config $c --steps 40 --warmup 5 --repeats 5 --no-pmc --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json
try:
    d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('$v $c', d['ms_per_step'], d['ms_per_step_min_median_max'], d['parity']['hash_ids_bit_exact'], d['parity']['max_abs_dlogit'])
except Exception as e: print('$v $c FAILED',e); print(open('$OUT/bench.err').read()[-1500:])
"
done; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --config c4 --steps 30 --warmup 5 --pool 8 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c4_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c4_kernel_stats.csv 35 > $OUT/c4_kernel_stats.md; head -20 $OUT/c4_kernel_stats.md
rm -rf $OUT/prof
