#!/bin/bash
# round 5, call 18: small-table kernels tuned (ring of bags, one round of staging loads, smaller forward workgroups): tests + C4 + the default bench with the recalibrated traffic
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call18; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_c4.py -q -m gpu -x > $OUT/pytest.txt 2>&1; tail -n 5 $OUT/pytest.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c4_full_size_with" > $OUT/pytest2.txt 2>&1; tail -n 5 $OUT/pytest2.txt
timeout 600 python bench.py --config c4 --steps 40 --warmup 5 --repeats 5 --no-pmc --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; python -c "
import json
d=json.loads(open('$OUT/bench_c4.json').read().strip().splitlines()[-1]); print('c4', d['ms_per_step'], d['repeats_ms_per_step'], d['parity']['max_abs_dlogit'])"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --config c4 --steps 30 --warmup 5 --pool 8 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c4_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c4_kernel_stats.csv 35 > $OUT/c4_kernel_stats.md; grep "small\|bucket_update\|k_gemm" $OUT/c4_kernel_stats.md
rm -rf $OUT/prof
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; python -c "
import json
d=json.loads(open('$OUT/bench_c2.json').read().strip().splitlines()[-1]); print('c2', d['ms_per_step'], d['ms_per_step_min_median_max']); r=d['roofline']; print({k:r.get(k) for k in ('frac','traffic','moved_GBps','moved_frac_of_peak','avg_launch_us')}); print(r['traffic_source'][:600]); print(d['roofline_gather_kernel']['traffic_source'][:700])"; tail -3 $OUT/bench_c2.err
