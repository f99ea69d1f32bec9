#!/bin/bash
# round 6: the bag-parallel crossed-column emit kernel -- parity tests that go through the featurizer, its launches timed apart,
# kernel stats, and configs[3] timed from tokens.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r6feat}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_c1.py tests/test_gpu_c4.py tests/test_gpu_kernels.py -q -m gpu -x > $OUT/pytest_feat.txt 2>&1; tail -n 5 $OUT/pytest_feat.txt
timeout 300 python scripts/bench_featurizer.py --check > $OUT/featurizer.json 2> $OUT/featurizer.err; cat $OUT/featurizer.json; tail -n 3 $OUT/featurizer.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python scripts/bench_featurizer.py --iters 30 > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/c4_featurizer_kernel_stats.csv
python scripts/summarize_stats.py $OUT/c4_featurizer_kernel_stats.csv 35 > $OUT/c4_featurizer_kernel_stats.md; head -14 $OUT/c4_featurizer_kernel_stats.md
rm -rf $OUT/prof
timeout 600 python bench.py --config c4 --steps 60 --no-cpu-baseline --no-pmc > $OUT/bench_c4_tokens.json 2> $OUT/bench_c4.err; tail -c 1500 $OUT/bench_c4_tokens.json; tail -n 5 $OUT/bench_c4.err
timeout 600 python bench.py --config c4 --steps 60 --no-cpu-baseline --no-pmc --ids-input --no-parity > $OUT/bench_c4_ids.json 2>> $OUT/bench_c4.err; python -c "
import json,sys
for f in ('bench_c4_tokens','bench_c4_ids'):
    try:
        d=json.loads(open('$OUT/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d.get('repeats_ms_per_step'), d.get('parity'))
    except Exception as e: print(f,'FAILED',e)
"
