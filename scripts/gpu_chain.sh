#!/bin/bash
# one-launch tower: full GPU suite, C2 bench with / without it, kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/chain; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest.log
cat $OUT/pytest.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2> $OUT/c2.err | tee $OUT/c2_chain.json | cut -c1-170
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dist zipf 2> $OUT/c2z.err | tee $OUT/c2_chain_zipf.json | cut -c1-170
WD_CHAIN=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2> $OUT/c2b.err | tee $OUT/c2_layers.json | cut -c1-170
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --no-cpu-baseline > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace*.csv" -delete
python scripts/summarize_stats.py $OUT/kernel_stats.csv 72 > $OUT/kernel_stats.md; head -16 $OUT/kernel_stats.md
