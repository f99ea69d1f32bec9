#!/bin/bash
# Run on the GPU box (via gpurun): GPU parity tests, bench line, rocprof kernel stats. Output -> gpurun_out/<tag>/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r1a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
nproc > $OUT/nproc.txt; rocm-smi --showproductname 2>/dev/null | head -20 > $OUT/gpu.txt
python -c "import torch;print(torch.cuda.device_count(), torch.cuda.get_device_name(0))" >> $OUT/gpu.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
timeout 600 python bench.py --dist zipf --no-cpu-baseline > $OUT/bench_zipf.json 2>> $OUT/bench.err
cat $OUT/bench_zipf.json
timeout 600 python bench.py --no-graph --no-cpu-baseline > $OUT/bench_nograph.json 2>> $OUT/bench.err
cat $OUT/bench_nograph.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace*.csv" -delete; find $OUT/prof -name "*.db" -delete
tail -2 $OUT/prof_bench.log
head -40 $OUT/kernel_stats.csv
