#!/bin/bash
set -u
out=gpurun_out/${1:-head}
mkdir -p $out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_half.py tests/test_gpu_step.py -x -q -k "head or half or modes or dense or resnet or dropout or tower" > $out/pytest.txt 2>&1
echo "rc=$?" >> $out/pytest.txt; tail -n 3 $out/pytest.txt
for cfg in c5 c4; do
  python bench.py --config $cfg --steps 60 --warmup 20 --no-cpu-baseline --no-pmc --no-parity > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  WD_HEAD_SCALAR=1 python bench.py --config $cfg --steps 60 --warmup 20 --no-cpu-baseline --no-pmc --no-parity > $out/bench_${cfg}_scalar.json 2>> $out/bench_$cfg.err
  cut -c1-160 $out/bench_$cfg.json; cut -c1-160 $out/bench_${cfg}_scalar.json
done
scripts/gpu_stats.sh ${1:-head}/c5 40 --config c5 --steps 32 --warmup 8 --pool 8 > /dev/null 2>&1; head -24 $out/c5/kernel_stats.md
