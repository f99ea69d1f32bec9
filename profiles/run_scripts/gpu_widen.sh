#!/bin/bash
# round-2 widening check: new optimizer variants, crelu, sharded engine with mixed dims / indicator columns / world 4,
# plus two bench probes (16-row tower tile in the pipelined step; one-rank sharded step)
set -u
out=gpurun_out/${1:-widen}
mkdir -p $out
python -m pytest tests/test_gpu_optimizers.py tests/test_gpu_tf_known_answers.py -x -q > $out/pytest_opt.txt 2>&1
echo "opt rc=$?" >> $out/pytest_opt.txt
python -m pytest tests/test_gpu_step.py -x -q -k "crelu or dropout or activation" > $out/pytest_crelu.txt 2>&1
echo "crelu rc=$?" >> $out/pytest_crelu.txt
python -m pytest tests/test_gpu_dist.py -x -q -k "mixed or indicator or onehot or chain4" > $out/pytest_dist.txt 2>&1
echo "dist rc=$?" >> $out/pytest_dist.txt
WD_CHAIN_RT=16 python bench.py --steps 20 --warmup 5 --no-pmc --no-parity --no-cpu-baseline > $out/bench_rt16.json 2> $out/bench_rt16.err
python bench.py --steps 20 --warmup 5 --no-pmc --no-parity --no-cpu-baseline > $out/bench_rt32.json 2> $out/bench_rt32.err
python bench.py --steps 20 --warmup 5 --force-sharded --no-pmc --no-parity --no-cpu-baseline > $out/bench_sharded1.json 2> $out/bench_sharded1.err
tail -3 $out/pytest_opt.txt $out/pytest_crelu.txt $out/pytest_dist.txt
cat $out/bench_rt16.json $out/bench_rt32.json $out/bench_sharded1.json | cut -c1-400
