#!/bin/bash
# round 5 call 22: window-mode forward stages with a split (dedicated scratch)
O=gpurun_out/r5_call22; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -m gpu > $O/pytest_chain.txt 2>&1; tail -3 $O/pytest_chain.txt
CHAIN_MODE=resnet timeout 300 python scripts/bench_chain.py > $O/bench_chain_resnet.txt 2>&1; tail -12 $O/bench_chain_resnet.txt
timeout 300 python bench.py --config c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; cat $O/bench_c4.json | cut -c1-400
timeout 300 python bench.py --config c4-nocross --no-cpu-baseline > $O/bench_c4nc.json 2> $O/bench_c4nc.err; cat $O/bench_c4nc.json | cut -c1-300
