#!/bin/bash
# round 5 call 37: the GPU suite + smoke + the driver's bench command on the final tree
O=gpurun_out/r5_call37; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -n 3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err; cut -c1-400 $O/bench_driver_args.json
