#!/bin/bash
# round 5 call 23: chain tests after the forward split; configs[3] step timeline (which kernels are the step's long pole)
O=gpurun_out/r5_call23; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -m gpu > $O/pytest_chain.txt 2>&1; tail -3 $O/pytest_chain.txt
B="--no-cpu-baseline --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --config c4 --steps 30 --warmup 5 --pool 8 --repeats 1 $B --no-parity > $O/prof_c4.log 2>&1
T=$(find $O/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 20 1 > $O/c4_step_timeline.txt; grep -v hash_bucket $O/c4_step_timeline.txt
rm -rf $O/prof
