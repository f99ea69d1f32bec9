#!/bin/bash
# round 5 call 28: 512-chunk bucketing + one-round column scan; small tables' update behind the dense tail
O=gpurun_out/r5_call28; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_c4.py tests/test_gpu_step.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py tests/test_gpu_chain.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
B="--no-cpu-baseline --no-pmc"
r() { n=$1; shift; echo "== $n"; "$@" > $O/$n.json 2> $O/$n.err || tail -3 $O/$n.err; cut -c1-260 $O/$n.json; }
r c4 timeout 300 python bench.py --config c4 $B
r c4nc timeout 300 python bench.py --config c4-nocross $B --no-parity
MASTER_PORT=29561 r c2_sharded timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded $B --no-parity
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --config c4 --steps 30 --warmup 5 --pool 8 --repeats 1 $B --no-parity > $O/prof_c4.log 2>&1
T=$(find $O/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 20 1 > $O/c4_step_timeline.txt; grep -v hash_bucket $O/c4_step_timeline.txt
rm -rf $O/prof
