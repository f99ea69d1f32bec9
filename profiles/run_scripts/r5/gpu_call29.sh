#!/bin/bash
# round 5 call 29: the tower launch waits for the ragged bucketing (which cannot share a CU with it)
O=gpurun_out/r5_call29; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_c4.py tests/test_gpu_fullsize.py -x -q -m gpu -k "c4 or small or multi" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="--no-cpu-baseline --no-pmc"
r() { n=$1; shift; echo "== $n"; "$@" > $O/$n.json 2> $O/$n.err || tail -3 $O/$n.err; cut -c1-260 $O/$n.json; }
r c4 timeout 300 python bench.py --config c4 $B --no-parity
r c4nc timeout 300 python bench.py --config c4-nocross $B --no-parity
WD_BUCKET_JOIN=0 r c4nc_nojoin timeout 300 python bench.py --config c4-nocross $B --no-parity
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --config c4-nocross --steps 30 --warmup 5 --pool 8 --repeats 1 $B --no-parity > $O/prof.log 2>&1
T=$(find $O/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 20 1 > $O/c4nc_step_timeline.txt; grep -v hash_bucket $O/c4nc_step_timeline.txt
rm -rf $O/prof
