#!/bin/bash
# round 5 call 24: general pipelined capture (bucketing of t+1 beside tower(t)), forked input layer / small-table update
O=gpurun_out/r5_call24; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_c4.py -x -q -m gpu -k "c4 or small or multi" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
B="--no-cpu-baseline --no-pmc"
for v in "" "WD_FORK=0" "WD_PIPELINE_GENERAL=0" ; do
  echo "== $v"; env $v timeout 300 python bench.py --config c4 $B > $O/bench_c4_$v.json 2> $O/bench_c4_$v.err; cut -c1-330 $O/bench_c4_$v.json
done
timeout 300 python bench.py --config c4-nocross $B > $O/bench_c4nc.json 2> $O/bench_c4nc.err; cut -c1-330 $O/bench_c4nc.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --config c4 --steps 30 --warmup 5 --pool 8 --repeats 1 $B --no-parity > $O/prof_c4.log 2>&1
T=$(find $O/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 20 1 > $O/c4_step_timeline.txt; grep -v hash_bucket $O/c4_step_timeline.txt
rm -rf $O/prof
