#!/bin/bash
# one-rank RCCL group diagnostics of the sharded bench path (bench.py --force-sharded)
set -u
out=gpurun_out/${1:-shdbg}
mkdir -p $out
B="--steps 20 --warmup 5 --force-sharded --no-pmc --no-parity --no-cpu-baseline"
run() { name=$1; shift; ( "$@" ) > $out/$name.json 2> $out/$name.err; echo "$name rc=$?" >> $out/summary.txt; head -c 300 $out/$name.json >> $out/summary.txt; echo >> $out/summary.txt; }
run eager      timeout 120 python -X faulthandler bench.py $B --no-graph
run pool2      timeout 120 python -X faulthandler bench.py $B --pool 2
run default    timeout 120 python -X faulthandler bench.py $B
run nocache    env TORCH_NCCL_CUDA_EVENT_CACHE=0 timeout 120 python -X faulthandler bench.py $B
python -m pytest tests/test_gpu_step.py -x -q -k "crelu" > $out/pytest_crelu.txt 2>&1
echo "crelu rc=$?" >> $out/summary.txt
cat $out/summary.txt
for f in eager pool2 default nocache; do echo "== $f"; grep -n "File \"\|Thread\|Error\|error" $out/$f.err | head -30; done
