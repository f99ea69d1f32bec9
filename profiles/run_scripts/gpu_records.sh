#!/bin/bash
# row-record table layout: parity tests + A/B bench on one box
set -u
out=gpurun_out/${1:-rec}
mkdir -p $out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_chain.py -x -q > $out/pytest_a.txt 2>&1
echo "a rc=$?" >> $out/pytest_a.txt
python -m pytest tests/test_gpu_fullsize.py -x -q > $out/pytest_full.txt 2>&1
echo "full rc=$?" >> $out/pytest_full.txt
B="--steps 20 --warmup 5 --no-pmc --no-cpu-baseline"
python bench.py $B > $out/bench_rec.json 2> $out/bench_rec.err
WD_ROW_RECORDS=0 python bench.py $B > $out/bench_sep.json 2> $out/bench_sep.err
python bench.py $B --dist zipf > $out/bench_rec_zipf.json 2>> $out/bench_rec.err
WD_ROW_RECORDS=0 python bench.py $B --dist zipf > $out/bench_sep_zipf.json 2>> $out/bench_sep.err
python bench.py $B > $out/bench_rec2.json 2>> $out/bench_rec.err
tail -n 4 $out/pytest_a.txt; tail -n 4 $out/pytest_full.txt
for f in rec sep rec_zipf sep_zipf rec2; do python - $out/bench_$f.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("table_layout"), "| roofline", d["roofline"].get("avg_launch_us", d["roofline"].get("duration_us")), d["roofline"]["frac"], "| parity", (d.get("parity") or {}).get("max_abs_dlogit"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
