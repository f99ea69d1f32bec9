#!/bin/bash
# A/B of one environment switch on one box: gpu_ab.sh <tag> <VAR=value>
set -u
out=gpurun_out/${1:-ab}; shift
mkdir -p $out
B="${BARGS:---steps 40 --warmup 10} --no-pmc --no-cpu-baseline --no-parity"
for i in 1 2; do
python bench.py $B > $out/new$i.json 2> $out/new.err
env "$@" python bench.py $B > $out/old$i.json 2> $out/old.err
done
python bench.py $B --dist zipf > $out/new_zipf.json 2>> $out/new.err
env "$@" python bench.py $B --dist zipf > $out/old_zipf.json 2>> $out/old.err
for f in new1 old1 new2 old2 new_zipf old_zipf; do python - $out/$f.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], d["ms_per_step"], "gather", d["roofline"]["avg_launch_us"], "tower", d["roofline_tower"]["avg_launch_us"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -3 $out/new.err
