#!/bin/bash
# parser threads x device featurizer on the C1 loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-feat3}; mkdir -p $OUT
for th in 1 2 4; do
  WD_INGEST_THREADS=$th C1_BATCH=8192 C1_REPEAT=400 timeout 90 python scripts/bench_c1.py > $OUT/b8192_t$th.json 2> $OUT/err.txt
  WD_INGEST_THREADS=$th C1_REPEAT=100 timeout 90 python scripts/bench_c1.py > $OUT/b512_t$th.json 2>> $OUT/err.txt
done
for f in b8192_t1 b8192_t2 b8192_t4 b512_t1 b512_t2 b512_t4; do python - $OUT/$f.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "loop", d["train_loop_examples_per_sec"], "parse", d["host_parse_rows_per_sec"], "feat", d["gpu_featurize_rows_per_sec"], "step", d["train_step_only_examples_per_sec"])
PY
done
