#!/bin/bash
# round 6: python train.py's loop on the shipped conf -- captured featurizer + step per batch size -- at batch 64 / 512 / 8192,
# against the eager launches (WD_TRAIN_GRAPH=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r6c1}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_c1.py -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest.log; cat $OUT/pytest.log
for bs in 64 512 8192; do
  rep=200; [ $bs = 8192 ] && rep=1500; [ $bs = 64 ] && rep=40     # >= 100 steps each: the capture (once per batch size) is amortised as in a real run
  C1_BATCH=$bs C1_REPEAT=$rep timeout 400 python scripts/bench_c1.py > $OUT/b${bs}_graph.json 2> $OUT/b${bs}_graph.err
  WD_TRAIN_GRAPH=0 C1_BATCH=$bs C1_REPEAT=$rep timeout 400 python scripts/bench_c1.py > $OUT/b${bs}_eager.json 2> $OUT/b${bs}_eager.err
  tail -n 2 $OUT/b${bs}_graph.err
  python - $OUT $bs <<'PY'
import json, sys
for tag in ("graph", "eager"):
    try:
        d = json.loads(open("%s/b%s_%s.json" % (sys.argv[1], sys.argv[2], tag)).read().strip().splitlines()[-1])
        print("batch %s %s: loop %.0f ex/s, step only %.0f, featurize %.0f, parse %.0f" % (sys.argv[2], tag, d["train_loop_examples_per_sec"], d["train_step_only_examples_per_sec"], d["gpu_featurize_rows_per_sec"], d["host_parse_rows_per_sec"]))
    except Exception as e:
        print(sys.argv[2], tag, "FAILED", e)
PY
done
