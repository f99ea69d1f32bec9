#!/usr/bin/env python
"""BASELINE configs[0] (C1) end to end on the GPU box: repo-default conf on real rows (tests/golden/c1_rows.tsv repeated),
`python train.py`-equivalent loop at --batch_size 512: host TSV parse + GPU featurizer + train step.  Prints examples/sec
and where the time goes (the hot path itself vs the host ingest that SURVEY 8(f) row f1 leaves for a later round)."""
import os, sys, time, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd import build_estimator as BE, dataset as DS

rep = int(os.environ.get("C1_REPEAT", "20"))
bs = int(os.environ.get("C1_BATCH", "512"))
lines = open(os.path.join(ROOT, "tests", "golden", "c1_rows.tsv"), "rb").read().splitlines()
tmp = tempfile.mkdtemp()
import atexit, shutil
atexit.register(shutil.rmtree, tmp, True)      # (the model directory holds a 1.3 GB checkpoint per run)
path = os.path.join(tmp, "train.tsv")
with open(path, "wb") as f:
    for _ in range(rep):
        f.write(b"\n".join(lines) + b"\n")
n = rep * len(lines)
m = BE.build_custom_estimator(os.path.join(tmp, "model"), "wide_deep", max_batch=bs)
m.train(input_fn=lambda: DS.input_fn(path, None, "train", bs), steps=2)          # engine build + warm-up
t0 = time.time(); m.train(input_fn=lambda: DS.input_fn(path, None, "train", bs)); t_all = time.time() - t0
# split: parse only / parse + featurize / steps on pre-featurized batches
t0 = time.time(); raws = list(DS.input_fn(path, None, "eval", bs)); t_parse = time.time() - t0
fz = m._featurizer
t0 = time.time(); bts = [fz.to_device(r) for r in raws]; torch.cuda.synchronize(); t_feat = time.time() - t0
t0 = time.time()
for bt in bts:
    m.engine.train_step(bt)
torch.cuda.synchronize(); t_step = time.time() - t0
print(json.dumps({"config": "C1 repo-default conf, real rows, batch %d" % bs, "rows": n,
                  "train_examples_per_sec": round(n / t_all, 1),
                  "train_loop_examples_per_sec": round(m.last_train["examples"] / m.last_train["seconds"], 1),
                  "train_loop_steps": m.last_train["steps"], "train_loop_seconds": round(m.last_train["seconds"], 4),
                  "train_loop_host_seconds": m.last_train.get("host_seconds"),
                  "checkpoint_restore_save_sec": round(t_all - m.last_train["seconds"], 3),
                  "host_parse_rows_per_sec": round(n / t_parse, 1),
                  "gpu_featurize_rows_per_sec": round(n / t_feat, 1),
                  "train_step_only_examples_per_sec": round(n / t_step, 1),
                  "note": "train = checkpoint restore + loop (parse + featurize + step) + checkpoint save; eager launches (no hipGraph) at batch %d" % bs}))
