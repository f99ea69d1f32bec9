"""Diagnostics: where a captured train step of the shipped conf (estimator._GraphStep) spends its time -- host packing + copy
(FixedStage.fill), the graph replay on the GPU (HIP events), against the eager featurizer + step."""
import os, sys, time, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd import build_estimator as BE, dataset as DS
from wide_deep_amd.estimator import _GraphStep

bs = int(os.environ.get("C1_BATCH", "512"))
lines = open(os.path.join(ROOT, "tests", "golden", "c1_rows.tsv"), "rb").read().splitlines()
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "train.tsv")
with open(path, "wb") as f:
    for _ in range(int(os.environ.get("C1_REPEAT", "10"))):
        f.write(b"\n".join(lines) + b"\n")
m = BE.build_custom_estimator(os.path.join(tmp, "model"), "wide_deep", max_batch=bs)
m.train(input_fn=lambda: DS.input_fn(path, None, "train", bs), steps=3)
raws = [r for r in DS.input_fn(path, None, "train", bs) if r.B == bs][:20]
g = m._graph_steps.get(bs) or _GraphStep(m, raws[:3])
out = {"batch": bs, "stage_bytes": g.stage.size, "tok_cap": g.stage.tok_cap}
torch.cuda.synchronize()
t0 = time.time()
for r in raws:
    g.stage.fill(r)
torch.cuda.synchronize()
out["fill_ms"] = round(1e3 * (time.time() - t0) / len(raws), 3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    g.graph.replay()
e0.record()
for _ in range(20):
    g.graph.replay()
e1.record(); e1.synchronize()
out["replay_gpu_ms"] = round(e0.elapsed_time(e1) / 20, 3)
t0 = time.time()
for r in raws:
    g.step(r)
torch.cuda.synchronize()
out["step_ms (fill + replay)"] = round(1e3 * (time.time() - t0) / len(raws), 3)
fz, eng = m._featurizer, m.engine
bts = [fz.to_device(r) for r in raws]
torch.cuda.synchronize()
e0.record()
for bt in bts:
    eng.train_step(bt)
e1.record(); e1.synchronize()
out["eager_step_gpu_ms"] = round(e0.elapsed_time(e1) / len(bts), 3)
t0 = time.time()
for r in raws:
    eng.train_step(fz.to_device(r))
torch.cuda.synchronize()
out["eager_featurize+step_ms"] = round(1e3 * (time.time() - t0) / len(raws), 3)
pdb = g.pdb
e0.record()
for _ in range(20):
    fz.run(pdb)
e1.record(); e1.synchronize()
out["featurizer_run_eager_gpu_ms"] = round(e0.elapsed_time(e1) / 20, 3)
print(json.dumps(out))

# ---- where does fill + replay lose its time?  wall clock over 40 iterations each, one synchronize at the end
def wall(fn, n=40):
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return round(1e3 * (time.time() - t0) / n, 3)

st = g.stage
host = st._host[0]
small = torch.zeros(1024, dtype=torch.uint8).pin_memory()
dsmall = torch.zeros(1024, dtype=torch.uint8, device="cuda")
out2 = {
    "replay_only": wall(lambda i: g.graph.replay()),
    "copy_only": wall(lambda i: st.dbuf.copy_(host, non_blocking=True)),
    "copy+replay": wall(lambda i: (st.dbuf.copy_(host, non_blocking=True), g.graph.replay())),
    "tiny_copy+replay": wall(lambda i: (dsmall.copy_(small, non_blocking=True), g.graph.replay())),
    "kernel_fill+replay": wall(lambda i: (dsmall.fill_(1), g.graph.replay())),
}
cs = torch.cuda.Stream()
def copy_side(i):
    with torch.cuda.stream(cs):
        st.dbuf.copy_(host, non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
    torch.cuda.current_stream().wait_event(ev)
    g.graph.replay()
out2["copy_on_side_stream+replay"] = wall(copy_side)
print(json.dumps(out2))

evs = []
def ev_sync(i):
    if len(evs) >= 3:
        evs.pop(0).synchronize()
    st.dbuf.copy_(host, non_blocking=True)
    e = torch.cuda.Event(); e.record(); evs.append(e)
    g.graph.replay()
def ev_query(i):
    if len(evs) >= 3:
        e = evs.pop(0)
        while not e.query():
            pass
    st.dbuf.copy_(host, non_blocking=True)
    e = torch.cuda.Event(); e.record(); evs.append(e)
    g.graph.replay()
out3 = {"event_sync_3_back": wall(ev_sync)}
evs.clear()
out3["event_query_3_back"] = wall(ev_query)
evs.clear()
import numpy as np
v = st._views[0]
def np_fill(i):
    st.fill(raws[i % len(raws)])
out3["fill_only_again"] = wall(np_fill)
def fill_nosync(i):
    r = raws[i % len(raws)]
    k = i % 3
    vv = st._views[k]
    nb, T = len(r.tok_bytes), len(r.tok_offs) - 2
    vv["bytes"][:nb] = r.tok_bytes
    vv["toffs"][: T + 2] = r.tok_offs
    st.dbuf.copy_(st._host[k], non_blocking=True)
    g.graph.replay()
out3["partial_fill_no_event+replay"] = wall(fill_nosync)
print(json.dumps(out3))

torch.cuda.synchronize()
cpu = []
e0.record()
for r in raws:
    t0 = time.time(); g.step(r); cpu.append(time.time() - t0)
e1.record(); e1.synchronize()
out4 = {"step_gpu_ms": round(e0.elapsed_time(e1) / len(raws), 3), "step_cpu_ms_each": [round(1e3 * c, 2) for c in cpu]}
torch.cuda.synchronize()
cpu = []
for r in raws:
    t0 = time.time(); st.fill(r); t1 = time.time(); g.graph.replay(); cpu.append((t1 - t0, time.time() - t1))
torch.cuda.synchronize()
out4["fill_cpu_ms, replay_cpu_ms"] = [(round(1e3 * a, 2), round(1e3 * b, 2)) for a, b in cpu]
print(json.dumps(out4))
