"""diagnostics: a sequence of phases (eN = N eager steps, gN = one N-step graph, sN = N one-step graphs), oracle compared after each phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from tests.helpers import CompactOracle
from wide_deep_amd import synth
from wide_deep_amd.engine import WideDeepEngine
from wide_deep_amd.pipeline import StepGraph, step_eager
from wide_deep_amd.plan import criteo_spec

B = 8192
phases = sys.argv[1:]
N = sum(int(p[1:]) for p in phases)
spec = criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=16, hidden=(256, 128, 64), mode="simple")
eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * 26 * 4, seed=0)
hbs = [synth.make_raw_batch(eng.plan, B, seed=20260925 + i) for i in range(N)]
tbs = [synth.TokenBatch(eng.plan, hb) for hb in hbs]
dev = []
for tb in tbs:
    bt = synth.hash_tokens(eng, tb)
    torch.cuda.synchronize()
    dev.append((bt.ids.cpu().numpy().copy(), bt.bag_offs.cpu().numpy(), B))
co = CompactOracle(eng, dev)
side = torch.cuda.Stream()
k = 0
for p in phases:
    n = int(p[1:])
    if p[0] == "e":
        with torch.cuda.stream(side):
            for i in range(k, k + n):
                step_eager(eng, tbs[i])
    elif p[0] == "g":
        StepGraph(eng, tbs[k:k + n], stream=side).replay()
    elif p[0] == "s":
        for i in range(k, k + n):
            StepGraph(eng, [tbs[i]], stream=side).replay()
    torch.cuda.synchronize()
    for i in range(k, k + n):
        oloss, ologits = co.ora.train_step(co.batch(dev[i][0], dev[i][1], B, hbs[i]["dense"], hbs[i]["labels"]))
    k += n
    d = (eng.logit[:B].cpu() - ologits).abs().max().item()
    print("%-4s steps done %2d  loss %.4f oracle %.4f  max|dlogit| %.3e" % (p, k, float(eng.loss), oloss, d), flush=True)
