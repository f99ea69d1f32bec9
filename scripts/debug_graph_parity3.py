"""diagnostics: per-step comparison from re-synchronised state (CompactOracle.resync) -- separates a wrong step from the
step-to-step amplification of rounding differences."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 11
dist = sys.argv[2] if len(sys.argv) > 2 else "uniform"
spec = criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=16, hidden=(256, 128, 64), mode="simple")
eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * 26 * 4, seed=0)
hbs = [synth.make_raw_batch(eng.plan, B, seed=20260925 + i, dist=dist) for i in range(N)]
tbs = [synth.TokenBatch(eng.plan, hb) for hb in hbs]
dev = []
for tb in tbs:
    bt = synth.hash_tokens(eng, tb)
    torch.cuda.synchronize()
    dev.append((bt.ids.cpu().numpy().copy(), bt.bag_offs.cpu().numpy(), B))
co = CompactOracle(eng, dev)
side = torch.cuda.Stream()
for i in range(N):
    co.resync()
    with torch.cuda.stream(side):
        step_eager(eng, tbs[i])
    torch.cuda.synchronize()
    oloss, ologits = co.ora.train_step(co.batch(dev[i][0], dev[i][1], B, hbs[i]["dense"], hbs[i]["labels"]))
    d = (eng.logit[:B].cpu() - ologits).abs().max().item()
    msg = "ok"
    for tol in ((1e-5, 1e-6), (1e-4, 1e-5), (5e-4, 1e-5), (1e-3, 1e-4), (1e-2, 1e-3)):
        try:
            co.assert_state_matches(*tol); msg = "state within rtol %g atol %g" % tol; break
        except AssertionError as e:
            msg = str(e)[:150]
    print("step %2d loss %.4f oracle %.4f max|dlogit| %.3e | %s" % (i, float(eng.loss), oloss, d, msg), flush=True)
