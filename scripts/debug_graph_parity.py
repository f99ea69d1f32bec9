"""diagnostics: n consecutive steps as eager launches / one-step graphs / one n-step graph vs the CPU oracle."""
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

B, N = 8192, int(os.environ.get("N", "8"))
print("env", {k: v for k, v in os.environ.items() if k.startswith("WD_")})
spec = criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=16, hidden=(256, 128, 64), mode="simple")
ref = None
for mode in sys.argv[1:] or ["eager_sync", "eager", "singles", "multi"]:
    eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * 26 * 4, seed=0)
    hbs = [synth.make_raw_batch(eng.plan, B, seed=20260925 + i) for i in range(N + 1)]
    tbs = [synth.TokenBatch(eng.plan, hb) for hb in hbs]
    dev = []
    for tb in tbs:
        bt = synth.hash_tokens(eng, tb)
        torch.cuda.synchronize()
        dev.append((bt.ids.cpu().numpy().copy(), bt.bag_offs.cpu().numpy(), B))
        if os.environ.get("ZERO") == "1":
            bt.ids.zero_()
    co = CompactOracle(eng, dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step_eager(eng, tbs[N])         # warm-up step (also stepped by the oracle)
    torch.cuda.synchronize()
    co.ora.train_step(co.batch(dev[N][0], dev[N][1], B, hbs[N]["dense"], hbs[N]["labels"]))
    if os.environ.get("CLONE") == "1":
        emb0 = eng.emb.clone(); wide0 = eng.wide.clone(); touched = co.touched_mask()
    if mode == "eager_sync":
        with torch.cuda.stream(side):
            for i in range(N):
                step_eager(eng, tbs[i]); torch.cuda.synchronize()
    elif mode == "eager":
        with torch.cuda.stream(side):
            for i in range(N):
                step_eager(eng, tbs[i])
    elif mode == "singles":
        gs = [StepGraph(eng, [tbs[i]], stream=side) for i in range(N)]
        for g in gs:
            g.replay()
    elif mode == "multi":
        StepGraph(eng, tbs[:N], stream=side).replay()
    elif mode == "multi_noside":
        os.environ["WD_SPARSE_SIDE"] = "0"
        StepGraph(eng, tbs[:N], stream=side).replay()
    torch.cuda.synchronize()
    for i in range(N):
        oloss, ologits = co.ora.train_step(co.batch(dev[i][0], dev[i][1], B, hbs[i]["dense"], hbs[i]["labels"]))
    d = (eng.logit[:B].cpu() - ologits).abs().max().item()
    print("%-12s loss %.4f oracle %.4f  max|dlogit| %.3e" % (mode, float(eng.loss), oloss, d), flush=True)
    try:
        co.assert_state_matches(5e-4, 1e-5); print("   state ok")
    except AssertionError as e:
        print("   state:", str(e)[:200])
    del eng, co
    torch.cuda.empty_cache()
