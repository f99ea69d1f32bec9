#!/usr/bin/env python
"""Microbenchmark of the grouped weight-gradient products (wd_gemm_tn_splitk_group) at the C2 tower shape, alone (run on the
GPU box): HIP-event time per launch and TFLOP/s against the fp32 MFMA peak.  WD_TN_STREAM=0: the LDS-tiled kernel of mlp.hip;
WD_TN_SPLIT: slices of the batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd import synth
from wide_deep_amd.engine import WideDeepEngine
from wide_deep_amd.plan import criteo_spec

B = int(os.environ.get("CHAIN_B", "8192"))
iters = int(os.environ.get("CHAIN_ITERS", "100"))
hidden = tuple(int(v) for v in os.environ.get("CHAIN_HIDDEN", "256,128,64").split(","))
spec = criteo_spec(buckets=1000, hidden=hidden)
eng = WideDeepEngine(spec, max_batch=B, seed=1)
assert eng.chain
hb = synth.make_raw_batch(eng.plan, B, seed=3, pos_rate=0.3)
bt = synth.to_device_ids(eng.plan, hb)
eng.train_step(bt)
torch.cuda.synchronize()
tw = eng.towers[0]
st = torch.cuda.current_stream().cuda_stream
fn = lambda: eng._tower_backward(tw, B, st, True, True)
for _ in range(3):
    fn()
torch.cuda.synchronize()
# the launches as ONE hipGraph: the Python of a call (14 job descriptors through ctypes) costs more than the kernel
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
G = torch.cuda.CUDAGraph()
per = 10
with torch.cuda.graph(G, stream=side):
    st_c = torch.cuda.current_stream().cuda_stream
    for _ in range(per):
        eng._tower_backward(tw, B, st_c, True, True)
for _ in range(3):
    G.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters // per):
    G.replay()
e1.record(); e1.synchronize()
iters = iters // per * per
us = e0.elapsed_time(e1) / iters * 1e3
fl = 2.0 * B * sum(m["K"] * m["N"] for m in tw["metas"][:-1])
print("products B=%d hidden=%s nsplit=%s stream=%s: %.1f us per launch = %.1f TFLOP/s = %.3f of 157.3"
      % (B, hidden, tw["nsplit"][:-1], os.environ.get("WD_TN_STREAM", "1"), us, fl / us / 1e6, fl / us / 1e6 / 157.3))
