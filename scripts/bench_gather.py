#!/usr/bin/env python
"""Embedding-gather kernel alone on the C2 workload (run on the GPU box): HIP-event timing, and -- under
`rocprofv3 --pmc` request counters by size (bench.py pmc_traffic; profiles/r5_gather_counter_calibration.md) -- its HBM traffic next to a streaming copy of KNOWN size that
calibrates the counters (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd import synth
from wide_deep_amd.engine import WideDeepEngine
from wide_deep_amd.plan import criteo_spec
import bench as _bench

B = int(os.environ.get("GATHER_BATCH", "8192"))
iters = int(os.environ.get("GATHER_ITERS", "200"))
pool = int(os.environ.get("GATHER_POOL", "16"))
dist = os.environ.get("GATHER_DIST", "uniform")
spec, mean_len = _bench.make_spec(os.environ.get("GATHER_CONFIG", "c2"))
eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * 26 * (2 * mean_len + 2))
plan = eng.plan
bts = [synth.to_device_ids(plan, synth.make_raw_batch(plan, B, seed=20260925 + i, dist=dist, mean_len=mean_len)) for i in range(pool)]
tw0 = eng.towers[0]
ld = tw0["layout"].ld
xp = tw0["act"].data_ptr() + 4 * tw0["layout"].seg_start[0]
st = torch.cuda.current_stream().cuda_stream
(dim, gs), = list(eng.group_slots.items())

# calibration: copy of a KNOWN 256 MiB (read 256 MiB + write 256 MiB), 16 B per lane
src = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()

# the gather as the engine launches it: wd_prefetch_onehot (row records + wide weights + numeric columns: what the step runs
# for one-id-per-bag batches) or the embedding-bag kernel of the C ABI
if eng.prefetch and bts[0].one_hot:
    run = lambda i: eng._prefetch_input(bts[i % pool], st, i % eng.n_act)
    kname = "prefetch_onehot"
else:
    run = lambda i: eng.embag_fwd(dim, gs, bts[i % pool], xp, ld, st)
    kname = "embag_fwd"
for i in range(10):
    run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(iters):
    run(i)
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
bt = bts[0]
alg = bt.nnz * dim * 4 + bt.nnz * 4 + (bt.B * plan.S + 1) * 4 + bt.B * gs.numel() * dim * 4
print(json.dumps({"kernel": kname, "ids": dist, "avg_us": round(us, 2), "alg_bytes": alg,
                  "GBps": round(alg / us / 1e3, 1), "frac_of_8TBps": round(alg / us / 1e3 / 8000, 4)}))
