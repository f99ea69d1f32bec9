#!/usr/bin/env python
"""Calibration of the HBM read counters ON THE GATHER'S OWN ACCESS PATTERN.  `GC_ROWS` random rows out of a
26 M x 128-byte record table (3.3 GB: every request misses L2 and the Infinity Cache), requested as GC_MODE says
(wd_diag_gather_modes):
    6   64 bytes per row (4 lanes x 16 B, the first half of the record), nothing written    -> 64 B per row are NEEDED
    2   the whole 128-byte record (8 lanes x 16 B)                                          -> 128 B per row
    1   the 64 bytes + the row written (what the bare gather does)
    g   the step's own launch, wd_prefetch_onehot: 64 bytes + the 4-byte wide weight at byte 64 of the record
Run under `rocprofv3 --kernel-trace --pmc <counters>`; the counter per launch / rows = what the counter charges per row, next to a
256 MiB streaming copy in the same pass (known bytes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd.capi import call, ptr, load
load()
mode = os.environ.get("GC_MODE", "6")
n = int(os.environ.get("GC_ROWS", "212992"))
iters = int(os.environ.get("GC_ITERS", "10"))
st = torch.cuda.current_stream().cuda_stream
src = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
dst.copy_(src)
torch.cuda.synchronize()
if mode == "g":
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=16, hidden=(256, 128, 64), mode="simple")
    eng = WideDeepEngine(spec, max_batch=8192, max_nnz=8192 * 26 * 4)
    bts = [synth.to_device_ids(eng.plan, synth.make_raw_batch(eng.plan, 8192, seed=20260925 + i)) for i in range(iters)]
    assert eng.prefetch and bts[0].one_hot and bts[0].nnz == n
    for i in range(iters):
        eng._prefetch_input(bts[i], st, i % eng.n_act)
else:
    rows, rs = 26_000_000, 32
    rec = torch.empty(rows * rs, dtype=torch.float32, device="cuda").normal_()
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    out = torch.empty(n * 16, dtype=torch.float32, device="cuda")
    for i in range(iters):
        ids = torch.randint(0, rows, (n,), dtype=torch.int32, device="cuda", generator=g)
        call("wd_diag_gather_modes", ptr(rec), rs, ptr(ids), n, int(mode), ptr(out), st)
torch.cuda.synchronize()
print("mode %s rows %d iters %d" % (mode, n, iters))
