#!/usr/bin/env python
"""How fast can an MI355X answer the random row requests of ONE input-layer launch?  (the 0.6 x 8 TB/s
target of the embedding gather against the ceiling of the access pattern itself.)

n random 64-byte rows out of a 26 M x 128-byte record table (3.3 GB, far beyond L2 + Infinity Cache) are gathered into a dense
[n][16] matrix -- the contract's bytes: n * 64 read + n * 4 ids + n * 64 written -- with the request issued in eight ways
(wd_diag_gather_modes), for one C2 batch (n = 212,992) and for 2 / 4 / 8 / 16 batches per launch (the asymptote a multi-batch
look-ahead gather could reach).  HIP events over back-to-back launches on fresh id sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd.capi import call, ptr, load
load()
st = torch.cuda.current_stream().cuda_stream
rows, rs = 26_000_000, 32
rec = torch.empty(rows * rs, dtype=torch.float32, device="cuda").normal_()
g = torch.Generator(device="cuda"); g.manual_seed(7)
names = {0: "4 lanes x 16 B per row, plain loads", 1: "4 lanes x 16 B, nontemporal loads (the step's launch)",
         2: "8 lanes x 16 B: whole 128-byte record requested", 3: "1 lane per row, four 16-byte loads per lane",
         4: "LDS-DMA global_load_lds_dwordx4 -> LDS -> HBM", 5: "mode 1 + nontemporal stores",
         6: "mode 1, read side alone (no row write)", 7: "mode 1, 4 rows in flight per lane group"}
base = 212_992
print("%-58s %9s %10s %12s %10s" % ("strategy", "batches", "us/launch", "us per batch", "frac 8TB/s"))
for k in (1, 2, 4, 8, 16):
    n = base * k
    pools = [torch.randint(0, rows, (n,), dtype=torch.int32, device="cuda", generator=g) for _ in range(6)]
    out = torch.empty(n * 16, dtype=torch.float32, device="cuda")
    for mode in range(8):
        if k > 1 and mode not in (1, 5, 6, 7):
            continue
        run = lambda i: call("wd_diag_gather_modes", ptr(rec), rs, ptr(pools[i % 6]), n, mode, ptr(out), st)
        for i in range(4):
            run(i)
        torch.cuda.synchronize()
        iters = 40 if k <= 4 else 12
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            run(i)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        alg = n * 64 + n * 4 + (0 if mode == 6 else n * 64)
        print("%-58s %9d %10.2f %12.2f %10.3f" % (names[mode], k, us, us / k, alg / us / 1e3 / 8000), flush=True)
    del pools, out
