#!/usr/bin/env python
"""Measured ceilings on the GPU box: streaming copy (float4) and random 64-byte row gathers from a 1.66 GB table."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd.capi import call, ptr, load
load()
st = torch.cuda.current_stream().cuda_stream
rows = 26_000_000
tab = torch.empty(rows * 16, dtype=torch.float32, device="cuda").normal_()
out = torch.zeros(1 << 22, dtype=torch.float32, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(i) if fn.__code__.co_argcount else fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

res = []
for n in (212992, 1 << 20, 1 << 22, 1 << 24):
    pool = [torch.randint(0, rows, (n,), dtype=torch.int32, device="cuda", generator=g) for _ in range(8 if n < (1 << 22) else 2)]
    for per in (1, 2, 4, 8):
        k = [0]
        def run():
            k[0] += 1
            call("wd_diag_gather64", ptr(tab), ptr(pool[k[0] % len(pool)]), n, per, ptr(out), st)
        us = timeit(run, 50 if n < (1 << 22) else 10)
        res.append({"rows": n, "per": per, "us": round(us, 2), "row_GBps": round(n * 64 / us / 1e3, 1)})
        print(res[-1])
src = torch.empty(64 << 20, dtype=torch.float32, device="cuda").normal_(); dst = torch.empty_like(src)
us = timeit(lambda: dst.copy_(src), 20)
print({"copy_256MiB_us": round(us, 1), "GBps(read+write)": round(2 * src.numel() * 4 / us / 1e3, 1)})
