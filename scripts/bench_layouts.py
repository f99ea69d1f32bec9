#!/usr/bin/env python
"""Prices of table-row layouts on the GPU box (wd_diag_access): random reads / read-modify-writes of one batch's 212,992 rows
out of 26 M, for the layouts the forward gather and the fused row update could use."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd.capi import call, ptr, load
load()
st = torch.cuda.current_stream().cuda_stream
rows, n = 26_000_000, 212992
g = torch.Generator(device="cuda"); g.manual_seed(1)
pool = [torch.randint(0, rows, (n,), dtype=torch.int32, device="cuda", generator=g) for _ in range(8)]
out = torch.zeros(1 << 16, dtype=torch.float32, device="cuda")
bufs = {k: torch.zeros(rows * v // 4, dtype=torch.float32, device="cuda") for k, v in
        dict(e64=64, a64=64, w16=16, r128=128, r256=256).items()}


def run(name, segs, rmw, per=2, iters=40):
    ns = len(segs)
    base = (ctypes.c_void_p * ns)(*[bufs[s[0]].data_ptr() + s[3] for s in segs])
    stride = (ctypes.c_int64 * ns)(*[s[1] for s in segs])
    nbytes = (ctypes.c_int32 * ns)(*[s[2] for s in segs])
    k = [0]
    def f():
        k[0] += 1
        call("wd_diag_access", base, stride, nbytes, ns, ptr(pool[k[0] % 8]), n, per, rmw, ptr(out), st)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    tot = sum(s[2] for s in segs)
    print("%-64s %s per %d  %7.2f us  %7.1f GB/s touched" % (name, "RMW " if rmw else "read", per, us, n * tot * (2 if rmw else 1) / us / 1e3), flush=True)


# (buffer, stride bytes, bytes touched, byte offset inside the record)
for per in (1, 2, 4):
    run("fwd now: 64 B row + 16 B wide line (2 tables)", [("e64", 64, 64, 0), ("w16", 16, 16, 0)], 0, per)
    run("fwd: 64 B row only", [("e64", 64, 64, 0)], 0, per)
    run("fwd record128: row + {w,z,n} in one 128 B line (80 B)", [("r128", 128, 80, 0)], 0, per)
    run("fwd record128: whole 128 B line", [("r128", 128, 128, 0)], 0, per)
    run("fwd record256: first 80 B of a 256 B record", [("r256", 256, 80, 0)], 0, per)
    run("upd now: row 64 + accumulator 64 + wide 16 (3 tables)", [("e64", 64, 64, 0), ("a64", 64, 64, 0), ("w16", 16, 16, 0)], 1, per)
    run("upd record128 {row, wzn} + accumulator 64 (2 tables)", [("r128", 128, 80, 0), ("a64", 64, 64, 0)], 1, per)
    run("upd record128 {row, acc} + wide 16 (2 tables)", [("r128", 128, 128, 0), ("w16", 16, 16, 0)], 1, per)
    run("upd record256 {row, wzn, -, acc}: 80 B + 64 B of one record", [("r256", 256, 80, 0), ("r256", 256, 64, 128)], 1, per)
    run("upd record256: first 144 B contiguous {row, acc, wzn}", [("r256", 256, 144, 0)], 1, per)
