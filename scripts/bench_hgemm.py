#!/usr/bin/env python
"""fp16-operand GEMMs of the C5 tower as stand-alone launches (HIP events): NN / NT(dz) / NT(acc) / TN at the layer shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd.capi import call, ptr, load
load()
st = torch.cuda.current_stream().cuda_stream
B = 8192
f16 = dict(dtype=torch.float16, device="cuda")
iters = int(os.environ.get("ITERS", "30"))


def hz(r, c):
    return (torch.randn(r * c + 64, device="cuda") * 0.1).half()[: r * c].view(r, c)


def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


r8 = lambda v: (v + 7) // 8 * 8
which = os.environ.get("WHICH", "nn,tn,dz").split(",")
for K, N in ((1677, 1024), (2701, 512), (3213, 256), (3469, 128)):
    ld = r8(3597) + 8
    A = hz(B, ld); WT = hz(N, r8(K)); bias = torch.zeros(16 * N, device="cuda")
    C = hz(B, ld); CT = hz(N, B)
    if "nn" in which:
        us = t(lambda: call("wd_hgemm_nn", ptr(A), ld, ptr(WT), r8(K), ptr(bias), 16, 1, ptr(C), ld, ptr(CT), B, B, N, K, st))
        print("NN  M %5d N %5d K %5d  %7.1f us  %6.1f TFLOP/s" % (B, N, K, us, 2.0 * B * N * K / us / 1e6), flush=True)
    if "tn" in which:
        AT = hz(K, B); ZT = hz(N, B)
        ns = max(1, min(-(-512 // (-(-(K + 1) // 128) * -(-N // 128))), 64, B // 256))
        Gp = torch.zeros(ns * (K + 1) * N, device="cuda")
        us = t(lambda: call("wd_hgemm_tn_splitk", ptr(AT), B, ptr(ZT), B, ptr(Gp), K, N, B, ns, st))
        print("TN  M %5d N %5d K %5d  %7.1f us  %6.1f TFLOP/s (nsplit %d)" % (K + 1, N, B, us, 2.0 * B * N * K / us / 1e6, ns), flush=True)
    if "dz" in which:
        Z = hz(B, r8(N)); W = hz(K, r8(N)); Dh = hz(B, ld); DT = hz(K, B); act = hz(B, ld)
        us = t(lambda: call("wd_hgemm_nt", ptr(Z), r8(N), ptr(W), r8(N), B, K, N, None, 0, 0, ptr(Dh), ld, ptr(DT), B, ptr(act), ld, 1, st))
        print("NTd M %5d N %5d K %5d  %7.1f us  %6.1f TFLOP/s" % (B, K, N, us, 2.0 * B * N * K / us / 1e6), flush=True)
