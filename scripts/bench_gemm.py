#!/usr/bin/env python
"""Microbenchmark of the fp32 MFMA GEMM kernels at the C2 tower shapes (run on the GPU box).
Prints TFLOP/s per (kind, shape) from HIP-event timing; used under rocprofv3 --pmc for MFMA-busy counters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd.capi import call, ptr, load

load()
B = int(os.environ.get("GEMM_B", "8192"))
iters = int(os.environ.get("GEMM_ITERS", "50"))
st = torch.cuda.current_stream().cuda_stream
shapes = [(432, 256), (256, 128), (128, 64)]
if os.environ.get("GEMM_SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["GEMM_SHAPES"].split(",")]
if os.environ.get("GEMM_C5"):
    shapes = [(1680, 1024), (1024, 512), (512, 256), (256, 128)]


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


for (K, N) in shapes:
    ld = 896 if K <= 896 else K
    A = torch.randn(B, ld, device="cuda"); W = torch.randn(K, N, device="cuda") * 0.05
    bias = torch.zeros(16 * N, device="cuda"); C = torch.zeros(B, N, device="cuda")
    dz = torch.randn(B, N, device="cuda"); dA = torch.zeros(B, ld, device="cuda")
    tiles = -(-(K + 1) // 64) * -(-N // 64)
    ns = max(1, min(-(-512 // tiles), 64, max(1, B // 256)))
    Gp = torch.zeros(ns * (K + 1) * N, device="cuda")
    fl = 2.0 * B * K * N
    t = timeit(lambda: call("wd_gemm_nn_bias_act", ptr(A), ld, ptr(W), N, ptr(bias), 16, 1, ptr(C), N, B, N, K, st))
    print("NN  %5dx%5dx%5d  %8.2f us  %6.1f TF/s" % (B, N, K, t, fl / t / 1e6))
    t = timeit(lambda: call("wd_gemm_nt", ptr(dz), N, ptr(W), N, ptr(dA), ld, B, K, N, 0, st))
    print("NT  %5dx%5dx%5d  %8.2f us  %6.1f TF/s" % (B, K, N, t, fl / t / 1e6))
    t = timeit(lambda: call("wd_gemm_tn_splitk", ptr(A), ld, ptr(dz), N, ptr(Gp), K, N, B, ns, 1, st))
    print("TN  %5dx%5dx%5d  %8.2f us  %6.1f TF/s  (nsplit %d)" % (K + 1, N, B, t, fl / t / 1e6, ns))
