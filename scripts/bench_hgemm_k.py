import os, sys
sys.path.insert(0, "/root/repo")
import torch
from wide_deep_amd.capi import call, ptr, load
load()
st = torch.cuda.current_stream().cuda_stream
B = 8192
def hz(r, c): return (torch.randn(r * c + 64, device="cuda") * 0.1).half()[: r * c].view(r, c)
def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
N = 1024
for K in (64, 256, 512, 1024, 1677, 3328):
    ld = 3608
    A = hz(B, ld); WT = hz(N, (K + 7) // 8 * 8); bias = torch.zeros(16 * N, device="cuda"); C = hz(B, ld); CT = hz(N, B)
    us = t(lambda: call("wd_hgemm_nn", ptr(A), ld, ptr(WT), (K + 7) // 8 * 8, ptr(bias), 16, 1, ptr(C), ld, ptr(CT), B, B, N, K, st))
    us2 = t(lambda: call("wd_hgemm_nn", ptr(A), ld, ptr(WT), (K + 7) // 8 * 8, ptr(bias), 16, 1, ptr(C), ld, None, 0, B, N, K, st))
    print("NN K %5d: %7.1f us (%6.1f TF/s) | without the transposed copy %7.1f us" % (K, us, 2.0 * B * N * K / us / 1e6, us2), flush=True)
