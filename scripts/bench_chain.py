#!/usr/bin/env python
"""Microbenchmark of the one-launch tower (wd_tower_chain) at the C2 tower shape (run on the GPU box): HIP-event time of
forward-only / forward + gradient chain without dx / full, so that stage groups can be priced; used under rocprofv3 --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd import synth
from wide_deep_amd.engine import WideDeepEngine
from wide_deep_amd.plan import criteo_spec

B = int(os.environ.get("CHAIN_B", "8192"))
iters = int(os.environ.get("CHAIN_ITERS", "50"))
hidden = tuple(int(v) for v in os.environ.get("CHAIN_HIDDEN", "256,128,64").split(","))
spec = criteo_spec(buckets=1000, hidden=hidden, mode=os.environ.get("CHAIN_MODE", "simple"))
eng = WideDeepEngine(spec, max_batch=B, seed=1)
assert eng.chain
hb = synth.make_raw_batch(eng.plan, B, seed=3, pos_rate=0.3)
bt = synth.to_device_ids(eng.plan, hb)
eng.train_step(bt)
torch.cuda.synchronize()
tw = eng.towers[0]
st = torch.cuda.current_stream().cuda_stream


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


fl_f = 2.0 * B * sum(m["K"] * m["N"] for m in tw["metas"][:-1])
fuse = os.environ.get("CHAIN_FUSE", "1") == "1" and eng._chain_input_ok(bt)    # input layer built inside the kernel
fused = timeit(lambda: eng._tower_chain(tw, bt, B, st, True, fuse)) if fuse else float("nan")
full = timeit(lambda: eng._tower_chain(tw, bt, B, st, True))
dxc = tw["dx_cols"]
tw["dx_cols"] = 0
nodx = timeit(lambda: eng._tower_chain(tw, bt, B, st, True))
tw["dx_cols"] = dxc
fwd = timeit(lambda: eng._tower_chain(tw, bt, B, st, False))
print("chain B=%d hidden=%s: %s %.1f us | x from HBM + given wide logit: full %.1f us, no dx %.1f us, forward only %.1f us "
      "(forward GEMM flops %.2f G)" % (B, hidden, "as launched in the step (x from HBM, wide logit from the prefetched weight list)"
                                       if eng.prefetch else "with fused input layer", fused, full, nodx, fwd, fl_f / 1e9))


stamps = torch.zeros(64 + 128, dtype=torch.int64, device="cuda")
eng._chain_stamps = stamps.data_ptr()
eng._tower_chain(tw, bt, B, st, True, fuse)
torch.cuda.synchronize()
eng._chain_stamps = None
names = ["x tile"] + ["F%d" % l for l in range(len(hidden))] + ["head"] + ["B%d" % l for l in range(len(hidden) - 1, 0, -1)] + ["dx"]
for wg, off in ((0, 0), (100, 32)):
    v = stamps[off: off + len(names) + 1].cpu().tolist()
    print("workgroup %d cycles: " % wg + ", ".join("%s %d" % (n, v[i + 1] - v[i]) for i, n in enumerate(names)) + ", total %d" % (v[len(names)] - v[0]))

# two-wavefronts-per-SIMD kernel: per-stage stamps of workgroup 0's wavefronts 0 and 4 (mlp_chain8.hip stage8)
sv = stamps[64:192].cpu().view(8, 2, 8)
if int(sv.abs().sum()):
    snames = ["F%d" % l for l in range(len(hidden))] + ["B%d" % l for l in range(len(hidden) - 1, 0, -1)] + ["dx"]
    for si, nm in enumerate(snames):
        for wi, wv_ in enumerate((0, 4)):
            v = sv[si, wi].tolist()
            if v[0]:
                print("  %s wave %d: first unit mma %d, ring complete +%d, to epilogue/barrier +%d, epilogue +%d, rest of stage + end barrier +%d"
                      % (nm, wv_, v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], v[5] - v[4]))


# in-step gather span from the per-tile realtime stamps (100 MHz): max(x tile ready) - min(start) over all workgroups
nt = (B + eng.chain_rt - 1) // eng.chain_rt
ts = torch.zeros(2 * nt, dtype=torch.int64, device="cuda")
eng._chain_tile_stamps = ts.data_ptr()
for rep in range(3):
    eng._tower_chain(tw, bt, B, st, True, fuse)
    torch.cuda.synchronize()
    v = ts.cpu().view(nt, 2)
    print("tile stamps (rep %d): start spread %.2f us, gather span (first start -> last x tile ready) %.2f us, median per-tile %.2f us"
          % (rep, (v[:, 0].max() - v[:, 0].min()).item() / 100.0, (v[:, 1].max() - v[:, 0].min()).item() / 100.0,
             (v[:, 1] - v[:, 0]).float().median().item() / 100.0))
eng._chain_tile_stamps = None
