#!/usr/bin/env python
"""At which shader clock does the tower kernel run -- alone, and inside the pipelined step?  Two workgroups of wd_tower_chain stamp
the shader-cycle counter (s_memtime) and the 100 MHz realtime clock at their start and end (wd_chain_opts_t.stamps): cycles per
microsecond = the clock the launch really ran at.  If the step were power / clock limited (kernels beside the tower pulling the
clock down), the in-step figure would be lower than the stand-alone one.  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd import pipeline, synth
from wide_deep_amd.engine import WideDeepEngine
from wide_deep_amd.plan import criteo_spec

B = 8192
spec = criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=16, hidden=(256, 128, 64), mode="simple")
eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * 26 * 4, seed=0)
tbs = [synth.TokenBatch(eng.plan, synth.make_raw_batch(eng.plan, B, seed=20260925 + i)) for i in range(16)]
stamps = torch.zeros(64 + 128, dtype=torch.int64, device="cuda")


NAMES = ["x tile", "F0", "F1", "F2", "head", "B2", "B1", "dx"]


def report(tag):
    v = stamps.cpu().tolist()
    for wg, o in ((0, 0), (100, 32)):
        last = int(v[o + 27])
        cyc, rt = v[o + last] - v[o], (v[o + 29] - v[o + 28]) / 100.0
        stages = ", ".join("%s %d" % (NAMES[i] if i < len(NAMES) else "s%d" % i, v[o + i + 1] - v[o + i]) for i in range(last))
        print("%-34s workgroup %3d: %7d cycles in %6.2f us = %.0f MHz | %s" % (tag, wg, cyc, rt, cyc / max(rt, 1e-9), stages))


side = pipeline.warm(eng, tbs)
tw = eng.towers[0]
bt = tbs[0].batch
eng._chain_stamps = stamps.data_ptr()
st = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    eng._tower_chain(tw, bt, B, st, True, False) if eng.prefetch else eng._tower_chain(tw, bt, B, st, True, True)
    torch.cuda.synchronize()
report("tower alone (eager launch)")
# back to back, 200 launches: the clock under sustained MFMA load
for rep in range(200):
    eng._tower_chain(tw, bt, B, st, True, False) if eng.prefetch else eng._tower_chain(tw, bt, B, st, True, True)
torch.cuda.synchronize()
report("tower alone, 200th back to back")
# in the step: the chained multi-step graphs bench.py times, captured with the stamp buffer attached
runner = pipeline.StepRunner(eng, tbs, steps=20, stream=side)
runner.warm_up()
for rep in range(3):
    runner.run(20)
    torch.cuda.synchronize()
    report("tower in the step (replay %d)" % rep)
eng._chain_stamps = None
