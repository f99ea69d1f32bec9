"""Device featurizer at the BASELINE configs[3] shape (26 hash slots x mean 5 tokens + two 200-bucket crossed columns over 2 and 3
slots: ~2.3 M ids per batch of 8192): the launches of features.Featurizer.run timed apart with HIP events, ids and bag offsets
checked bit-exact against the oracle (oracle/harness.parsed_batch_ids).  Diagnostics; `rocprofv3 --kernel-trace --stats` over this
script gives the per-kernel table of profiles/r6_c4_featurizer_*.

    python scripts/bench_featurizer.py [--batch 8192] [--iters 50] [--check]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--padding", default="ragged")
    args = ap.parse_args()
    import bench
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.features import Featurizer
    from wide_deep_amd.plan import FeaturePlan
    spec, mean_len = bench.make_spec("c4")
    for s in spec.slots:        # small tables: the featurizer does not touch them (keeps the engine's footprint down)
        if s.kind == "hash":
            s.num_buckets = 100_000
    gp = FeaturePlan(spec)
    B = args.batch
    raw, hb = synth.make_parsed_batch(gp, B, seed=20260925, mean_len=mean_len, weights=(0.99, 0.01))
    eng = WideDeepEngine(spec, max_batch=B, max_nnz=int(1.02 * hb["nnz"]) + 1024, seed=0)
    fz = Featurizer(eng, cross_padding=args.padding)
    pdb = fz.resident(raw, ids_capacity=eng.max_nnz)
    fz.run(pdb)
    bt = fz.finalize(pdb)
    out = {"batch": B, "nnz": bt.nnz, "tokens": pdb.T, "bags": B * gp.S}
    if args.check:
        from oracle.harness import parsed_batch_ids
        want, woffs = parsed_batch_ids(eng.plan, hb)
        out["bag_offs_bit_exact"] = bool(np.array_equal(bt.bag_offs.cpu().numpy(), woffs))
        out["ids_bit_exact"] = bool(np.array_equal(bt.ids.cpu().numpy()[: bt.nnz].astype(np.int64), want))
        # the sized entry point (Featurizer.to_device: host wait for the count) must give the same
        bt2 = fz.to_device(raw)
        out["to_device_equal"] = bool(bt2.nnz == bt.nnz and torch.equal(bt2.ids[: bt.nnz], bt.ids[: bt.nnz]) and torch.equal(bt2.bag_offs, bt.bag_offs))
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        e1.synchronize()
        return round(1e3 * e0.elapsed_time(e1) / args.iters, 2)

    out["us_head (fingerprint64 + lens + scan)"] = timed(lambda: fz._run_head(pdb, st))
    out["us_emit (wd_feat_emit)"] = timed(lambda: fz._run_emit(pdb, st))
    out["us_run (all launches)"] = timed(lambda: fz.run(pdb))
    out["G ids/s (emit)"] = round(bt.nnz / out["us_emit (wd_feat_emit)"] / 1e3, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
