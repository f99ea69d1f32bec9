#!/usr/bin/env python
"""Stand-alone timings (HIP events, launches back to back on one stream) of the one-id-per-bag kernels at BASELINE configs[1]
size: wd_bucket_onehot vs wd_sparse_bucketize, wd_prefetch_onehot, the row update with and without the patch of the next batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wide_deep_amd import synth
from wide_deep_amd.engine import WideDeepEngine
from wide_deep_amd.plan import criteo_spec

B = int(os.environ.get("PB", "8192"))
dist = os.environ.get("PDIST", "uniform")
spec = criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=16, hidden=(256, 128, 64), mode="simple")
eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * 26 * 4, seed=0)
tbs = [synth.TokenBatch(eng.plan, synth.make_raw_batch(eng.plan, B, seed=50 + i, dist=dist)) for i in range(8)]
for tb in tbs:
    synth.hash_tokens(eng, tb)
eng.train_step(tbs[0].batch)
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream


def timeit(name, fn, iters=50):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    e1.synchronize()
    print("%-64s %8.2f us" % (name, e0.elapsed_time(e1) / iters * 1e3), flush=True)


bts = [tb.batch for tb in tbs]
timeit("wd_hash_bucket_cols", lambda i: synth.hash_tokens(eng, tbs[i % 8]))
timeit("wd_bucket_onehot (slot-major ids)", lambda i: eng._sparse_bucketize(bts[i % 8], st, i & 1))
saved = [bt.ids_cols for bt in bts]
for bt in bts:
    bt.ids_cols = None
timeit("wd_bucket_onehot (example-major ids)", lambda i: eng._sparse_bucketize(bts[i % 8], st, i & 1))
for bt, c in zip(bts, saved):
    bt.ids_cols = c
os.environ["WD_BUCKET_ONEHOT"] = "0"
timeit("wd_sparse_bucketize (hist + colscan + scatter)", lambda i: eng._sparse_bucketize(bts[i % 8], st, i & 1))
del os.environ["WD_BUCKET_ONEHOT"]
if eng.flat_update:
    from wide_deep_amd.capi import call, ptr
    bsets = eng._bucket_sets
    def only_bucket(i):
        bs, bt = bsets[i & 1], bts[i % 8]
        call("wd_bucket_onehot", ptr(eng.slots_dev), eng.plan.S, ptr(bt.ids_cols), 1, bt.B, ptr(bs["start"]), ptr(bs["pairs"]),
             eng.n_buckets, eng.max_slot_buckets, None, ptr(bs["long_list"]), st)
    def only_sort(i, prev):
        bs, bp = bsets[i & 1], bsets[(i + 1) & 1]
        bs["long_list"][:2].zero_()
        call("wd_bucket_sort", ptr(bs["start"]), ptr(bs["pairs"]), eng.n_buckets, ptr(bs["long_list"]), (bs["long_list"].numel() - 2) // 2,
             ptr(bs["big_list"]), B, 26,
             ptr(bp["start"]) if prev else None, ptr(bp["pairs"]) if prev else None, ptr(bp["patch"]) if prev else None, st)
    timeit("  wd_bucket_onehot alone (no launch order)", only_bucket)
    eng._sparse_bucketize(bts[0], st, 0); eng._sparse_bucketize(bts[1], st, 1)
    timeit("  wd_bucket_sort alone (already sorted input), no patch list", lambda i: only_sort(i, False))
    timeit("  wd_bucket_sort alone (already sorted input), with the patch list of the previous batch", lambda i: only_sort(i, True))
    torch.cuda.synchronize()
    print("  long segments in batch 0: %d" % int(bsets[0]["long_list"][0]))
timeit("wd_prefetch_onehot", lambda i: eng._prefetch_input(bts[i % 8], st, i & 1))
timeit("wd_embag_fwd (records, gather only)", lambda i: eng.embag_fwd(16, eng.group_slots[16], bts[i % 8], eng._x_ptr(eng.towers[0]), eng.towers[0]["layout"].ld, st))
# update: bucket sets prepared for batch i (set 0) and batch i+1 (set 1)
eng._sparse_bucketize(bts[0], st, 0)
eng._sparse_bucketize(bts[1], st, 1)
eng._prefetch_input(bts[1], st, 1)
eng._sparse_bucketize(bts[1], st, 1, prev=0)
timeit("row update (flat: wd_row_update), no patch", lambda i: eng._sparse_backward(bts[0], st, bucketized=True, pset=0))
timeit("row update (flat: wd_row_update), patch of next", lambda i: eng._sparse_backward(bts[0], st, bucketized=True, pset=0, patch=(1, 1)))
os.environ["WD_BUCKET_ONEHOT"] = "0"
eng._sparse_bucketize(bts[0], st, 0)
del os.environ["WD_BUCKET_ONEHOT"]
timeit("wd_sparse_apply_rec (stable buckets of wd_sparse_bucketize)", lambda i: eng._sparse_backward(bts[0], st, bucketized=True, pset=0))
tw = eng.towers[0]
eng._apar = 0
eng._prefetch_input(bts[0], st, 0)
timeit("wd_tower_chain, x from HBM + wide weight list", lambda i: eng._tower_chain(tw, bts[0], B, st, True, True))
eng.prefetch = False
timeit("wd_tower_chain, fused input layer", lambda i: eng._tower_chain(tw, bts[i % 8], B, st, True, True))
