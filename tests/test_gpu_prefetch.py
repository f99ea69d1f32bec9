"""GPU: the one-id-per-bag fast path of round 3 (csrc/onehot_path.hip + the patch in k_bucket_update):

  * wd_bucket_onehot (one launch) builds the same buckets as wd_sparse_bucketize (three launches);
  * the input layer as its own launch (wd_prefetch_onehot) + the tower reading x from HBM is BIT-identical to the tower that
    gathers its own x tile (WD_INPUT_AHEAD=0);
  * the pipelined multi-step hipGraph -- gather of batch t+1 BEFORE update(t), rewritten rows patched by update(t) -- is
    bit-identical to eager launches (gather after the update), also when consecutive batches share most of their rows
    (tiny vocabularies, Zipf ids: every second row is patched)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(spec, B, seed=3, **env):
    from wide_deep_amd.engine import WideDeepEngine
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return WideDeepEngine(spec, max_batch=B, seed=seed)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _state_equal(a, b, what):
    for name in ("rec", "emb_acc", "bias", "P", "Pacc", "logit", "loss", "wide_logit"):
        x, y = getattr(a, name), getattr(b, name)
        assert torch.equal(x, y), "%s: %s differs (max |d| %.3g)" % (what, name, float((x - y).abs().max()))


@pytest.mark.parametrize("buckets,dist", [(1_000_000, "uniform"), (20000, "zipf"), (40, "uniform"), (3, "uniform")])
def test_bucket_onehot_builds_the_buckets_of_bucketize(buckets, dist):
    from wide_deep_amd import synth
    from wide_deep_amd.plan import criteo_spec
    B = 4096
    spec = criteo_spec(n_dense=4, n_sparse=5, buckets=buckets, dim=16, hidden=(32, 32))
    eng = _engine(spec, B, WD_FLAT_UPDATE=0)
    assert eng.rec is not None
    hb = synth.make_raw_batch(eng.plan, B, seed=11, dist=dist)
    bt = synth.to_device_ids(eng.plan, hb)
    assert bt.one_hot and eng._bucket_onehot_ok(bt)
    st = torch.cuda.current_stream().cuda_stream
    eng._sparse_bucketize(bt, st, 0)                      # one launch
    os.environ["WD_BUCKET_ONEHOT"] = "0"
    try:
        eng._sparse_bucketize(bt, st, 1)                  # hist -> colscan -> scatter
    finally:
        del os.environ["WD_BUCKET_ONEHOT"]
    torch.cuda.synchronize()
    a, b = eng._bucket_sets[0], eng._bucket_sets[1]
    assert a["unsorted"] and not b["unsorted"]
    nb = eng.n_buckets
    sa, sb = a["start"][: nb + 1].cpu().numpy(), b["start"][: nb + 1].cpu().numpy()
    assert np.array_equal(sa, sb), "bucket starts differ"
    pa, pb = a["pairs"][: bt.nnz].cpu().numpy().view(np.uint64), b["pairs"][: bt.nnz].cpu().numpy().view(np.uint64)
    for k in range(nb):                                   # same pairs per bucket (arrival order is free)
        assert np.array_equal(np.sort(pa[sa[k]: sa[k + 1]]), np.sort(pb[sb[k]: sb[k + 1]])), "bucket %d" % k
    order = a["start"][nb + 2: 2 * nb + 2].cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(nb)), "launch order is not a permutation of the buckets"
    size = (sa[1:] - sa[:-1])[order]
    cls = np.where(size > 0, np.floor(np.log2(np.maximum(size, 1))) + 1, 0)
    assert np.all(cls[:-1] >= cls[1:]), "launch order is not largest-class-first"


@pytest.mark.parametrize("buckets,dist", [(200000, "uniform"), (20000, "zipf"), (1000000, "zipf"), (40, "uniform"), (3, "uniform")])
def test_bucket_sort_long_list_and_patch_list(buckets, dist):
    """wd_bucket_sort: every bucket sorted on (row, bag); the long list holds exactly the rows with more than 32 occurrences;
    the patch list of the previous batch points at the first pair of the same row in this batch (or -1)."""
    from wide_deep_amd import synth
    from wide_deep_amd.plan import criteo_spec
    B = 4096
    spec = criteo_spec(n_dense=16, n_sparse=5, buckets=buckets, dim=16, hidden=(32, 32))     # K0 = 96
    eng = _engine(spec, B)
    assert eng.prefetch and eng.flat_update
    st = torch.cuda.current_stream().cuda_stream
    bts = [synth.to_device_ids(eng.plan, synth.make_raw_batch(eng.plan, B, seed=21 + i, dist=dist)) for i in range(2)]
    eng._sparse_bucketize(bts[0], st, 0)
    eng._sparse_bucketize(bts[1], st, 1, prev=0)
    torch.cuda.synchronize()
    S, nb = eng.plan.S, eng.n_buckets
    sets = []
    for k, bt in enumerate(bts):
        bs = eng._bucket_sets[k]
        assert bs["sorted"]
        start = bs["start"][: nb + 1].cpu().numpy()
        pairs = bs["pairs"][: bt.nnz].cpu().numpy().view(np.uint64)
        ids = bt.ids.cpu().numpy().astype(np.int64).reshape(B, S)
        rb = np.asarray(eng.plan.row_base, dtype=np.int64)
        want = np.sort(((ids + rb[None, :]).astype(np.uint64) << np.uint64(32)) | (np.arange(B * S, dtype=np.uint64).reshape(B, S)), axis=None)
        assert np.array_equal(pairs, want), "batch %d: pairs are not the globally (row, bag)-sorted list" % k     # buckets partition the rows in order
        keys = (pairs >> np.uint64(32)).astype(np.int64)
        heads = np.flatnonzero(np.r_[True, keys[1:] != keys[:-1]])
        lens = np.diff(np.r_[heads, len(keys)])
        ll = bs["long_list"].cpu().numpy()
        got = sorted((int(ll[2 + 2 * q]), int(ll[3 + 2 * q])) for q in range(int(ll[0])))
        assert got == sorted((int(h), int(n)) for h, n in zip(heads, lens) if n > 32), "long list of batch %d" % k
        sets.append((keys, heads, start))
    (k0, h0, _), (k1, h1, _) = sets
    patch = eng._bucket_sets[0]["patch"][: 2 * bts[0].nnz].cpu().numpy().reshape(-1, 2).astype(np.int64)
    len1 = np.diff(np.r_[h1, len(k1)])
    first1 = {int(k1[h]): (int(h), int(n)) for h, n in zip(h1, len1)}
    want = np.tile(np.asarray([[-1, 0]], dtype=np.int64), (len(k0), 1))
    for h in h0:
        want[h] = first1.get(int(k0[h]), (-1, 0))
    assert np.array_equal(patch, want), "patch list (position, count)"


@pytest.mark.parametrize("kw,B,dist", [
    (dict(n_dense=13, n_sparse=26, buckets=5000, dim=16, hidden=(256, 128, 64)), 1024, "uniform"),
    (dict(n_dense=13, n_sparse=26, buckets=5000, dim=16, hidden=(256, 128, 64)), 1000, "zipf"),
    (dict(n_dense=16, n_sparse=6, buckets=37, dim=8, hidden=(64, 32)), 512, "uniform"),     # one-row buckets, every row patched
    (dict(n_dense=0, n_sparse=8, buckets=700, dim=4, hidden=(32,)), 200, "zipf"),
])
def test_prefetched_input_and_patched_graph_are_bit_identical(kw, B, dist):
    """(a) eager: prefetch engine == engine whose tower gathers its own x tile, bit for bit; (b) the pipelined graph (gather a
    step ahead + patch) == eager launches, bit for bit, over 7 steps in graphs of 1, 2 and 4 steps."""
    from wide_deep_amd import synth
    from wide_deep_amd.pipeline import StepGraph, step_eager, warm
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(**kw)
    a = _engine(spec, B)                       # prefetch + patch, graph
    b = _engine(spec, B)                       # prefetch, eager launches
    c = _engine(spec, B, WD_INPUT_AHEAD=0)        # the tower gathers its own x tile
    assert a.prefetch and b.prefetch and not c.prefetch and a.chain and c.chain
    hbs = [synth.make_raw_batch(a.plan, B, seed=100 + i, dist=dist, pos_rate=0.3) for i in range(8)]
    tbs = {e: [synth.TokenBatch(e.plan, hb) for hb in hbs] for e in (a, b, c)}
    assert a._chain_input_ok(synth.hash_tokens(a, tbs[a][0]))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for e in (a, b, c):
            step_eager(e, tbs[e][0])           # step 0 eagerly everywhere (what a capture needs first)
    torch.cuda.synchronize()
    _state_equal(a, b, "step 0")
    _state_equal(b, c, "step 0, prefetched vs fused input layer")
    graphs = [StepGraph(a, tbs[a][1:2], stream=side), StepGraph(a, tbs[a][2:4], stream=side), StepGraph(a, tbs[a][4:8], stream=side)]
    assert all(g.pipelined for g in graphs)
    with torch.cuda.stream(side):
        for i in range(1, 8):
            step_eager(b, tbs[b][i])
            step_eager(c, tbs[c][i])
    for g in graphs:
        g.replay()
    torch.cuda.synchronize()
    _state_equal(b, c, "7 eager steps, prefetched vs fused input layer")
    _state_equal(a, b, "graph replay (prefetch a step ahead + patch) vs eager launches")
    assert a.global_step == b.global_step


@pytest.mark.parametrize("kw,B,dist", [
    (dict(n_dense=13, n_sparse=26, buckets=5000, dim=16, hidden=(256, 128, 64)), 1024, "zipf"),
    (dict(n_dense=16, n_sparse=6, buckets=37, dim=8, hidden=(64, 32)), 512, "uniform"),     # every row shared with the next batch
])
def test_chained_graphs_are_bit_identical(kw, B, dist):
    """Graphs chained through lookahead / phase / primed (the next graph's first batch is hashed, bucketed, sorted and gathered
    by the previous graph, whose last update patches it) == eager launches, bit for bit: a chain walked in order, a chain
    entered in the middle (the primed graph does its own input work), and a closed cycle replayed twice."""
    from wide_deep_amd import synth
    from wide_deep_amd.pipeline import StepGraph, step_eager
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(**kw)
    a, b = _engine(spec, B), _engine(spec, B)
    assert a.prefetch and b.prefetch
    hbs = [synth.make_raw_batch(a.plan, B, seed=300 + i, dist=dist, pos_rate=0.3) for i in range(7)]
    ta, tb_ = [synth.TokenBatch(a.plan, hb) for hb in hbs], [synth.TokenBatch(b.plan, hb) for hb in hbs]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step_eager(a, ta[0])
        step_eager(b, tb_[0])
    torch.cuda.synchronize()
    # cycle over batches 1..6: g1 = [1], g2 = [2, 3], g3 = [4, 5, 6]; 6 steps close the (set, buffer) phases
    g1 = StepGraph(a, ta[1:2], stream=side, lookahead=ta[2], phase=(0, 0), primed=True)
    g2 = StepGraph(a, ta[2:4], stream=side, lookahead=ta[4], phase=g1.next_phase, primed=True)
    g3 = StepGraph(a, ta[4:7], stream=side, lookahead=ta[1], phase=g2.next_phase, primed=True)
    assert g1.chained and g2.chained and g3.chained and g3.next_phase == (0, 0)
    assert g1.next_phase == (1, 1) and g2.next_phase == (0, 0)
    order = [1, 2, 3, 4, 5, 6]
    for g in (g1, g2, g3):
        g.replay()                                      # g1 primes itself, g2 and g3 find their input work in place
    assert a._primed is not None and a._primed[0] == id(ta[1])
    with torch.cuda.stream(side):
        for i in order:
            step_eager(b, tb_[i])
    torch.cuda.synchronize()
    _state_equal(a, b, "chain walked in order")
    for g in (g1, g2, g3):                              # the cycle again: g1 now continues from g3's lookahead
        g.replay()
    with torch.cuda.stream(side):
        for i in order:
            step_eager(b, tb_[i])
    torch.cuda.synchronize()
    _state_equal(a, b, "closed cycle, second round")
    g3.replay()                                         # entered in the middle: g3's input work is not in place (g1 is primed)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for i in (4, 5, 6):
            step_eager(b, tb_[i])
        step_eager(a, ta[0])                            # an eager step invalidates what g3 left for g1
        step_eager(b, tb_[0])
    torch.cuda.synchronize()
    g1.replay()
    with torch.cuda.stream(side):
        step_eager(b, tb_[1])
    torch.cuda.synchronize()
    _state_equal(a, b, "chain entered in the middle / after an eager step")
    assert a.global_step == b.global_step


@pytest.mark.parametrize("ragged", [False, True])
def test_row_update_counts_a_rows_occurrences_at_every_boundary(ragged):
    """k_row_update finds how often a row occurs from ONE round of loads behind its first sorted position (the four lanes load the
    next 32 pairs and count the key's prefix): rows with exactly 1, 2, 3, 4, 5, 31, 32 (the last one the flat part takes), 33 and 70
    (the long-row list) occurrences, the rows with the largest keys included -- their pairs end the sorted list, so the count must stop
    at the list's end -- against the oracle: one-id bags through the pipelined graph (wd_row_update + patch of the next batch) and
    ragged bags through eager steps (wd_row_update_ragged)."""
    from tests.helpers import assert_close, oracle_batch, oracle_from_engine
    from wide_deep_amd import synth
    from wide_deep_amd.pipeline import StepGraph, step_eager
    from wide_deep_amd.plan import criteo_spec
    B, S, NB = 256, 6, 400
    spec = criteo_spec(n_dense=16, n_sparse=S, buckets=NB, dim=8, hidden=(64, 32))
    eng = _engine(spec, B)
    assert eng.rec is not None and eng.prefetch
    counts = [1, 2, 3, 4, 5, 31, 32, 33, 70]

    def host_batch(seed):
        rng = np.random.default_rng(seed)
        lens = np.ones((B, S), dtype=np.int64)
        if ragged:
            lens[rng.random((B, S)) < 0.3] = 2
        nnz = int(lens.sum())
        slot_of = np.repeat(np.tile(np.arange(S), B), lens.reshape(-1))
        raw = np.empty(nnz, dtype=np.int64)
        for s in range(S):
            pos = np.flatnonzero(slot_of == s)
            rng.shuffle(pos)
            # the LAST rows of the slot (for the last slot: the largest keys of the whole sorted list) take the boundary counts,
            # the other occurrences fall on distinct low rows
            k = 0
            for j, c in enumerate(counts):
                raw[pos[k: k + c]] = NB - 1 - j
                k += c
            rest = pos[k:]
            raw[rest] = np.arange(len(rest)) % (NB - len(counts))
        return {"B": B, "lens": lens, "raw": raw, "dense": rng.standard_normal((B, 16)).astype(np.float32),
                "labels": (rng.random(B) < 0.4).astype(np.float32)}

    hbs = [host_batch(7 + i) for i in range(3)]
    tbs = [synth.FeaturizedBatch(synth.to_device_ids(eng.plan, hb), hb) for hb in hbs]
    assert all(tb.batch.one_hot == (not ragged) for tb in tbs)
    ora = oracle_from_engine(eng)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step_eager(eng, tbs[0], ids_input=True)
        if ragged:
            for tb in tbs[1:]:
                step_eager(eng, tb, ids_input=True)
    torch.cuda.synchronize()
    if ragged:
        bs = eng._bucket_sets[0]
        assert bs["sorted"] and bs.get("ragged")                       # the flat ragged update ran
    else:
        g = StepGraph(eng, tbs[1:3], ids_input=True, stream=side)       # gather a step ahead, flat update + patch
        assert g.pipelined and any(b["sorted"] for b in eng._bucket_sets)
        g.replay()
        torch.cuda.synchronize()
    for hb, tb in zip(hbs, tbs):
        ora.train_step(oracle_batch(eng.plan, tb.batch.ids.cpu().numpy().astype(np.int64), synth.offsets_from_lens(hb["lens"]), B,
                                    hb["dense"], hb["labels"]))
    st, so = eng.export_state(), ora.state
    n = 0
    for k in st:
        if st[k].dtype.is_floating_point and k in so:
            assert_close(st[k], so[k], 5e-4, 2e-5, k)
            n += 1
    assert n >= 2 * S
