"""GPU: wd_gemm_tn_group_tail -- the weight-gradient launch that finishes the dense tail itself (the last workgroup of an
output tile sums the split-K partials, takes the Adagrad step, rewrites the packed kernels) -- against the two launches it
replaces (wd_gemm_tn_splitk_group + wd_chain_tail), bit for bit, over several steps (every step reads what the previous
one packed), eager and as a multi-step hipGraph."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(spec, B, **env):
    from wide_deep_amd.engine import WideDeepEngine
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return WideDeepEngine(spec, max_batch=B, seed=5)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _same(a, b, what):
    names = ["P", "Pacc", "G", "logit", "loss", "bias"] + (["rec", "emb_acc"] if a.rec is not None else ["emb", "wide"])
    for name in names:
        x, y = getattr(a, name), getattr(b, name)
        assert torch.equal(x, y), "%s: %s differs (max |d| %.3g)" % (what, name, float((x - y).abs().max()))
    for l, (x, y) in enumerate(zip(a.towers[0]["Wpk"] + a.towers[0]["WTpk"], b.towers[0]["Wpk"] + b.towers[0]["WTpk"])):
        assert torch.equal(x, y), "%s: packed kernel copy %d differs" % (what, l)


@pytest.mark.parametrize("kw,B", [
    (dict(n_dense=13, n_sparse=26, buckets=5000, dim=16, hidden=(256, 128, 64)), 8192),       # the C2 tower, full batch
    (dict(n_dense=16, n_sparse=5, buckets=300, dim=16, hidden=(32, 32)), 1000),                # half-empty tiles, ragged batch
    (dict(n_dense=0, n_sparse=10, buckets=700, dim=16, hidden=(96, 64, 32), batch_norm=False), 2048),
])
def test_products_launch_that_finishes_the_tail_is_bit_identical(kw, B):
    from wide_deep_amd import synth
    from wide_deep_amd.pipeline import StepGraph, step_eager
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(**kw)
    a, b = _engine(spec, B, WD_FUSE_TAIL=1), _engine(spec, B, WD_FUSE_TAIL=0)
    assert a.chain and b.chain and a._tail_fusable() and not b._tail_fusable()
    hbs = [synth.make_raw_batch(a.plan, B, seed=500 + i, pos_rate=0.3) for i in range(8)]
    ta, tb = [synth.TokenBatch(a.plan, hb) for hb in hbs], [synth.TokenBatch(b.plan, hb) for hb in hbs]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for i in range(3):
            step_eager(a, ta[i])
            step_eager(b, tb[i])
            torch.cuda.synchronize()
            assert a._tail_fused and not b._tail_fused
            _same(a, b, "eager step %d" % i)
    assert int(a._tile_counters.abs().sum()) == 0, "tile counters must return to zero"
    ga, gb = StepGraph(a, ta[3:8], stream=side), StepGraph(b, tb[3:8], stream=side)
    for _ in range(3):
        ga.replay()
        gb.replay()
    torch.cuda.synchronize()
    _same(a, b, "three replays of a 5-step graph")
    assert int(a._tile_counters.abs().sum()) == 0
