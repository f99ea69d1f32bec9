"""GPU: the one-launch tower (wd_tower_chain, csrc/mlp_chain8.hip) against the CPU oracle and against the per-layer GEMM
launches it replaces (same engine with WD_CHAIN=0).  The parametrised shapes cover the kernel's phase kinds: whole phases of 8
column tiles, reduction-split phases (4 x 2, 2 x 4, 1 x 8 slices), units shorter than the weight ring and of lengths that are
no multiple of it, a ragged last row tile, a non-ReLU activation.

Tolerances: as tests/test_gpu_step.py (fp32 summation order is the only difference: MFMA k-order of a 32-row tile vs the
64x64 tiles of the GEMMs vs BLAS)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _engines(spec, max_batch, seed=5):
    from wide_deep_amd.engine import WideDeepEngine
    chain = WideDeepEngine(spec, max_batch=max_batch, seed=seed)
    os.environ["WD_CHAIN"] = "0"
    try:
        layered = WideDeepEngine(spec, max_batch=max_batch, seed=seed)
    finally:
        del os.environ["WD_CHAIN"]
    assert chain.chain and not layered.chain and chain.chain_rt == 32
    return chain, layered


@pytest.mark.parametrize("B,kw", [
    (100, dict(n_dense=8, n_sparse=3, buckets=500, dim=8, hidden=(64, 32))),               # K0 = 32, ragged last tile
    (96, dict(n_dense=0, n_sparse=4, buckets=300, dim=16, hidden=(32,))),                  # one hidden layer
    (257, dict(n_dense=13, n_sparse=26, buckets=2000, dim=16, hidden=(256, 128, 64))),     # BASELINE configs[1] tower
    (64, dict(n_dense=4, n_sparse=9, buckets=100, dim=16, hidden=(96, 160, 32, 32), activation="tanh")),
])
def test_chain_matches_per_layer_launches(B, kw):
    from wide_deep_amd import synth
    from wide_deep_amd.plan import criteo_spec
    from tests.helpers import assert_close
    kw = dict(kw)
    activation = kw.pop("activation", None)
    spec = criteo_spec(**kw)
    if activation:
        spec.activation = activation
    a, b = _engines(spec, max_batch=B + 7)
    for step in range(2):
        hb = synth.make_raw_batch(a.plan, B, seed=40 + step, pos_rate=0.3)
        bta, btb = synth.to_device_ids(a.plan, hb), synth.to_device_ids(b.plan, hb)
        la, lb = a.train_step(bta), b.train_step(btb)
        torch.cuda.synchronize()
        assert_close(a.logit[:B], b.logit[:B], 1e-5, 1e-5, "logits step %d" % step)
        assert_close(a.prob[:B], b.prob[:B], 1e-5, 1e-6, "prob")
        assert_close(a.dlogit[:B], b.dlogit[:B], 1e-5, 1e-6, "dlogit")
        assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb)))
        assert_close(a.G, b.G, 2e-4, 2e-6, "dense gradient step %d" % step)
    sa, sb = a.export_state(), b.export_state()
    for k in sb:
        assert_close(sa[k], sb[k], 2e-4, 2e-6, k)
    # forward only (evaluate / predict): no labels, nothing but logits and probabilities is written
    hb = synth.make_raw_batch(a.plan, B, seed=77)
    bta, btb = synth.to_device_ids(a.plan, hb), synth.to_device_ids(b.plan, hb)
    bta.labels = btb.labels = None
    a.forward(bta, need_loss=False); b.forward(btb, need_loss=False)
    torch.cuda.synchronize()
    assert_close(a.logit[:B], b.logit[:B], 1e-5, 1e-5, "predict logits")


def test_fused_input_layer_and_env_switch():
    """The tower kernel builds its own x tile (wd_chain_opts_t.input) for one-id-per-bag batches; WD_CHAIN_INPUT=0 keeps
    the separate input-layer launch; x is bit-equal between the two, and both train like the per-layer engine."""
    from wide_deep_amd import synth
    from wide_deep_amd.plan import criteo_spec
    from tests.helpers import assert_close
    spec = criteo_spec(n_dense=16, n_sparse=7, buckets=400, dim=16, hidden=(64, 32))   # K0 = 128
    a, b = _engines(spec, max_batch=80)
    hb = synth.make_raw_batch(a.plan, 77, seed=3, pos_rate=0.3)
    bts = [synth.to_device_ids(e.plan, hb) for e in (a, b)]
    assert a._chain_input_ok(bts[0]) and not b._chain_input_ok(bts[1])
    a.train_step(bts[0]); b.train_step(bts[1])
    os.environ["WD_CHAIN_INPUT"] = "0"
    try:
        c, _ = _engines(spec, max_batch=80)
        btc = synth.to_device_ids(c.plan, hb)
        assert not c._chain_input_ok(btc)
        c.train_step(btc)
    finally:
        del os.environ["WD_CHAIN_INPUT"]
    torch.cuda.synchronize()
    assert_close(a.logit[:77], b.logit[:77], 1e-5, 1e-5, "logits")
    assert_close(a.wide_logit[:77], b.wide_logit[:77], 1e-6, 1e-6, "wide logit")
    tl = a.towers[0]["layout"]
    assert torch.equal(a.towers[0]["act"][:77, : tl.seg_width[0]], c.towers[0]["act"][:77, : tl.seg_width[0]])   # x bit-equal
    sa, sb, sc = a.export_state(), b.export_state(), c.export_state()
    for k in sb:
        assert_close(sa[k], sb[k], 2e-4, 2e-6, k)
        assert_close(sa[k], sc[k], 2e-4, 2e-6, k)


def test_chain_steps_match_oracle():
    from tests.test_gpu_step import _run
    from wide_deep_amd.plan import criteo_spec
    eng, _ = _run(criteo_spec(n_dense=8, n_sparse=3, buckets=300, dim=8, hidden=(64, 32, 32)), B=96, steps=3)
    assert eng.chain
    eng, _ = _run(criteo_spec(n_dense=13, n_sparse=26, buckets=500, dim=16, hidden=(256, 128, 64)), B=200, steps=2,
                  max_batch=256)
    assert eng.chain


def test_chain_unsupported_shapes_fall_back_to_layer_launches():
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.plan import criteo_spec
    assert not WideDeepEngine(criteo_spec(n_dense=3, n_sparse=5, buckets=50, dim=16, hidden=(32, 16, 8)), max_batch=64).chain
    assert not WideDeepEngine(criteo_spec(n_dense=8, n_sparse=3, buckets=50, dim=8, hidden=(64, 32), mode="dense"),
                              max_batch=64).chain


@pytest.mark.parametrize("mode,B,kw", [
    ("resnet", 100, dict(n_dense=16, n_sparse=3, buckets=500, dim=16, hidden=(32, 32))),                   # K0 = 64 = sum of the widths
    ("dense", 70, dict(n_dense=16, n_sparse=7, buckets=300, dim=16, hidden=(64, 32, 32))),                 # K0 = 128
    ("resnet", 257, dict(n_dense=13, n_sparse=26, buckets=2000, dim=16, hidden=(256, 128, 64))),           # BASELINE configs[3] tower
    ("dense", 96, dict(n_dense=13, n_sparse=26, buckets=2000, dim=16, hidden=(256, 128, 64), activation="tanh")),
    ("resnet", 64, dict(n_dense=0, n_sparse=6, buckets=100, dim=16, hidden=(96,))),                        # one hidden layer
])
def test_concatenating_towers_in_the_one_launch_match_per_layer_launches_and_the_oracle(mode, B, kw):
    """dnn_connected_mode 'resnet' / 'dense' (python/lib/dnn.py:155-193: every layer reads the concat of what came before) in the
    one-launch tower -- row tile mirroring the activation row, pull over [dz_l | .. | dz_{L-1}] + the logits layer's rank-1 term,
    concatenated packed operands written by wd_chain_tail -- against the per-layer launches (WD_CHAIN=0) and the oracle's
    autograd, multi-hot bags, three steps."""
    from wide_deep_amd import synth
    from wide_deep_amd.plan import criteo_spec
    from tests.helpers import assert_close, oracle_batch, oracle_from_engine
    kw = dict(kw)
    activation = kw.pop("activation", None)
    spec = criteo_spec(mode=mode, **kw)
    if activation:
        spec.activation = activation
    a, b = _engines(spec, max_batch=B + 7)
    assert a.towers[0]["windows"] is not None and not a.prefetch
    ora = oracle_from_engine(a)
    for step in range(3):
        hb = synth.make_raw_batch(a.plan, B, seed=60 + step, pos_rate=0.3, mean_len=2)
        bta, btb = synth.to_device_ids(a.plan, hb), synth.to_device_ids(b.plan, hb)
        la, lb = a.train_step(bta), b.train_step(btb)
        torch.cuda.synchronize()
        oloss, ologits = ora.train_step(oracle_batch(a.plan, bta.ids.cpu().numpy(), bta.bag_offs.cpu().numpy(), B, hb["dense"], hb["labels"]))
        assert_close(a.logit[:B], b.logit[:B], 2e-5, 2e-5, "logits vs per-layer, step %d" % step)
        assert_close(a.logit[:B], ologits, 2e-4, 2e-5, "logits vs oracle, step %d" % step)
        assert_close(a.dlogit[:B], b.dlogit[:B], 1e-5, 1e-6, "dlogit")
        assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb))) and abs(float(la) - oloss) <= 2e-4 * max(1.0, abs(oloss))
        # step 0 starts from identical parameters; afterwards the two summation orders have been through Adagrad's g / sqrt(acc),
        # which turns a rounding-level difference of a near-zero gradient into a difference of the parameter
        assert_close(a.G, b.G, 2e-4, 2e-6 if step == 0 else 5e-5, "dense gradient step %d" % step)
        dxa = a.towers[0]["dact"][:B, a.towers[0]["layout"].seg_start[0]: a.towers[0]["layout"].seg_start[0] + a.towers[0]["dx_cols"]]
        dxb = b.towers[0]["dact"][:B, b.towers[0]["layout"].seg_start[0]: b.towers[0]["layout"].seg_start[0] + a.towers[0]["dx_cols"]]
        assert_close(dxa, dxb, 2e-4, 2e-6, "dx step %d" % step)
    sa, sb = a.export_state(), b.export_state()
    for k in sb:
        assert_close(sa[k], sb[k], 2e-4, 2e-6, k)
    for k, v in ora.state.items():
        if k != "global_step" and "moving_" not in k:
            assert_close(sa[k], v.detach(), 5e-4, 1e-5, "vs oracle: " + k)
    hb = synth.make_raw_batch(a.plan, B, seed=77, mean_len=2)
    bta, btb = synth.to_device_ids(a.plan, hb), synth.to_device_ids(b.plan, hb)
    bta.labels = btb.labels = None
    a.forward(bta, need_loss=False); b.forward(btb, need_loss=False)
    torch.cuda.synchronize()
    assert_close(a.logit[:B], b.logit[:B], 2e-5, 2e-5, "predict logits")
