"""GPU: every optimizer the reference accepts (python/lib/utils/model_util.py:84-90: Adagrad, Adam, Ftrl, RMSProp, SGD), on
either scope (dnn_optimizer: embeddings + tower variables; linear_optimizer: wide weights + bias), against the CPU oracle
-- whole train steps, same weights, same batches, duplicates inside the batch (tiny tables), several steps so that slot
state and Adam's beta powers matter.  Tolerances as tests/test_gpu_step.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SGD = ("SGD", 0.05)
ADAGRAD = ("Adagrad", 0.05, 0.1)
FTRL = ("Ftrl", 0.1, 0.5, 1.0, 0.1)
FTRL_DNN = ("Ftrl", 0.05, 0.001, 0.01, 0.1)
RMSPROP = ("RMSProp", 0.01, 0.9, 0.0, 1e-10)
RMSPROP_MOM = ("RMSProp", 0.01, 0.8, 0.5, 1e-6)
ADAM = ("Adam", 0.01, 0.9, 0.999, 1e-8)
FTRL_POW = ("Ftrl", 0.1, 0.5, 1.0, 0.1, -0.7)                  # learning_rate_power != -0.5: FtrlCompute's general branch
FTRL_POW_DNN = ("Ftrl", 0.05, 0.001, 0.01, 0.1, -0.3)
FTRL_SHRINK = ("Ftrl", 0.1, 0.5, 1.0, 0.1, -0.5, 0.05)         # l2_shrinkage_regularization_strength (round 4)
FTRL_SHRINK_POW_DNN = ("Ftrl", 0.05, 0.001, 0.01, 0.1, -0.3, 0.02)
RMSPROP_CENTERED = ("RMSProp", 0.01, 0.9, 0.0, 1e-6, True)     # ApplyCenteredRMSProp: third slot (mean gradient)
RMSPROP_CENTERED_MOM = ("RMSProp", 0.01, 0.8, 0.5, 1e-6, True)


def _spec(dnn, lin, **kw):
    from wide_deep_amd.plan import criteo_spec
    base = dict(n_dense=3, n_sparse=5, buckets=40, dim=16, hidden=(32, 16))   # 40-row tables: many duplicate rows
    base.update(kw)
    s = criteo_spec(**base)
    s.dnn_opt, s.lin_opt = dnn, lin
    return s


@pytest.mark.parametrize("dnn,lin", [(SGD, SGD), (RMSPROP, ADAGRAD), (ADAM, ADAM), (FTRL_DNN, RMSPROP_MOM), (ADAGRAD, ADAM),
                                     (RMSPROP_MOM, FTRL), (ADAM, FTRL), (ADAGRAD, FTRL_POW), (RMSPROP_CENTERED, FTRL_POW),
                                     (FTRL_POW_DNN, RMSPROP_CENTERED_MOM), (RMSPROP_CENTERED_MOM, RMSPROP_CENTERED),
                                     (ADAGRAD, FTRL_SHRINK), (FTRL_SHRINK_POW_DNN, FTRL_SHRINK)])
def test_train_steps_match_oracle(dnn, lin):
    from tests.test_gpu_step import _run
    eng, ora = _run(_spec(dnn, lin), B=96, steps=4)
    assert not eng.default_opts
    st = eng.export_state()
    if dnn[0] == "Adam":
        assert abs(float(st["beta1_power"]) - 0.9 ** 5) < 1e-6 and abs(float(st["beta2_power"]) - 0.999 ** 5) < 1e-6
    if dnn[0] == "Adam" and lin[0] == "Adam":
        assert "beta1_power_1" in st


def test_multi_hot_dims_and_modes():
    """mixed embedding dims (incl. one that is not a multiple of 4), multi-hot bags, a resnet tower, wide-only / deep-only"""
    from tests.test_gpu_step import _run
    from wide_deep_amd.plan import criteo_spec
    s = _spec(ADAM, RMSPROP, mode="resnet")
    for sl, d in zip(s.slots, (16, 8, 6, 16, 32)):
        sl.dim = d
    _run(s, B=64, steps=3, mean_len=3, dist="zipf")
    _run(_spec(ADAM, ADAM, model_type="wide"), B=64, steps=3)
    _run(_spec(RMSPROP_MOM, FTRL, model_type="deep"), B=64, steps=3)


def test_adam_moves_rows_without_gradient():
    """AdamOptimizer._apply_sparse_shared: after a row was hit once its m keeps pushing it in later steps that do not
    touch it (and the touched bitmap is left clean)."""
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    eng = WideDeepEngine(_spec(ADAM, ADAM, buckets=5000), max_batch=64, seed=1)
    hb0 = synth.make_raw_batch(eng.plan, 64, seed=1, pos_rate=0.5)
    hb1 = synth.make_raw_batch(eng.plan, 64, seed=2, pos_rate=0.5)
    eng.train_step(synth.to_device_ids(eng.plan, hb0))
    torch.cuda.synchronize()
    e1, w1 = eng.emb.clone(), eng.wide.clone()
    eng.train_step(synth.to_device_ids(eng.plan, hb1))
    torch.cuda.synchronize()
    assert int(eng.touched.abs().sum()) == 0
    s0 = eng.plan.slots[0]
    rows0 = set((hb0["raw"].reshape(64, -1)[:, 0] % s0.num_buckets).tolist())
    rows1 = set((hb1["raw"].reshape(64, -1)[:, 0] % s0.num_buckets).tolist())
    only0 = sorted(rows0 - rows1)
    never = sorted(set(range(s0.num_buckets)) - rows0 - rows1)[:50]
    assert only0 and never
    E1 = e1[: s0.num_buckets * s0.dim].view(s0.num_buckets, s0.dim)
    E2 = eng.emb[: s0.num_buckets * s0.dim].view(s0.num_buckets, s0.dim)
    assert bool((E1[only0] != E2[only0]).any(dim=1).all())      # momentum moved them without a gradient
    assert torch.equal(E1[never], E2[never])                    # m = v = 0: no movement
    W1, W2 = w1[:s0.num_buckets], eng.wide[:s0.num_buckets]
    assert bool((W1[only0, 0] != W2[only0, 0]).all()) and torch.equal(W1[never], W2[never])


def test_graph_replay_advances_beta_powers():
    """The captured step reads Adam's beta powers from HBM: replays equal eager steps."""
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    spec = _spec(ADAM, ADAM, buckets=300)
    a, b = WideDeepEngine(spec, max_batch=64, seed=2), WideDeepEngine(spec, max_batch=64, seed=2)
    hb = synth.make_raw_batch(a.plan, 64, seed=5, pos_rate=0.4)
    bta, btb = synth.to_device_ids(a.plan, hb), synth.to_device_ids(b.plan, hb)
    replay = a.capture_train_step(bta, warmup=1)      # one real (warm-up) step; the capture itself executes nothing
    b.train_step(btb)
    for _ in range(3):
        replay()
        b.train_step(btb)
    torch.cuda.synchronize()
    sa, sb = a.export_state(), b.export_state()
    for k in sb:
        if k != "global_step":
            assert torch.allclose(sa[k], sb[k], rtol=1e-5, atol=1e-7), k


def test_centered_rmsprop_slot_names_and_ftrl_power_zero_is_sgd():
    """TF numbers an optimizer's slot variables in creation order (rms, mg, momentum): centered=True moves the momentum slot to
    /RMSProp_2; and tf's ftrl_test.py equivalence: lr_power = 0 without regularisation from zero weights is plain SGD."""
    import ctypes
    from wide_deep_amd import capi, synth
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.plan import opt_params
    eng = WideDeepEngine(_spec(RMSPROP_CENTERED_MOM, RMSPROP_CENTERED, buckets=300), max_batch=64, seed=2)
    hb = synth.make_raw_batch(eng.plan, 64, seed=5, pos_rate=0.4)
    eng.train_step(synth.to_device_ids(eng.plan, hb))
    st = eng.export_state()
    k = "dnn/dnn_1/hiddenlayer_0/kernel"
    assert {k + "/RMSProp", k + "/RMSProp_1", k + "/RMSProp_2"} <= set(st)
    assert float(st[k + "/RMSProp_1"].abs().max()) > 0 and float(st[k + "/RMSProp_1"].abs().max()) < 1.0   # mg: small, signed
    assert float(st[k + "/RMSProp"].min()) > 0.5                                                            # rms: started at 1
    assert "linear/linear_model/bias_weights/RMSProp_2" in st
    b = WideDeepEngine(_spec(RMSPROP_CENTERED_MOM, RMSPROP_CENTERED, buckets=300), max_batch=64, seed=9)
    b.import_state(st)
    eng.train_step(synth.to_device_ids(eng.plan, hb)); b.train_step(synth.to_device_ids(b.plan, hb))
    torch.cuda.synchronize()
    sa, sb = eng.export_state(), b.export_state()
    for kk in sa:
        assert torch.equal(sa[kk], sb[kk]), kk
    # Ftrl(lr_power = 0, l1 = l2 = 0) from w = 0 == SGD, through wd_opt_dense
    n = 1000
    g = torch.Generator(device="cuda").manual_seed(3)
    w1, z, acc = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.full((n,), 0.1, device="cuda")
    w2 = torch.zeros(n, device="cuda")
    of, os_ = capi.WdOpt(), capi.WdOpt()
    of.kind, of.lr = capi.WD_OPT_KINDS["Ftrl"], 3.0
    of.p0, of.p1, of.p2, of.p3 = opt_params(("Ftrl", 3.0, 0.0, 0.0, 0.1, 0.0))
    os_.kind, os_.lr = capi.WD_OPT_KINDS["SGD"], 3.0
    for _ in range(3):
        gr = torch.randn(n, device="cuda", generator=g) * 0.1
        capi.call("wd_opt_dense", w1.data_ptr(), z.data_ptr(), acc.data_ptr(), gr.data_ptr(), n, ctypes.byref(of), None)
        capi.call("wd_opt_dense", w2.data_ptr(), None, None, gr.data_ptr(), n, ctypes.byref(os_), None)
    torch.cuda.synchronize()
    assert torch.allclose(w1, w2, rtol=1e-5, atol=1e-6)


def test_checkpoint_round_trip_restores_slots(tmp_path):
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    spec = _spec(RMSPROP_MOM, ADAM, buckets=300)
    a = WideDeepEngine(spec, max_batch=64, seed=2)
    hb = synth.make_raw_batch(a.plan, 64, seed=5, pos_rate=0.4)
    a.train_step(synth.to_device_ids(a.plan, hb))
    st = a.export_state()
    assert any(k.endswith("/RMSProp_1") for k in st) and any(k.endswith("/Adam") for k in st) and "beta1_power" in st
    b = WideDeepEngine(spec, max_batch=64, seed=9)
    b.import_state(st)
    a.train_step(synth.to_device_ids(a.plan, hb)); b.train_step(synth.to_device_ids(b.plan, hb))
    torch.cuda.synchronize()
    sa, sb = a.export_state(), b.export_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_learning_rates_set_per_step_train_like_the_oracle_with_the_same_rates():
    """engine.set_learning_rates (what Estimator.train calls before every step under the opt-in `lr_decay`): the launches that
    follow use the new rates -- default optimizers (specialised kernels) and generic ones (wd_opt_t) -- like an oracle whose
    optimizer tuples carry the same per-step rates."""
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from tests.helpers import oracle_batch, oracle_from_engine, assert_close
    from tests.test_gpu_step import L_RTOL, L_ATOL, P_RTOL, P_ATOL
    for dnn, lin, kw in ((ADAGRAD, FTRL, {}), (RMSPROP_MOM, FTRL_SHRINK, {}), (ADAGRAD, FTRL, dict(hidden=(64, 32), n_dense=16, n_sparse=3))):
        spec = _spec(dnn, lin, **kw)
        spec.lr_decay = {"dnn": (0.5, 4.0), "linear": (0.8, 4.0)}
        eng = WideDeepEngine(spec, max_batch=128, seed=3)
        ora = oracle_from_engine(eng)
        B = 96
        for step in range(4):
            gs = eng.global_step
            ld, ll = dnn[1] * 0.5 ** (gs / 4.0), lin[1] * 0.8 ** (gs / 4.0)
            eng.set_learning_rates(dnn=ld, linear=ll)
            ora.dnn_opt = (dnn[0], ld) + tuple(dnn[2:])
            ora.lin_opt = (lin[0], ll) + tuple(lin[2:])
            hb = synth.make_raw_batch(eng.plan, B, seed=200 + step, pos_rate=0.3)
            bt = synth.to_device_ids(eng.plan, hb)
            loss = eng.train_step(bt)
            torch.cuda.synchronize()
            oloss, ologits = ora.train_step(oracle_batch(eng.plan, bt.ids.cpu().numpy(), bt.bag_offs.cpu().numpy(), B, hb["dense"], hb["labels"]))
            assert_close(eng.logit[:B], ologits, L_RTOL, L_ATOL, "logits step %d" % step)
        assert eng.global_step == 12 and abs(eng.spec.dnn_opt[1] - dnn[1] * 0.5 ** (9 / 4.0)) < 1e-9
        st = eng.export_state()
        for k, v in ora.state.items():
            if k != "global_step" and "moving_" not in k:
                assert_close(st[k], v.detach(), P_RTOL, P_ATOL, k)
