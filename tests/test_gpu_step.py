"""GPU: whole train steps of the HIP engine against the CPU oracle (same weights, same batches).

Tolerance (north_star: "logits within stated fp32 tolerance"): the logits differ from the oracle only by
fp32 summation order (MFMA k-order vs BLAS blocking, bag-sum order), so we require
|logit - oracle| <= 2e-4 + 2e-4*|oracle| after several optimizer steps, and the trained tables / weights
within 5e-4 relative (+1e-5 absolute)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

L_RTOL, L_ATOL = 2e-4, 2e-4
P_RTOL, P_ATOL = 5e-4, 1e-5


def _run(spec, B=96, steps=3, mean_len=1, weights=False, dist="uniform", max_batch=128):
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from tests.helpers import oracle_batch, oracle_from_engine, assert_close
    eng = WideDeepEngine(spec, max_batch=max_batch, seed=3)
    ora = oracle_from_engine(eng)
    for step in range(steps):
        hb = synth.make_raw_batch(eng.plan, B, seed=100 + step, mean_len=mean_len, dist=dist, pos_rate=0.3)
        w = None
        if weights:
            w = np.where(hb["labels"] > 0, spec.pos_weight, spec.neg_weight).astype(np.float32)
        bt = synth.to_device_ids(eng.plan, hb, weights=w)
        masks = eng.dropout_masks(B)         # keep decisions the step is about to use (None without dnn_dropout)
        loss = eng.train_step(bt)
        torch.cuda.synchronize()
        ob = oracle_batch(eng.plan, bt.ids.cpu().numpy(), bt.bag_offs.cpu().numpy(), B, hb["dense"], hb["labels"], w)
        if masks is not None:
            ob["dropout_masks"] = masks
        oloss, ologits = ora.train_step(ob)
        assert_close(eng.logit[:B], ologits, L_RTOL, L_ATOL, "logits step %d" % step)
        assert abs(float(loss) - oloss) <= 1e-3 * max(1.0, abs(oloss)), (float(loss), oloss)
    st = eng.export_state()
    assert eng.state_shapes() == {k: tuple(v.shape) for k, v in st.items()}      # what a restore checks a checkpoint against
    for k, v in ora.state.items():
        if k == "global_step" or "moving_" in k:
            continue
        assert_close(st[k], v.detach(), P_RTOL, P_ATOL, k)
    return eng, ora


def _spec(**kw):
    from wide_deep_amd.plan import criteo_spec
    base = dict(n_dense=3, n_sparse=5, buckets=300, dim=16, hidden=(32, 16, 8))
    base.update(kw)
    return criteo_spec(**base)


@pytest.mark.parametrize("mode", ["simple", "dense", "resnet", "last_dense", "first_dense"])
def test_train_steps_match_oracle_modes(mode):
    _run(_spec(mode=mode))


@pytest.mark.parametrize("hidden", [(16,), (32, 16), (32, 16, 8, 12), (16, 8, 8, 4, 12)])
def test_first_dense_depths(hidden):
    """first_dense (python/lib/dnn.py:117-133): 1-5 hidden layers -> 0, 0, 1 and 2 extra copies of the deep input."""
    _run(_spec(mode="first_dense", hidden=hidden))


@pytest.mark.parametrize("conn,hidden", [((("0-1"), ("0-3"), ("1-2")), (32, 16, 8)), ("0-2", (16, 8)), ("0-1,1-2,2-3", (16, 8, 12)),
                                         ([(0, 3), (1, 3), (2, 3)], (8, 4, 12)), (["0-1", "0-2", "1-2", "0-4", "2-4"], (16, 8, 8, 4))])
def test_connection_list_train_steps_match_oracle(conn, hidden):
    """Connection lists (python/lib/dnn.py:65-66, 195-224): layer j reads [net_i, i -> j ascending | h_{j-1}] with net_i such a
    concat itself -- windows of copies, gradients of a repeated segment summed back into it."""
    eng, _ = _run(_spec(mode=conn, hidden=hidden))
    assert eng.towers[0]["layout"].mode == "list" and not eng.chain


def test_connection_list_chain_equals_dense_and_survives_dropout_and_two_towers():
    from wide_deep_amd.plan import TowerSpec
    a, _ = _run(_spec(mode="dense", hidden=(16, 8, 8)))
    b, _ = _run(_spec(mode="0-1,1-2,2-3", hidden=(16, 8, 8)))
    sa, sb = a.export_state(), b.export_state()          # same seed, same concat order -> the same model
    for k in sa:
        if k != "global_step":
            assert torch.allclose(sa[k], sb[k], rtol=1e-5, atol=1e-6), k
    s = _spec(mode=["0-1", "0-2"], hidden=(16, 8))
    s.dropout, s.activation = 0.25, "tanh"
    _run(s)
    s = _spec(mode="simple", hidden=(16, 8))
    s.towers = [TowerSpec([16, 8], ((0, 2),)), TowerSpec([8, 8, 4], ((0, 1), (1, 3))), TowerSpec([12], "simple")]
    _run(s, mean_len=3, dist="zipf")


@pytest.mark.parametrize("mode,rate,act", [("simple", 0.3, "relu"), ("resnet", 0.1, "tanh"), ("first_dense", 0.5, "relu"),
                                           ("dense", 0.2, "sigmoid")])
def test_dropout_train_steps_match_oracle(mode, rate, act):
    """dnn_dropout (python/lib/dnn.py:111-112): mask after the activation, before BN, TRAIN only; the oracle is driven by
    the engine's keep decisions (a stateless function of seed, step, layer, element), prediction stays mask-free."""
    from wide_deep_amd import synth
    s = _spec(mode=mode, hidden=(32, 16, 8))
    s.dropout, s.activation = rate, act
    eng, ora = _run(s, steps=3)
    assert eng.dropout == rate and not eng.chain
    m0 = eng.dropout_masks(96)
    keep = np.mean([m.mean() for m in m0[0]])
    assert abs(keep - (1 - rate)) < 0.06                    # Bernoulli(keep_prob)
    # evaluate / predict: no dropout
    hb = synth.make_raw_batch(eng.plan, 96, seed=999)
    bt = synth.to_device_ids(eng.plan, hb)
    from tests.helpers import oracle_batch, assert_close
    bt.labels = None
    eng.forward(bt, need_loss=False)
    torch.cuda.synchronize()
    ob = oracle_batch(eng.plan, bt.ids.cpu().numpy(), bt.bag_offs.cpu().numpy(), 96, hb["dense"], hb["labels"], None)
    logits, _ = ora.predict(ob)
    assert_close(eng.logit[:96], logits, L_RTOL, L_ATOL, "predict logits")


def test_dropout_graph_replay_draws_new_masks():
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    s = _spec(mode="simple", hidden=(32, 16))
    s.dropout = 0.4
    a, b = WideDeepEngine(s, max_batch=64, seed=4), WideDeepEngine(s, max_batch=64, seed=4)
    hb = synth.make_raw_batch(a.plan, 64, seed=5, pos_rate=0.4)
    bta, btb = synth.to_device_ids(a.plan, hb), synth.to_device_ids(b.plan, hb)
    replay = a.capture_train_step(bta, warmup=1)
    b.train_step(btb)
    for _ in range(3):
        replay()
        b.train_step(btb)
    torch.cuda.synchronize()
    assert int(a.drop_seed[1]) == 4 == int(b.drop_seed[1])
    sa, sb = a.export_state(), b.export_state()
    for k in sb:
        if k != "global_step":
            assert torch.allclose(sa[k], sb[k], rtol=1e-5, atol=1e-7), k


def test_multi_hot_zipf_with_weight_column():
    _run(_spec(mode="resnet", use_weight_column=True), mean_len=5, weights=True, dist="zipf")


def test_wide_only_and_deep_only():
    _run(_spec(model_type="wide"))
    _run(_spec(model_type="deep", mode="dense", hidden=(24, 12)))


def test_no_batch_norm_and_other_activation():
    s = _spec(batch_norm=False)
    s.activation = "tanh"
    _run(s)


@pytest.mark.parametrize("mode,hidden,dnn_opt,dropout", [
    ("simple", (32, 16, 8), None, 0.0), ("dense", (24, 12), None, 0.0), ("resnet", (16, 8), ("RMSProp", 0.01, 0.8, 0.5, 1e-6), 0.0),
    ("first_dense", (16, 8, 8), ("RMSProp", 0.01, 0.8, 0.5, 1e-6, True), 0.2), ("last_dense", (16, 12), ("Ftrl", 0.05, 0.001, 0.01, 0.1), 0.0)])
def test_crelu_train_steps_match_oracle(mode, hidden, dnn_opt, dropout):
    """activation `crelu` (python/lib/utils/model_util.py:52, tf.nn.crelu = concat(relu(z), relu(-z))): every layer hands 2N
    features to its BN and to the next layer; the TF variables stay kernel [K, N] / bias [N].  The engine runs a relu layer of
    width 2N with tied halves; the oracle computes tf.nn.crelu literally (torch autograd)."""
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    s = _spec(mode=mode, hidden=hidden)
    s.activation = "crelu"
    if dnn_opt:
        s.dnn_opt = dnn_opt
    if dropout:
        s.dropout = dropout
    eng, ora = _run(s, steps=4)
    assert eng.crelu and not eng.chain
    st = eng.export_state()
    # TF shapes: kernel of layer 1 has 2 * hidden[0] (+ whatever the mode concatenates) rows and hidden[1] columns
    k0, k1 = st["dnn/dnn_1/hiddenlayer_0/kernel"], st["dnn/dnn_1/hiddenlayer_1/kernel"]
    assert k0.shape[1] == hidden[0] and k1.shape[1] == hidden[1] and st["dnn/dnn_1/hiddenlayer_0/bias"].shape == (hidden[0],)
    assert st["dnn/dnn_1/hiddenlayer_0/batch_normalization/gamma"].shape == (2 * hidden[0],)
    if mode in ("simple", "last_dense"):
        assert k1.shape[0] == 2 * hidden[0]
    # the mirrored halves stayed exact mirrors (weights odd, accumulators even)
    m = eng.towers[0]["metas"][0]
    W = eng.P[m["w_off"]: m["w_off"] + m["K"] * m["N"]].view(m["K"], m["N"])
    assert torch.equal(W[:, :m["N_tf"]], -W[:, m["N_tf"]:]) and float(W.abs().max()) > 0
    if eng.Pacc is not None and s.dnn_opt[0] in ("Adagrad", "Ftrl"):
        A = eng.Pacc[m["w_off"]: m["w_off"] + m["K"] * m["N"]].view(m["K"], m["N"])
        assert torch.equal(A[:, :m["N_tf"]], A[:, m["N_tf"]:])
    # checkpoint round trip: a fresh engine restored from the TF-shaped state continues identically
    b = WideDeepEngine(s, max_batch=128, seed=11)
    b.import_state(st)
    hb = synth.make_raw_batch(eng.plan, 96, seed=777, pos_rate=0.3)
    eng.train_step(synth.to_device_ids(eng.plan, hb)); b.train_step(synth.to_device_ids(b.plan, hb))
    torch.cuda.synchronize()
    if not dropout:       # (dropout: the two engines draw from different seeds)
        sa, sb = eng.export_state(), b.export_state()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("hidden,mean_len,dim", [((64, 32), 1, 16), ((32, 16, 8), 1, 8), ((32, 16), 3, 16), ((64, 32), 1, 4)])
def test_row_record_layout_trains_like_separate_tables(hidden, mean_len, dim):
    """Two engines from the same seed, one on the row-record layout (embedding row + wide line {w, z, n} of a fused row in
    ONE record) and one on separate tables: same batches -> same logits and the same tables, accumulators and tower
    parameters after every step.  One id per bag + widths % 32 == 0: the one-launch tower gathers the records itself."""
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    spec = _spec(hidden=hidden, dim=dim, n_dense=3 if dim != 4 else 13)
    a = WideDeepEngine(spec, max_batch=128, seed=5, row_records=True)
    b = WideDeepEngine(spec, max_batch=128, seed=5, row_records=False)
    assert a.rec is not None and b.rec is None and a.rec.shape[1] == {4: 8, 8: 16, 16: 32}[dim]
    sa, sb = a.export_state(), b.export_state()
    for k in sb:
        assert torch.equal(sa[k], sb[k]), k           # same initial numbers in both layouts
    for step in range(4):
        hb = synth.make_raw_batch(a.plan, 96, seed=40 + step, mean_len=mean_len, pos_rate=0.3, dist="zipf" if step == 2 else "uniform")
        la = a.train_step(synth.to_device_ids(a.plan, hb))
        lb = b.train_step(synth.to_device_ids(b.plan, hb))
        torch.cuda.synchronize()
        if mean_len == 1:
            assert torch.equal(a.logit[:96], b.logit[:96]), "logits step %d" % step
            assert float(la) == float(lb)
        else:       # multi-hot: different gather kernels (strided generic / fused range), same sums up to the last bit
            assert torch.allclose(a.logit[:96], b.logit[:96], rtol=1e-5, atol=1e-6), "logits step %d" % step
    sa, sb = a.export_state(), b.export_state()
    for k in sb:
        if mean_len == 1:
            assert torch.equal(sa[k], sb[k]), k
        else:
            assert torch.allclose(sa[k].float(), sb[k].float(), rtol=1e-4, atol=1e-6), k
    sb = sa
    # pad floats of the records are never written
    D = dim
    assert float(a.rec[:, D + 4:].abs().max() if a.rec.shape[1] > D + 4 else 0.0) == 0.0
    # checkpoint round trip across layouts
    c = WideDeepEngine(spec, max_batch=128, seed=9, row_records=True)
    c.import_state(sb)
    sc = c.export_state()
    for k in sb:
        assert torch.equal(sc[k], sb[k]), k


def test_tf_checkpoint_container_round_trip_through_the_engine(tmp_path):
    """engine state -> TensorFlow's checkpoint container (tf_checkpoint.py) -> a fresh engine: identical training afterwards"""
    from wide_deep_amd import synth, tf_checkpoint as T
    from wide_deep_amd.engine import WideDeepEngine
    spec = _spec(hidden=(32, 16))
    a = WideDeepEngine(spec, max_batch=128, seed=5)
    hb = synth.make_raw_batch(a.plan, 96, seed=1, pos_rate=0.3)
    a.train_step(synth.to_device_ids(a.plan, hb))
    torch.cuda.synchronize()
    st = a.export_state()
    prefix = T.write_tf_checkpoint(str(tmp_path / ("model.ckpt-%d" % a.global_step)), {k: v.cpu().numpy() for k, v in st.items()})
    back = {k: torch.from_numpy(v) for k, v in T.read_tf_checkpoint(prefix).items()}
    assert sorted(back) == sorted(st) and all(torch.equal(back[k], st[k].cpu()) for k in st)
    b = WideDeepEngine(spec, max_batch=128, seed=9)
    b.import_state(back)
    assert b.global_step == a.global_step
    hb2 = synth.make_raw_batch(a.plan, 96, seed=2, pos_rate=0.3)
    a.train_step(synth.to_device_ids(a.plan, hb2)); b.train_step(synth.to_device_ids(b.plan, hb2))
    torch.cuda.synchronize()
    sa, sb = a.export_state(), b.export_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_mixed_embedding_dims_and_odd_sizes():
    s = _spec(n_sparse=6, hidden=(20, 10), n_dense=1)
    for i, d in enumerate([4, 8, 16, 32, 64, 8]):
        s.slots[i].dim = d
        s.slots[i].num_buckets = 50 + 37 * i
    _run(s, B=77, mean_len=2)


def test_two_towers_share_input_layer():
    from wide_deep_amd.plan import TowerSpec
    s = _spec()
    s.towers = [TowerSpec([32, 16], "simple"), TowerSpec([16], "dense")]
    _run(s)


def test_graph_capture_replays_same_step():
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from tests.helpers import assert_close
    spec = _spec()
    a = WideDeepEngine(spec, max_batch=128, seed=5)
    b = WideDeepEngine(spec, max_batch=128, seed=5)
    b.import_state(a.export_state())
    hb = synth.make_raw_batch(a.plan, 64, seed=1, pos_rate=0.3)
    bta, btb = synth.to_device_ids(a.plan, hb), synth.to_device_ids(b.plan, hb)
    st0 = a.export_state()
    replay = b.capture_train_step(btb, warmup=1)   # warmup + capture mutate b: reset, then replay
    b.import_state(st0)
    for _ in range(3):
        a.train_step(bta)
        replay()
    torch.cuda.synchronize()
    sa, sb = a.export_state(), b.export_state()
    for k in sa:
        if k != "global_step":
            assert_close(sb[k], sa[k], 1e-6, 1e-7, k)


def test_full_size_c2_properties():
    """BASELINE config 2 at full size: properties that do not need the oracle at this scale."""
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec()
    eng = WideDeepEngine(spec, max_batch=8192, seed=0)
    hb = synth.make_raw_batch(eng.plan, 8192, seed=20260925)
    tb = synth.TokenBatch(eng.plan, hb)
    bt = synth.hash_tokens(eng, tb)
    torch.cuda.synchronize()
    ids = bt.ids.cpu().numpy()
    assert ids.min() >= 0 and ids.max() < 1_000_000
    # (1) hashing is a function of the token only: equal raw values in a slot give equal ids
    raw = hb["raw"].reshape(8192, 26)
    col = raw[:, 0]
    first = {}
    for r, i in zip(col.tolist(), ids.reshape(8192, 26)[:, 0].tolist()):
        assert first.setdefault(r, i) == i
    # (2) untouched rows stay bit-identical, touched rows change; accumulators never decrease
    emb0, acc0, wide0 = eng.emb.clone(), eng.emb_acc.clone(), eng.wide.clone()
    loss0 = float(eng.train_step(bt))
    torch.cuda.synchronize()
    touched = torch.zeros(eng.plan.total_rows, dtype=torch.bool, device="cuda")
    rows = torch.as_tensor(ids.reshape(8192, 26).astype(np.int64) + np.asarray(eng.plan.row_base)[None, :]).cuda().reshape(-1)
    touched[rows] = True
    e0, e1 = emb0.view(-1, 16), eng.emb.view(-1, 16)
    assert torch.equal(e0[~touched], e1[~touched])
    assert bool(((e0[touched] != e1[touched]).any(dim=1)).float().mean() > 0.99)
    assert bool((eng.emb_acc >= acc0).all())
    assert torch.equal(wide0[~touched], eng.wide[~touched])
    assert bool((eng.wide[touched][:, 2] > 0.1).all())
    # (3) loss of the same batch goes down after updates; everything finite
    for _ in range(5):
        loss = float(eng.train_step(bt))
    assert np.isfinite(loss) and loss < loss0
    assert bool(torch.isfinite(eng.emb).all()) and bool(torch.isfinite(eng.P).all())
