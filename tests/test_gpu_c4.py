"""GPU parity on the BASELINE configs[3] SHAPE (C4) at test size, through the whole drop-in path with a generated conf
directory: multi-hot hash slots (avg ~3 ids), 200-bucket crossed columns over 2 and 3 slots (one wide-only), ResDnn
(`resnet`) tower, weight column on (pos 0.99 / neg 0.01), continuous features with every normalizer.
dataset.input_fn -> Featurizer -> WideDeepEngine vs oracle/columns.py -> OracleWideDeep, both cross paddings."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def _make_conf(d, mode="resnet", hidden=(32, 16)):
    feats = ["c%d" % i for i in range(6)]
    nums = ["x0", "x1", "x2"]
    schema = {1: "clk"}
    for i, f in enumerate(feats + nums + ["idn", "unused"]):
        schema[i + 2] = f
    feature = {}
    for i, f in enumerate(feats):
        feature[f] = {"type": "category", "transform": "hash_bucket", "parameter": [1000, 1000, 500, 20000, 100, 1000][i]}
    feature["x0"] = {"type": "continuous", "transform": "min_max", "parameter": {"normalization": [0, 10], "boundaries": [2, 4, 6]}}
    feature["x1"] = {"type": "continuous", "transform": "standard", "parameter": {"normalization": [1.5, 2.0], "boundaries": None}}
    feature["x2"] = {"type": "continuous", "transform": None, "parameter": {"normalization": None, "boundaries": None}}
    feature["idn"] = {"type": "category", "transform": "identity", "parameter": 7}
    cross = {"c0&c1": {"hash_bucket_size": 0.2, "is_deep": 1},
             "c2&c3&c4": {"hash_bucket_size": 0.2, "is_deep": 1},
             "x0&idn&c5": {"hash_bucket_size": 0.2, "is_deep": 0}}
    model = {"linear_optimizer": "tf.train.FtrlOptimizer(learning_rate=0.1,l1_regularization_strength=0.5,l2_regularization_strength=1)",
             "linear_initial_learning_rate": 0.05, "linear_decay_rate": 0.8, "dnn_hidden_units": list(hidden),
             "dnn_connected_mode": mode, "dnn_optimizer": "Adagrad", "dnn_initial_learning_rate": 0.05, "dnn_decay_rate": 0.8,
             "dnn_activation_function": "relu", "dnn_l1": 0.1, "dnn_l2": 0.1, "dnn_dropout": None,
             "dnn_batch_normalization": 1, "cnn_use_flag": 0, "cnn_optimizer": "Adagrad"}
    train = {"train": {"model_dir": "model", "model_type": "wide_deep", "train_data": "x", "eval_data": "x", "test_data": "x",
                       "image_train_data": None, "image_eval_data": None, "image_test_data": None, "dynamic_train": False,
                       "train_epochs": 1, "epochs_per_eval": 1, "batch_size": 64, "keep_train": 0, "checkpoint_path": None,
                       "pos_sample_loss_weight": 0.99, "neg_sample_loss_weight": 0.01, "multivalue": 1,
                       "num_examples": 100, "num_parallel_calls": None},
             "distribution": {"is_distribution": 0, "cluster": {"ps": [], "chief": [], "worker": []}, "job_name": "ps", "task_index": 0},
             "runconfig": {"tf_random_seed": 1, "save_checkpoints_secs": None, "keep_checkpoint_max": 2, "log_step_count_steps": 0}}
    os.makedirs(d, exist_ok=True)
    for name, obj in (("schema", schema), ("feature", feature), ("cross_feature", cross), ("model", model), ("train", train)):
        with open(os.path.join(d, name + ".yaml"), "w") as f:
            yaml.safe_dump(obj, f, sort_keys=False)
    return feats


def _make_rows(n, seed):
    rng = np.random.default_rng(seed)
    lines = []
    for _ in range(n):
        parts = [b"1" if rng.random() < 0.3 else b"0"]
        for i in range(6):
            k = int(np.clip(1 + rng.poisson(2.0), 1, 8)) if rng.random() > 0.05 else 0
            toks = [("t%d_%d" % (i, int(rng.integers(0, 40)))).encode() for _ in range(k)]
            parts.append(b",".join(toks) if toks else b"-")
        parts.append(("%.3f" % (rng.random() * 10)).encode())
        parts.append(("%.3f" % rng.normal()).encode())
        parts.append(("%.3f" % rng.normal()).encode() if rng.random() > 0.1 else b"-")
        parts.append(str(int(rng.integers(-1, 9))).encode())
        parts.append(b"junk")
        lines.append(b"\t".join(parts))
    return lines


@pytest.mark.parametrize("padding,mode", [("ragged", "resnet"), ("tf_dense", "resnet"), ("ragged", "dense")])
def test_c4_shape_through_conf_path_matches_oracle(tmp_path, monkeypatch, padding, mode):
    from oracle import columns as OC
    from tests.helpers import assert_close, oracle_from_engine, slot_csr
    from wide_deep_amd import build_estimator as BE, dataset as DS
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.features import Featurizer
    from wide_deep_amd.read_conf import Config
    cdir = str(tmp_path / "conf")
    _make_conf(cdir, mode=mode)
    monkeypatch.setenv("WD_CONF_DIR", cdir)
    lines = _make_rows(3 * 64, seed=5)
    path = tmp_path / "rows.tsv"
    path.write_bytes(b"\n".join(lines) + b"\n")
    conf = Config()
    spec = BE.build_model_spec(conf, "wide_deep")
    assert spec.use_weight_column and spec.towers[0].mode == mode
    assert sorted(s.num_buckets for s in spec.slots if s.kind == "cross") == [200, 200, 200]
    eng = WideDeepEngine(spec, max_batch=64, max_nnz=64 * 12 * 64, seed=3)
    fz = Featurizer(eng, cross_padding=padding)
    oc = OC.Columns(cdir)
    ora = oracle_from_engine(eng)
    ora.dnn_opt, ora.lin_opt = oc.optimizers()
    k = 0
    for step, raw in enumerate(DS.input_fn(str(path), None, "eval", 64, conf=conf)):
        assert raw.weights is not None
        bt = fz.to_device(raw)
        ob = oc.transform(oc.parse(lines[k:k + raw.B]), cross_padding=padding)
        k += raw.B
        torch.cuda.synchronize()
        got = slot_csr(eng.plan, bt.ids.cpu().numpy(), bt.bag_offs.cpu().numpy(), raw.B)
        for name, (eids, eoffs) in ob["ids"].items():
            assert np.array_equal(got[name][1], eoffs), name
            assert np.array_equal(got[name][0], np.asarray(eids, dtype=np.int64)), name     # bit-exact ids
        loss = float(eng.train_step(bt))
        torch.cuda.synchronize()
        oloss, ologits = ora.train_step(ob)
        assert_close(eng.logit[: raw.B], ologits, 2e-4, 2e-5, "logits step %d" % step)
        assert abs(loss - oloss) <= 2e-4 * max(1.0, abs(oloss)), (step, loss, oloss)
    assert k == len(lines)


@pytest.mark.parametrize("dims", [(16, 4), (8, 8), (16, 10), (4, 6), (8, 1)])
def test_small_table_path_trains_like_the_general_path_and_the_oracle(monkeypatch, dims):
    """Crossed columns (200 / 37 buckets over 2 and 3 multi-hot slots, one of them wide-only) through csrc/small_tables.hip
    (tables in LDS, bags counted, no sort) against the SAME model on the general bucketed path (WD_SMALL_TABLES=0) and against
    the oracle: three steps, empty bags, a bag longer than a workgroup, batch not a multiple of the slice / workgroup sizes."""
    from tests.helpers import assert_close, oracle_batch, oracle_from_engine, parsed_batch_ids
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.features import Featurizer
    from wide_deep_amd.plan import CatSlot, CrossKey, FeaturePlan, criteo_spec
    B = 203
    spec = criteo_spec(n_dense=3, n_sparse=5, buckets=5000, dim=dims[0], hidden=(32, 16), mode="resnet", crosses=((0, 1), (2, 3, 4)),
                       cross_buckets=200, use_weight_column=True)
    for s in spec.slots:
        if s.kind == "cross":
            s.dim = dims[1]
    spec.slots.append(CatSlot(name="C01_X_C04", kind="cross", num_buckets=37, deep=None, dim=0, wide=True,
                              cross_keys=[CrossKey("C04", "string"), CrossKey("C01", "string")]))
    gp = FeaturePlan(spec)
    parsed = [synth.make_parsed_batch(gp, B, seed=50 + i, mean_len=4, weights=(0.99, 0.01), pos_rate=0.3) for i in range(3)]
    nnz = max(hb["nnz"] for _, hb in parsed) + 64
    monkeypatch.setenv("WD_SMALL_TABLES", "0")
    gen = WideDeepEngine(spec, max_batch=256, max_nnz=nnz, seed=4)
    monkeypatch.setenv("WD_SMALL_TABLES", "cross")
    eng = WideDeepEngine(spec, max_batch=256, max_nnz=nnz, seed=4)
    # (round 6: the small tables sit in row records of the big columns' width; the general path of `gen` keeps separate tables)
    # (a cross wider than the big columns does not fit their record: separate tables, dims (4, 6))
    assert not gen.small_idx and len(eng.small_idx) == 3 and (eng.rec is not None) == (dims[1] <= dims[0]) and gen.rec is None
    ora = oracle_from_engine(eng)
    fz, fzg = Featurizer(eng, cross_padding="ragged"), Featurizer(gen, cross_padding="ragged")
    for step, (raw, hb) in enumerate(parsed):
        bt, btg = fz.to_device(raw), fzg.to_device(raw)
        assert not bt.one_hot and eng._small_on(bt)
        ids, offs = parsed_batch_ids(eng.plan, hb)
        assert np.array_equal(bt.ids.cpu().numpy()[: bt.nnz].astype(np.int64), ids)
        loss, lossg = float(eng.train_step(bt)), float(gen.train_step(btg))
        torch.cuda.synchronize()
        oloss, ologits = ora.train_step(oracle_batch(eng.plan, ids, offs, B, hb["dense"], hb["labels"], hb["weights"]))
        assert_close(eng.logit[:B], ologits, 2e-4, 2e-5, "logits vs oracle, step %d" % step)
        assert_close(eng.logit[:B], gen.logit[:B], 2e-4, 2e-5, "logits vs the general path, step %d" % step)
        assert abs(loss - oloss) <= 2e-4 * max(1.0, abs(oloss)) and abs(loss - lossg) <= 2e-4 * max(1.0, abs(lossg))
    a, g = eng.export_state(), gen.export_state()
    for k in a:
        if a[k].dtype.is_floating_point:
            assert_close(a[k], g[k], 5e-4, 1e-5, "state %s vs the general path" % k)


def test_small_tables_are_admitted_against_their_joint_maxima(monkeypatch):
    """The small-table kernels size LDS and workspace by (longest table) x (widest embedding + 2) over ALL admitted columns:
    a 3900-row wide-only cross (alone: 7800 floats; its backward asks for 32 + 31 + 5 KB of dynamic LDS -- beyond the 64 KB a
    launch gets without the attribute) and a 1000-row cross of width 4 (alone: 6000) fit one by one and not together
    (3900 x 6 = 23400 > 8192).  The first is admitted, the second stays on the general path, and the step matches the model
    with every column on the general path and the oracle."""
    from tests.helpers import assert_close, oracle_batch, oracle_from_engine, parsed_batch_ids
    from wide_deep_amd import capi, synth
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.features import Featurizer
    from wide_deep_amd.plan import CatSlot, CrossKey, FeaturePlan, criteo_spec
    B = 131
    spec = criteo_spec(n_dense=2, n_sparse=4, buckets=3000, dim=8, hidden=(32, 16), mode="simple", crosses=((0, 1),),
                       cross_buckets=1000)
    for s in spec.slots:
        if s.kind == "cross":
            s.dim = 4
    spec.slots.insert(4, CatSlot(name="C02_X_C03", kind="cross", num_buckets=3900, deep=None, dim=0, wide=True,
                                 cross_keys=[CrossKey("C02", "string"), CrossKey("C03", "string")]))
    gp = FeaturePlan(spec)
    kinds = [(s.kind, s.num_buckets) for s in gp.slots]
    parsed = [synth.make_parsed_batch(gp, B, seed=70 + i, mean_len=3, pos_rate=0.3) for i in range(2)]
    nnz = max(hb["nnz"] for _, hb in parsed) + 64
    monkeypatch.setenv("WD_SMALL_TABLES", "0")
    gen = WideDeepEngine(spec, max_batch=256, max_nnz=nnz, seed=4)
    monkeypatch.setenv("WD_SMALL_TABLES", "cross")
    eng = WideDeepEngine(spec, max_batch=256, max_nnz=nnz, seed=4)
    adm = [kinds[i] for i in eng.small_idx]
    assert len(adm) == 1 and adm[0][0] == "cross", (kinds, eng.small_idx)
    assert eng.small_rows * (eng.small_dim + 2) <= capi.SMALL_MAX_FLOATS
    ora = oracle_from_engine(eng)
    fz, fzg = Featurizer(eng, cross_padding="ragged"), Featurizer(gen, cross_padding="ragged")
    for step, (raw, hb) in enumerate(parsed):
        bt, btg = fz.to_device(raw), fzg.to_device(raw)
        ids, offs = parsed_batch_ids(eng.plan, hb)
        loss, lossg = float(eng.train_step(bt)), float(gen.train_step(btg))
        torch.cuda.synchronize()
        oloss, ologits = ora.train_step(oracle_batch(eng.plan, ids, offs, B, hb["dense"], hb["labels"], hb.get("weights")))
        assert_close(eng.logit[:B], ologits, 2e-4, 2e-5, "logits vs oracle, step %d" % step)
        assert_close(eng.logit[:B], gen.logit[:B], 2e-4, 2e-5, "logits vs the general path, step %d" % step)
    a, g = eng.export_state(), gen.export_state()
    for k in a:
        if a[k].dtype.is_floating_point:
            assert_close(a[k], g[k], 5e-4, 1e-5, "state %s vs the general path" % k)


@pytest.mark.parametrize("buckets", [7, 3000])
def test_flat_ragged_update_equals_the_bucket_update_and_the_oracle(monkeypatch, buckets):
    """Multi-hot batches on row records (round 6): wd_sparse_bucketize -> wd_bucket_sort_ragged -> wd_row_update_ragged against the
    same steps with the sort inside the update's workgroups (WD_FLAT_RAGGED=0: wd_sparse_apply_rec) and against the oracle.
    3000-row vocabularies: short segments -- the two updates add every row's occurrences in the same order: bit-identical tables.
    7 rows: ids repeat INSIDE a bag (equal (row, bag) pairs in the sort) and every row is a long segment (hundreds of occurrences,
    reduced by a whole workgroup in a fixed tree whose shape differs between the two kernels: equal to 1e-5 relative)."""
    from tests.helpers import assert_close, oracle_batch, oracle_from_engine
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.plan import criteo_spec
    B = 301
    spec = criteo_spec(n_dense=3, n_sparse=5, buckets=buckets, dim=8, hidden=(32, 16), mode="resnet", use_weight_column=True)
    monkeypatch.setenv("WD_FLAT_RAGGED", "0")
    ref = WideDeepEngine(spec, max_batch=B, seed=9)
    monkeypatch.setenv("WD_FLAT_RAGGED", "1")
    eng = WideDeepEngine(spec, max_batch=B, seed=9)
    assert eng.rec is not None and ref.rec is not None
    ora = oracle_from_engine(eng)
    for step in range(3):
        hb = synth.make_raw_batch(eng.plan, B, seed=300 + step, mean_len=5, pos_rate=0.3)
        w = np.where(hb["labels"] > 0, 0.99, 0.01).astype(np.float32)
        bt, bt2 = synth.to_device_ids(eng.plan, hb, weights=w), synth.to_device_ids(eng.plan, hb, weights=w)
        assert not bt.one_hot and eng._flat_ragged_ok(bt) and not ref._flat_ragged_ok(bt2)
        if buckets == 7:      # ids repeated inside a bag exist
            ids, offs = bt.ids.cpu().numpy(), bt.bag_offs.cpu().numpy()
            assert any(len(set(ids[offs[g]: offs[g + 1]])) < offs[g + 1] - offs[g] for g in range(len(offs) - 1))
        loss, loss2 = float(eng.train_step(bt)), float(ref.train_step(bt2))
        torch.cuda.synchronize()
        assert eng._bucket_sets[0]["ragged"] and not ref._bucket_sets[0]["ragged"]
        oloss, ologits = ora.train_step(oracle_batch(eng.plan, bt.ids.cpu().numpy(), bt.bag_offs.cpu().numpy(), B, hb["dense"],
                                                     hb["labels"], w))
        assert_close(eng.logit[:B], ologits, 2e-4, 2e-5, "logits vs oracle, step %d" % step)
        if buckets == 3000:
            assert loss == loss2 and torch.equal(eng.rec, ref.rec) and torch.equal(eng.emb_acc, ref.emb_acc), \
                "flat ragged update differs from wd_sparse_apply_rec at step %d" % step
            assert torch.equal(eng.bias, ref.bias) and torch.equal(eng.P, ref.P)
        else:
            assert_close(eng.rec, ref.rec, 1e-5, 1e-6, "records vs wd_sparse_apply_rec, step %d" % step)
            assert_close(eng.emb_acc, ref.emb_acc, 1e-5, 1e-6, "accumulators vs wd_sparse_apply_rec, step %d" % step)
