"""GPU parity on BASELINE configs[0]: the repo-default conf (conf/*.yaml = the reference's shipped defaults) on real
rows of the reference's bundled click log (tests/golden/c1_rows.tsv), through the drop-in host layer
(dataset.input_fn -> Featurizer -> WideDeepEngine / WideAndDeepClassifier) against the CPU oracle
(oracle/columns.py -> oracle.OracleWideDeep).  Integer work bit-exact, logits within fp32 tolerance."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "c1_rows.tsv")
RTOL, ATOL = 2e-4, 2e-5


def _lines_with_na():
    from tests.test_conf_dataset import _with_na_rows
    from wide_deep_amd.read_conf import Config
    lines = open(FIXTURE, "rb").read().splitlines()
    return _with_na_rows(lines, Config().read_schema())


def _write(tmp_path, lines, name="rows.tsv"):
    p = tmp_path / name
    p.write_bytes(b"\n".join(lines) + b"\n")
    return str(p)


@pytest.mark.parametrize("mode", ["device", "host"])
@pytest.mark.parametrize("padding", ["tf_dense", "ragged"])
def test_every_column_id_bit_exact_on_real_rows(tmp_path, padding, mode):
    """mode: the featurizer that builds the whole bag CSR on the GPU from a device-side slot table (default), and round 1's
    per-column host arithmetic + one emit launch per column; both against oracle/columns.py, and against each other."""
    from oracle import columns as OC
    from tests.helpers import slot_csr
    from wide_deep_amd import build_estimator as BE, dataset as DS
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.features import Featurizer
    from wide_deep_amd.read_conf import Config, conf_dir
    lines = _lines_with_na()
    path = _write(tmp_path, lines)
    spec = BE.build_model_spec(Config(), "wide")          # wide-only: every categorical column, no big embedding tables
    eng = WideDeepEngine(spec, max_batch=512, max_nnz=512 * 70 * 16)
    fz = Featurizer(eng, cross_padding=padding, mode=mode)
    other = Featurizer(eng, cross_padding=padding, mode="host" if mode == "device" else "device")
    assert fz.mode == mode
    oc = OC.Columns(conf_dir())
    k = 0
    for raw in DS.input_fn(path, None, "eval", 512):
        bt = fz.to_device(raw)
        b2 = other.to_device(raw)
        torch.cuda.synchronize()
        assert bt.nnz == b2.nnz and bt.one_hot == b2.one_hot and torch.equal(bt.bag_offs, b2.bag_offs)
        assert torch.equal(bt.ids[:bt.nnz], b2.ids[:b2.nnz]) and torch.equal(bt.labels, b2.labels)
        assert (bt.dense is None and b2.dense is None) or torch.equal(bt.dense, b2.dense)
        got = slot_csr(eng.plan, bt.ids.cpu().numpy(), bt.bag_offs.cpu().numpy(), raw.B)
        exp = oc.transform(oc.parse(lines[k:k + raw.B]), cross_padding=padding)["ids"]
        k += raw.B
        assert set(got) == set(exp)
        for name in exp:
            eids, eoffs = exp[name]
            gids, goffs = got[name]
            assert np.array_equal(goffs, eoffs), name
            assert np.array_equal(gids, np.asarray(eids, dtype=np.int64)), name
    assert k == len(lines)


@pytest.mark.parametrize("padding", ["tf_dense", "ragged"])
def test_device_featurizer_on_mutated_rows(tmp_path, padding):
    """Real rows with randomly damaged fields -- empty / '-' / multi-valued / out-of-vocabulary tokens, ids at and beyond both
    ends of the identity ranges, -1, missing and extreme floats -- in batches of 1, 7 and 64 (a batch where a multi-valued
    feature has NO token at all included): the device featurizer == the host featurizer == oracle/columns.py, id for id."""
    import random
    from oracle import columns as OC
    from tests.helpers import slot_csr
    from wide_deep_amd import build_estimator as BE, dataset as DS
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.features import Featurizer
    from wide_deep_amd.read_conf import Config, conf_dir
    conf = Config()
    schema, fconf = conf.read_schema(), conf.read_feature_conf()
    pos = {v: k - 1 for k, v in schema.items()}
    rng = random.Random(20260926)
    base = open(FIXTURE, "rb").read().splitlines()
    words = [b"a", b"w", b"0", b"1", b"2", b"5", b"zz", b"iphone", b"-", b"", b"homepage", b"T1348648756099", b"x" * 40]

    def damage(line):
        parts = line.split(b"\t")
        for f, c in fconf.items():
            if f not in pos or rng.random() > 0.35:
                continue
            if c["type"] == "category" and c["transform"] == "identity":
                n = int(c["parameter"])
                parts[pos[f]] = str(rng.choice([-1, 0, 1, n - 1, n, n + 5, -7])).encode() if rng.random() < 0.9 else b"-"
            elif c["type"] == "category":
                k = rng.choice([0, 1, 1, 2, 3, 6])
                parts[pos[f]] = b",".join(rng.choice(words) for _ in range(k))
            else:
                parts[pos[f]] = rng.choice([b"-", b"", b"0", b"1e-3", b"123456.5", b"0.5", b"7"])
        return b"\t".join(parts)

    spec = BE.build_model_spec(conf, "wide")
    eng = WideDeepEngine(spec, max_batch=64, max_nnz=64 * 70 * 64)
    dev, host = Featurizer(eng, cross_padding=padding, mode="device"), Featurizer(eng, cross_padding=padding, mode="host")
    oc = OC.Columns(conf_dir())
    for B in (1, 7, 64, 64):
        lines = [damage(rng.choice(base)) for _ in range(B)]
        if B == 7:       # a multi-valued feature without a single token in the whole batch (Lmax = 0: its crosses are empty)
            for i, ln in enumerate(lines):
                parts = ln.split(b"\t")
                parts[pos["ad_cates"]] = b""
                lines[i] = b"\t".join(parts)
        path = tmp_path / ("m%d.tsv" % B)
        path.write_bytes(b"\n".join(lines) + b"\n")
        raw = next(iter(DS.input_fn(str(path), None, "eval", B, prefetch=0)))
        assert raw.B == B
        a, b = dev.to_device(raw), host.to_device(raw)
        torch.cuda.synchronize()
        assert a.nnz == b.nnz and a.one_hot == b.one_hot and torch.equal(a.bag_offs, b.bag_offs)
        assert torch.equal(a.ids[:a.nnz], b.ids[:b.nnz])
        got = slot_csr(eng.plan, a.ids.cpu().numpy(), a.bag_offs.cpu().numpy(), B)
        exp = oc.transform(oc.parse(lines), cross_padding=padding)["ids"]
        assert set(got) == set(exp)
        for name in exp:
            assert np.array_equal(got[name][1], exp[name][1]), name
            assert np.array_equal(got[name][0], np.asarray(exp[name][0], dtype=np.int64)), name


def test_default_conf_train_steps_match_oracle(tmp_path):
    """Full default model (70 wide columns, 12.7M rows, 47 embedding columns, tower [1024,512,256]) for 3 steps."""
    from oracle import columns as OC
    from tests.helpers import assert_close, oracle_from_engine
    from wide_deep_amd import build_estimator as BE, dataset as DS
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.features import Featurizer
    from wide_deep_amd.read_conf import Config, conf_dir
    lines = open(FIXTURE, "rb").read().splitlines()
    path = _write(tmp_path, lines)
    spec = BE.build_model_spec(Config(), "wide_deep")
    eng = WideDeepEngine(spec, max_batch=256, max_nnz=256 * 70 * 16, seed=123)
    fz = Featurizer(eng)
    oc = OC.Columns(conf_dir())
    ora = oracle_from_engine(eng)
    assert sorted(c["name"] for c in ora.deep_cols) == sorted(c["name"] for c in oc.deep_cols)
    k = 0
    for step, raw in enumerate(DS.input_fn(path, None, "eval", 256)):
        if step == 3:
            break
        bt = fz.to_device(raw)
        loss = float(eng.train_step(bt))
        torch.cuda.synchronize()
        ob = oc.transform(oc.parse(lines[k:k + raw.B]))
        k += raw.B
        oloss, ologits = ora.train_step(ob)
        assert_close(eng.logit[: raw.B], ologits, RTOL, ATOL, "logits step %d" % step)
        assert abs(loss - oloss) <= 2e-4 * max(1.0, abs(oloss)), (step, loss, oloss)
    assert eng.global_step == 9          # +3 per batch in wide_deep mode (quirk C.4)


def test_estimator_train_evaluate_predict_checkpoint(tmp_path):
    from wide_deep_amd import build_estimator as BE, dataset as DS
    lines = open(FIXTURE, "rb").read().splitlines()
    path = _write(tmp_path, lines)
    pred_path = _write(tmp_path, [ln.split(b"\t", 1)[1] for ln in lines[:20]], "pred.tsv")
    model_dir = str(tmp_path / "model" / "wide_deep")
    m = BE.build_custom_estimator(model_dir, "wide_deep", max_batch=128)
    ev0 = None
    m.train(input_fn=lambda: DS.input_fn(path, None, "train", 128), steps=2)
    assert m.engine.global_step == 6 and os.path.exists(os.path.join(model_dir, "model.ckpt-6.pt"))
    ev0 = m.evaluate(input_fn=lambda: DS.input_fn(path, None, "eval", 128))
    for key in ("accuracy", "accuracy_baseline", "auc", "auc_precision_recall", "average_loss", "label/mean", "loss",
                "precision", "prediction/mean", "recall", "global_step"):
        assert key in ev0, key
    assert ev0["global_step"] == 6 and 0.0 <= ev0["auc"] <= 1.0 and abs(ev0["label/mean"] - 6.0 / 560) < 1e-9
    m.train(input_fn=lambda: DS.input_fn(path, None, "train", 128))       # one full pass
    ev1 = m.evaluate(input_fn=lambda: DS.input_fn(path, None, "eval", 128))
    assert ev1["average_loss"] < ev0["average_loss"]                       # the reference's own test criterion
    # a fresh estimator on the same model_dir resumes from the checkpoint and predicts identically
    m2 = BE.build_custom_estimator(model_dir, "wide_deep", max_batch=128)
    p1 = [d["logistic"][0] for d in m.predict(input_fn=lambda: DS.input_fn(pred_path, None, "pred", 128))]
    p2 = [d["logistic"][0] for d in m2.predict(input_fn=lambda: DS.input_fn(pred_path, None, "pred", 128))]
    assert len(p1) == 20 and np.allclose(p1, p2, rtol=0, atol=0)
    assert m2.engine.global_step == m.engine.global_step
    keys = next(iter(m.predict(input_fn=lambda: DS.input_fn(pred_path, None, "pred", 128))))
    assert set(keys) == {"logits", "logistic", "probabilities", "class_ids", "classes"}


def test_train_py_cli_dynamic_mode(tmp_path):
    """`python train.py` end to end in the default dynamic_train mode on two small files."""
    import subprocess
    import sys
    lines = open(FIXTURE, "rb").read().splitlines()
    d = tmp_path / "train"
    d.mkdir()
    (d / "part1").write_bytes(b"\n".join(lines[:300]) + b"\n")
    (d / "part2").write_bytes(b"\n".join(lines[300:]) + b"\n")
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "train.py"), "--model_dir", str(tmp_path / "model"),
                          "--train_data", str(d), "--train_epochs", "1", "--batch_size", "128", "--model_type", "wide_deep"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Using dynamic train mode." in out.stdout and "auc:" in out.stdout and "examples/sec" in out.stdout
    assert any(f.startswith("model.ckpt-") for f in os.listdir(tmp_path / "model" / "wide_deep"))


def test_eval_py_and_pred_py_clis(tmp_path):
    """`python eval.py` / `python pred.py` (reference python/eval.py, python/pred.py) on a model trained by train.py:
    the printed metrics equal estimator.evaluate, every prediction line carries the winning class and its probability."""
    import subprocess
    import sys
    from wide_deep_amd import build_estimator as BE, dataset as DS
    lines = open(FIXTURE, "rb").read().splitlines()
    path = _write(tmp_path, lines)
    pred_path = _write(tmp_path, [ln.split(b"\t", 1)[1] for ln in lines[:20]], "pred.tsv")
    model_root = str(tmp_path / "model")
    m = BE.build_custom_estimator(os.path.join(model_root, "wide_deep"), "wide_deep", max_batch=128)
    m.train(input_fn=lambda: DS.input_fn(path, None, "train", 128), steps=3)
    ref = m.evaluate(input_fn=lambda: DS.input_fn(path, None, "eval", 128))
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "eval.py"), "--model_dir", model_root, "--test_data", path,
                          "--batch_size", "128", "--model_type", "wide_deep"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = dict(ln.split(": ", 1) for ln in out.stdout.splitlines() if ": " in ln and not ln.startswith(("INFO", "Model")))
    for key in ("auc", "average_loss", "accuracy", "label/mean", "global_step"):
        assert abs(float(got[key]) - float(ref[key])) <= 1e-6 * max(1.0, abs(float(ref[key]))), (key, got[key], ref[key])
    out = subprocess.run([sys.executable, os.path.join(root, "pred.py"), "--model_dir", model_root, "--data_dir", pred_path,
                          "--batch_size", "128", "--model_type", "wide_deep"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    preds = [ln for ln in out.stdout.splitlines() if ln.startswith("Prediction is")]
    assert len(preds) == 20 and all(p.startswith('Prediction is "0"') or p.startswith('Prediction is "1"') for p in preds)
    out = subprocess.run([sys.executable, os.path.join(root, "pred.py"), "--model_dir", model_root], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode != 0 and "Must specify prediction data_file by --data_dir" in out.stderr


@pytest.mark.parametrize("bs", [64, 128])
def test_train_loop_captured_steps_equal_eager_steps(tmp_path, monkeypatch, bs):
    """python/train.py:65-165 at the reference's batch sizes (64 shipped, conf/train.yaml:47): from the third batch of a size on the
    loop replays ONE hipGraph per step -- the featurizer's launches (token count and id count on the device, fixed-capacity stage)
    + the train step -- and must train exactly like the eager launches (WD_TRAIN_GRAPH=0) on the same rows: same logits on a probe
    batch, same parameters, the shorter last batch of the file on the eager path in both."""
    from wide_deep_amd import build_estimator as BE, dataset as DS
    lines = open(FIXTURE, "rb").read().splitlines()
    path = _write(tmp_path, lines * 2 + lines[: bs // 2 + 3])        # several full batches + a ragged last one
    states = {}
    for tag, env in (("graph", "1"), ("eager", "0")):
        monkeypatch.setenv("WD_TRAIN_GRAPH", env)
        m = BE.build_custom_estimator(str(tmp_path / tag), "wide_deep", max_batch=bs)
        m.train(input_fn=lambda: DS.input_fn(path, None, "train", bs))
        torch.cuda.synchronize()
        nfull = (2 * len(lines) + bs // 2 + 3) // bs
        assert m.last_train["steps"] == nfull + 1 and m.last_train["examples"] == 2 * len(lines) + bs // 2 + 3
        assert m.last_train["graph_batch_sizes"] == ([bs] if tag == "graph" else [])
        states[tag] = (m.engine.export_state(), m.engine.global_step, float(m.last_train["loss"]))
    (a, ga, la), (b, gb, lb) = states["graph"], states["eager"]
    assert ga == gb and abs(la - lb) <= 1e-5 * max(1.0, abs(lb))
    for k in b:
        if b[k].dtype.is_floating_point:
            assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-6), "%s differs (max |d| %.3g)" % (k, float((a[k] - b[k]).abs().max()))
