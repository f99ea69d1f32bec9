"""CPU: the Estimator-shaped object (wide_deep_amd/estimator.py: train / evaluate / predict, checkpoints, step accounting)
around a STAND-IN engine -- a logistic model over hashed wide ids computed in numpy -- so that the host logic the reference's
train.py / eval.py / pred.py rely on (python/train.py:65-165) is exercised without a GPU.  The real engine takes its place in
tests/test_gpu_c1.py."""
import os
import types

import numpy as np
import pytest
import torch

from tests.test_featurizer_host_cpu import _fake_call

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "c1_rows.tsv")


class StandInEngine(object):
    """wide-only logistic regression with plain SGD on the ids the featurizer emits; the attributes and methods are the ones
    WideAndDeepClassifier touches."""

    def __init__(self, spec, max_batch, max_nnz=None, seed=0):
        from wide_deep_amd.plan import FeaturePlan
        self.spec, self.plan = spec, FeaturePlan(spec)
        self.device = torch.device("cpu")
        self.max_batch, self.max_nnz = max_batch, max_nnz or max_batch * len(spec.slots) * 16
        self.global_step = 0
        self.w = torch.zeros(int(self.plan.total_rows))
        self.logit, self.prob = torch.zeros(max_batch), torch.zeros(max_batch)
        self.steps_seen = []

    def _rows(self, bt):
        S = self.plan.S
        offs = bt.bag_offs.numpy().astype(np.int64)
        slot_of = np.repeat(np.tile(np.arange(S), bt.B), np.diff(offs))
        ex_of = np.repeat(np.repeat(np.arange(bt.B), S), np.diff(offs))
        rows = np.asarray(self.plan.row_base, np.int64)[slot_of] + bt.ids.numpy()[: offs[-1]]
        return torch.as_tensor(rows), torch.as_tensor(ex_of)

    def forward(self, bt, need_loss=True):
        rows, ex = self._rows(bt)
        x = torch.zeros(bt.B).index_add_(0, ex, self.w[rows])
        self.logit[: bt.B], self.prob[: bt.B] = x, torch.sigmoid(x)
        return self.logit[: bt.B]

    def train_step(self, bt):
        x = self.forward(bt)
        y = bt.labels
        loss = (torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))).sum()
        rows, ex = self._rows(bt)
        self.w.index_add_(0, rows, -0.05 * (torch.sigmoid(x) - y)[ex])
        self.global_step += 3 if self.spec.model_type == "wide_deep" else 1        # quirk C.4
        self.steps_seen.append(bt.B)
        return loss

    def export_state(self):
        return {"w": self.w.clone(), "global_step": torch.tensor(self.global_step)}

    def state_shapes(self):
        return {"w": tuple(self.w.shape), "global_step": ()}

    def import_state(self, st):
        self.w.copy_(st["w"])
        self.global_step = int(st["global_step"])


@pytest.fixture
def make_model(monkeypatch, tmp_path):
    from wide_deep_amd import build_estimator as BE, estimator as E, features as F
    from wide_deep_amd.read_conf import Config
    monkeypatch.setattr(F, "call", _fake_call)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: types.SimpleNamespace(cuda_stream=0))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)

    def make(model_dir, runconfig=None, max_batch=128):
        spec = BE.build_model_spec(Config(), "wide_deep")
        rc = dict(Config().runconfig)
        rc.update(runconfig or {})
        return E.WideAndDeepClassifier(spec, model_dir=model_dir, runconfig=rc, engine=StandInEngine(spec, max_batch))
    return make


def _files(tmp_path):
    lines = open(FIXTURE, "rb").read().splitlines()
    p = tmp_path / "rows.tsv"
    p.write_bytes(b"\n".join(lines) + b"\n")
    q = tmp_path / "pred.tsv"
    q.write_bytes(b"\n".join(ln.split(b"\t", 1)[1] for ln in lines[:20]) + b"\n")
    return str(p), str(q), len(lines)


def test_train_steps_checkpoints_resume_and_rotation(tmp_path, make_model):
    from wide_deep_amd import dataset as DS
    path, _, n = _files(tmp_path)
    model_dir = str(tmp_path / "model")
    m = make_model(model_dir, {"keep_checkpoint_max": 2})
    m.train(input_fn=lambda: DS.input_fn(path, None, "train", 128), steps=2)
    assert m.engine.global_step == 6 and m.last_train["steps"] == 2 and m.last_train["examples"] == 256
    assert os.path.exists(os.path.join(model_dir, "model.ckpt-6.pt"))
    m.train(input_fn=lambda: DS.input_fn(path, None, "train", 128))                  # a full pass: ceil(560 / 128) steps
    assert m.last_train["steps"] == 5 and m.last_train["examples"] == n and m.engine.steps_seen[-1] == n - 4 * 128
    assert m.engine.global_step == 6 + 15
    m.train(input_fn=lambda: DS.input_fn(path, None, "train", 128), max_steps=27)    # global-step budget, not a step count
    assert m.engine.global_step == 27
    m.train(input_fn=lambda: DS.input_fn(path, None, "train", 128), max_steps=27)    # already there: nothing runs
    assert m.engine.global_step == 27
    kept = sorted(f for f in os.listdir(model_dir) if f.startswith("model.ckpt-"))
    assert kept == ["model.ckpt-21.pt", "model.ckpt-27.pt"]                          # keep_checkpoint_max = 2
    # a fresh object on the same model_dir resumes from the newest checkpoint
    m2 = make_model(model_dir)
    m2.train(input_fn=lambda: DS.input_fn(path, None, "train", 128), steps=1)
    assert m2.engine.global_step == 30 and m2.latest_checkpoint().endswith("model.ckpt-30.pt")
    # no model_dir: trains, writes nothing
    m3 = make_model(None)
    m3.train(input_fn=lambda: DS.input_fn(path, None, "train", 128), steps=1)
    assert m3.engine.global_step == 3 and m3.latest_checkpoint() is None


def test_resume_from_a_tensorflow_checkpoint_in_model_dir(tmp_path, make_model):
    """python/train.py:188-191 (`keep_train`): the reference resumes from what tf.estimator left in model_dir.  A model_dir that
    holds TensorFlow's checkpoint container and none of ours is restored through wide_deep_amd/tf_checkpoint.py; our own
    checkpoints, once written, take over; export_tf_checkpoint writes the container back."""
    from wide_deep_amd import dataset as DS, tf_checkpoint as T
    path, _, n = _files(tmp_path)
    model_dir = str(tmp_path / "tfmodel")
    os.makedirs(model_dir)
    m0 = make_model(None)
    w = torch.arange(int(m0.engine.plan.total_rows), dtype=torch.float32) * 1e-6
    T.write_tf_checkpoint(os.path.join(model_dir, "model.ckpt-300"), {"w": w.numpy(), "global_step": np.asarray(300, np.int64)})
    T.write_tf_checkpoint(os.path.join(model_dir, "model.ckpt-90"), {"w": (w * 0).numpy(), "global_step": np.asarray(90, np.int64)})
    open(os.path.join(model_dir, "checkpoint"), "w").write('model_checkpoint_path: "model.ckpt-300"\n')
    m = make_model(model_dir)
    assert m.latest_checkpoint() == os.path.join(model_dir, "model.ckpt-300.index")
    m.train(input_fn=lambda: DS.input_fn(path, None, "train", 128), steps=1)
    assert m.engine.global_step == 303                                   # restored 300, one wide_deep step = +3
    assert os.path.exists(os.path.join(model_dir, "model.ckpt-303.pt")) and m.latest_checkpoint().endswith("model.ckpt-303.pt")
    touched = (m.engine.w != w).sum()
    assert 0 < int(touched) < w.numel() // 2 and torch.equal(m.engine.w[m.engine.w == w], w[m.engine.w == w])
    prefix = m.export_tf_checkpoint()
    back = T.read_tf_checkpoint(prefix)
    assert int(back["global_step"]) == 303 and np.array_equal(back["w"], m.engine.w.numpy())


def test_evaluate_and_predict_contract(tmp_path, make_model):
    from wide_deep_amd import dataset as DS
    path, pred_path, n = _files(tmp_path)
    m = make_model(str(tmp_path / "model"))
    m.train(input_fn=lambda: DS.input_fn(path, None, "train", 128), steps=1)
    ev0 = m.evaluate(input_fn=lambda: DS.input_fn(path, None, "eval", 128))
    assert set(ev0) == {"accuracy", "accuracy_baseline", "auc", "auc_precision_recall", "average_loss", "label/mean", "loss",
                        "precision", "prediction/mean", "recall", "global_step"}
    assert ev0["global_step"] == 3 and abs(ev0["label/mean"] - 6.0 / n) < 1e-9 and 0.0 <= ev0["auc"] <= 1.0
    assert abs(ev0["loss"] - ev0["average_loss"] * n / 5) < 1e-6 * ev0["loss"]        # mean over 5 batches of the batch SUM
    ev_part = m.evaluate(input_fn=lambda: DS.input_fn(path, None, "eval", 128), steps=2)
    assert ev_part["average_loss"] != ev0["average_loss"]                             # only the first 256 rows
    m.train(input_fn=lambda: DS.input_fn(path, None, "train", 128))
    ev1 = m.evaluate(input_fn=lambda: DS.input_fn(path, None, "eval", 128))
    assert ev1["average_loss"] < ev0["average_loss"]                                  # the reference's own test criterion
    preds = list(m.predict(input_fn=lambda: DS.input_fn(pred_path, None, "pred", 8)))
    assert len(preds) == 20 and set(preds[0]) == {"logits", "logistic", "probabilities", "class_ids", "classes"}
    for d in preds:
        assert abs(d["probabilities"].sum() - 1.0) < 1e-6 and d["class_ids"][0] == int(d["logistic"][0] > 0.5)
        assert d["classes"][0] == str(d["class_ids"][0]).encode()
    only = next(iter(m.predict(input_fn=lambda: DS.input_fn(pred_path, None, "pred", 8), predict_keys=["logistic"])))
    assert list(only) == ["logistic"]
    with pytest.raises(ValueError, match="evaluate needs labels"):
        m.evaluate(input_fn=lambda: DS.input_fn(pred_path, None, "pred", 8))
