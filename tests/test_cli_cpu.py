"""CPU: the command lines (wide_deep_amd/cli.py behind train.py / eval.py / pred.py) against goldens recorded by EXECUTING the
reference's python/train.py, eval.py, pred.py with stub modules (tests/golden/make_ref_train_schedule_golden.py):
same flags with the same types and conf/train.yaml defaults, same sequence of Estimator calls for the three schedules."""
import argparse
import io
import json
import os

import pytest

from wide_deep_amd import cli
from wide_deep_amd.read_conf import Config

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_train_schedule.json")))
PATHS = {"--model_dir", "--train_data", "--eval_data", "--test_data"}      # relative to each repository's own layout


@pytest.mark.parametrize("command", ["train", "eval", "pred"])
def test_flags_types_and_defaults_equal_the_reference_parsers(command):
    p = cli.build_parser(command, Config().train)
    mine = {a.option_strings[0]: a for a in p._actions if a.option_strings and a.option_strings[0] != "-h"}
    ref = {f["flag"]: f for f in G["parsers"][command]}
    assert list(mine) == list(ref) or sorted(mine) == sorted(ref)
    for flag, f in ref.items():
        assert mine[flag].type.__name__ == f["type"], flag
        if flag not in PATHS:
            assert mine[flag].default == f["default"], flag


class _Model(object):
    def __init__(self):
        self.calls = []

    def train(self, input_fn, **kw):
        self.calls.append(["train"] + list(input_fn()))

    def evaluate(self, input_fn, **kw):
        self.calls.append(["evaluate"] + list(input_fn()))
        return {"auc": 0.5, "loss": 1.0}


@pytest.mark.parametrize("k", range(len(G["runs"])), ids=lambda k: "%s-e%d-p%d" % (
    G["runs"][k]["schedule"], G["runs"][k]["train_epochs"], G["runs"][k]["epochs_per_eval"]))
def test_schedules_issue_the_reference_call_sequence(monkeypatch, k):
    run = G["runs"][k]
    ref_calls = run["calls"]
    listing = []                      # the directory order the reference saw (os.listdir order is arbitrary)
    for c in ref_calls:
        if c[0] == "train" and c[1] not in listing:
            listing.append(c[1])
    if run["schedule"] == "dynamic_train":
        listing = list(reversed(run["files"]))      # any order: the schedule sorts by name; the last file is only evaluated
    monkeypatch.setattr(cli, "list_files", lambda d: list(listing))
    monkeypatch.setattr(cli, "input_fn", lambda csv, img, mode, bs: (os.path.basename(csv), img, mode, bs))
    F = argparse.Namespace(train_epochs=run["train_epochs"], epochs_per_eval=run["epochs_per_eval"], batch_size=64,
                           train_data="/d/train", eval_data="/d/EVAL", test_data="/d/TEST", image_train_data=None,
                           image_eval_data=None, image_test_data=None)
    sched = {"train_and_eval": cli.schedule_train_and_eval, "dynamic_train": cli.schedule_dynamic, "train": cli.schedule_train_only}
    m, out = _Model(), io.StringIO()
    cli.run_schedule(m, F, sched[run["schedule"]](F), out)
    exp = [c[:5] for c in ref_calls]
    for c in exp:
        if c[0] == "evaluate" and c[1] == "TEST":
            assert c[3] == "pred"    # the reference evaluates the test set in 'pred' mode (no labels: cannot work, train.py:98)
            c[3] = "eval"            # deliberate deviation: labels are read
    assert m.calls == exp
    text = out.getvalue()
    assert text.count("auc: 0.5") == sum(1 for c in exp if c[0] == "evaluate")          # metrics printed `key: value`, sorted


def test_dynamic_schedule_needs_two_files(monkeypatch):
    monkeypatch.setattr(cli, "list_files", lambda d: ["only"])
    F = argparse.Namespace(train_epochs=1, epochs_per_eval=1, batch_size=8, train_data="/d", eval_data="/e", test_data="/t")
    with pytest.raises(AssertionError, match="Dynamic train mode need more than 1 data file"):
        list(cli.schedule_dynamic(F))
    with pytest.raises(ValueError, match="Must specify prediction data_file by --data_dir"):
        cli.pred_main([])


def test_train_eval_pred_mains_end_to_end_on_the_stand_in_engine(tmp_path, monkeypatch):
    """the three mains on real rows with the numpy stand-in engine of tests/test_estimator_host_cpu.py: the output lines the
    GPU tests (tests/test_gpu_c1.py) scrape are produced on this path too"""
    import types
    import torch
    from tests.test_estimator_host_cpu import StandInEngine, FIXTURE
    from tests.test_featurizer_host_cpu import _fake_call
    from wide_deep_amd import build_estimator as BE, estimator as EST, features as FE
    monkeypatch.setattr(FE, "call", _fake_call)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: types.SimpleNamespace(cuda_stream=0))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)

    def build(model_dir, model_type, conf=None, **kw):
        spec = BE.build_model_spec(conf or Config(), model_type)
        return EST.WideAndDeepClassifier(spec, model_dir=model_dir, runconfig=(conf or Config()).runconfig,
                                         engine=StandInEngine(spec, kw.get("max_batch", 128)))
    monkeypatch.setattr(cli, "build_custom_estimator", build)
    lines = open(FIXTURE, "rb").read().splitlines()
    d = tmp_path / "train"
    d.mkdir()
    (d / "part1").write_bytes(b"\n".join(lines[:300]) + b"\n")
    (d / "part2").write_bytes(b"\n".join(lines[300:]) + b"\n")
    pred = tmp_path / "pred.tsv"
    pred.write_bytes(b"\n".join(ln.split(b"\t", 1)[1] for ln in lines[:20]) + b"\n")
    root = str(tmp_path / "model")
    out = io.StringIO()
    model = cli.train_main(["--model_dir", root, "--train_data", str(d), "--train_epochs", "1", "--batch_size", "128",
                            "--model_type", "wide_deep"], out)
    text = out.getvalue()
    assert "Using dynamic train mode." in text and "auc:" in text and "examples/sec" in text and "Model Type: wide_deep" in text
    assert any(f.startswith("model.ckpt-") for f in os.listdir(os.path.join(root, "wide_deep")))
    ref = model.evaluate(input_fn=lambda: cli.input_fn(str(d / "part2"), None, "eval", 128))
    out = io.StringIO()
    cli.eval_main(["--model_dir", root, "--test_data", str(d / "part2"), "--batch_size", "128", "--model_type", "wide_deep"], out)
    got = dict(ln.split(": ", 1) for ln in out.getvalue().splitlines() if ": " in ln and not ln.startswith(("INFO", "Model")))
    for key in ("auc", "average_loss", "accuracy", "label/mean", "global_step"):
        assert abs(float(got[key]) - float(ref[key])) <= 1e-6 * max(1.0, abs(float(ref[key]))), key
    out = io.StringIO()
    n = cli.pred_main(["--model_dir", root, "--data_dir", str(pred), "--batch_size", "128", "--model_type", "wide_deep"], out)
    preds = [ln for ln in out.getvalue().splitlines() if ln.startswith("Prediction is")]
    assert n == 20 and len(preds) == 20 and all(p.startswith(('Prediction is "0"', 'Prediction is "1"')) for p in preds)


def test_estimator_methods_take_the_keyword_arguments_the_reference_scripts_pass():
    """python/train.py:72-87, 128-143 and python/pred.py:65-68 call the model with these keywords (tf.estimator.Estimator's)"""
    import inspect
    from wide_deep_amd.estimator import WideAndDeepClassifier as C
    names = lambda f: [p for p in inspect.signature(f).parameters if p != "self"]
    assert names(C.train) == ["input_fn", "hooks", "steps", "max_steps", "saving_listeners"]
    assert names(C.evaluate) == ["input_fn", "steps", "hooks", "checkpoint_path", "name"]
    assert names(C.predict) == ["input_fn", "predict_keys", "hooks", "checkpoint_path"]
    from wide_deep_amd import build_estimator as BE, dataset as DS
    assert names(BE.build_custom_estimator)[:2] == ["model_dir", "model_type"]
    assert names(DS.input_fn)[:4] == ["csv_data_file", "img_data_file", "mode", "batch_size"]
