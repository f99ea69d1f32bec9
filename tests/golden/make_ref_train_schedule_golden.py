#!/usr/bin/env python
"""Golden call sequences produced by EXECUTING the three training schedules of the reference's python/train.py
(`train_and_eval` :65-93, `dynamic_train` :96-148, `train` :151-165) with a recording model object: which Estimator method
is called on which file, in which input_fn mode, with which batch size, in what order.
Stub modules stand in for tensorflow (tf.gfile mapped onto os), lib.dataset.input_fn (records its arguments) and
lib.build_estimator; FLAGS is the argparse namespace the script would have parsed.
Output: tests/golden/ref_train_schedule.json, replayed by tests/test_cli_cpu.py.  Run in the build container only."""
import argparse
import builtins
import json
import os
import sys
import tempfile
import types

import yaml

_load = yaml.load
yaml.load = lambda stream, Loader=None: _load(stream, Loader=Loader or yaml.SafeLoader)
builtins.unicode = str


class Rec(object):
    def __init__(self, name):
        self._name = name

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        r = Rec(self._name + "." + k)
        setattr(self, k, r)
        return r

    def __call__(self, *a, **kw):
        return None


tf = types.ModuleType("tensorflow")
tf.logging = Rec("tf.logging")
tf.gfile = types.SimpleNamespace(IsDirectory=os.path.isdir, ListDirectory=os.listdir, Exists=os.path.exists)
tf.data = Rec("tf.data")
tf.int32, tf.string, tf.float32 = "int32", "string", "float32"
sys.modules["tensorflow"] = tf
CALLS = []
ds = types.ModuleType("lib.dataset")
ds.input_fn = lambda csv, img, mode, batch_size: ("input", os.path.basename(csv), img, mode, batch_size)
be = types.ModuleType("lib.build_estimator")
be.build_estimator = be.build_custom_estimator = None
sys.modules["lib.dataset"], sys.modules["lib.build_estimator"] = ds, be
sys.path.insert(0, "/root/reference/python")
sys.argv = ["train.py"]
import train as RT  # noqa: E402


class Model(object):
    def train(self, input_fn, hooks=None, steps=None, max_steps=None, saving_listeners=None):
        CALLS.append(["train"] + list(input_fn()[1:]) + [steps, max_steps])

    def evaluate(self, input_fn, steps=None, hooks=None, checkpoint_path=None, name=None):
        CALLS.append(["evaluate"] + list(input_fn()[1:]) + [steps, checkpoint_path])
        return {"auc": 0.5}


if __name__ == "__main__":
    out = {"_source": __doc__.split("\n\n")[0], "runs": []}
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "train")
        os.mkdir(d)
        for name in ("part2", "part1", "part3", ".hidden"):
            open(os.path.join(d, name), "w").write("x\n")
        for epochs, per_eval in ((2, 2), (3, 2), (1, 1)):
            RT.FLAGS = argparse.Namespace(train_epochs=epochs, epochs_per_eval=per_eval, batch_size=64, train_data=d,
                                          eval_data=os.path.join(tmp, "EVAL"), test_data=os.path.join(tmp, "TEST"),
                                          image_train_data=None, image_eval_data=None, image_test_data=None)
            for fn in ("train_and_eval", "dynamic_train", "train"):
                del CALLS[:]
                devnull = open(os.devnull, "w")
                old, sys.stdout = sys.stdout, devnull
                try:
                    getattr(RT, fn)(Model())
                finally:
                    sys.stdout = old
                out["runs"].append({"schedule": fn, "train_epochs": epochs, "epochs_per_eval": per_eval, "files": ["part1", "part2", "part3"],
                                    "calls": [list(c) for c in CALLS]})
    # the three argparse parsers, as the reference scripts build them at import (defaults come from conf/train.yaml)
    import importlib

    def flags_of(parser):
        return [{"flag": a.option_strings[0], "type": getattr(a.type, "__name__", None), "default": a.default}
                for a in parser._actions if a.option_strings and a.option_strings[0] != "-h"]
    out["parsers"] = {"train": flags_of(RT.parser)}
    for name in ("eval", "pred"):
        sys.argv = [name + ".py"]
        out["parsers"][name] = flags_of(importlib.import_module(name).parser)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_train_schedule.json")
    json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
    print({k: [f["flag"] for f in v] for k, v in out["parsers"].items()})
    for r in out["runs"][:1]:
        print(r["schedule"], r["train_epochs"], r["epochs_per_eval"], r["calls"])
