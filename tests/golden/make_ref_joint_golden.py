#!/usr/bin/env python
"""Golden train-op wiring produced by EXECUTING the reference's `_wide_deep_combined_model_fn` (python/lib/joint.py:81-269)
with a recording stub in place of TensorFlow, for model_type wide / deep / wide_deep with the shipped conf/model.yaml:
which optimizer (class + constructor arguments, as built by the reference's own get_optimizer_instance) minimises the
variables of which scope, what the learning rate object is (an exponential_decay over a FRESH `tf.Variable(0)` that nothing
increments: quirk C.2), how many times the real global step is advanced per batch (each minimize(global_step=...) + the
final assign_add: quirk C.4), how the logits are combined, and the linear_model / head arguments.
Output: tests/golden/ref_joint.json, replayed by tests/test_ref_joint_golden.py.
Environment shims only (PyYAML Loader default, `unicode`, stub modules).  Run in the build container only."""
import builtins
import json
import os
import sys
import types

import yaml

_load = yaml.load
yaml.load = lambda stream, Loader=None: _load(stream, Loader=Loader or yaml.SafeLoader)
builtins.unicode = str

EVENTS = []


class T(object):
    n = 0

    def __init__(self, op, **attrs):
        T.n += 1
        self.op, self.attrs, self.name = op, attrs, "%s_%d" % (op, T.n)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def desc(self):
        return {"op": self.op, **{k: enc(v) for k, v in self.attrs.items()}}


def enc(v):
    if isinstance(v, T):
        return v.desc()
    if isinstance(v, Optimizer):
        return v.desc()
    if isinstance(v, (list, tuple)):
        return [enc(x) for x in v]
    if isinstance(v, Rec):
        return repr(v)
    if isinstance(v, dict):
        return {k: enc(x) for k, x in v.items()}
    return v


class Optimizer(object):
    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = list(args), dict(kwargs)

    def desc(self):
        return {"class": type(self).__name__, "args": enc(self.args), "kwargs": enc(self.kwargs)}

    def minimize(self, loss, global_step=None, var_list=None):
        EVENTS.append({"event": "minimize", "optimizer": self.desc(), "global_step": enc(global_step), "var_list": enc(var_list)})
        return T("minimize_op", advances_global_step=global_step is not None)


class Rec(object):
    def __init__(self, name):
        self._name = name

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        r = Rec(self._name + "." + k)
        setattr(self, k, r)
        return r

    def __repr__(self):
        return self._name

    def __mro_entries__(self, bases):
        return (object,)

    def __call__(self, *a, **kw):
        n = self._name
        if n == "tf.Variable":
            return T("fresh_variable", initial_value=a[0])
        if n == "tf.train.exponential_decay":
            return T("exponential_decay", initial=a[0], global_step=kw["global_step"], decay_steps=kw["decay_steps"],
                     decay_rate=kw["decay_rate"], staircase=kw["staircase"])
        if n == "tf.train.get_global_step":
            return T("the_global_step")
        if n == "tf.get_collection":
            return T("collection", key=repr(a[0]), scope=kw.get("scope"))
        if n == "tf.assign_add":
            EVENTS.append({"event": "assign_add", "target": enc(a[0]), "value": a[1]})
            return T("assign_add")
        if n == "tf.add_n":
            EVENTS.append({"event": "add_n", "inputs": enc(list(a[0]))})
            return T("logits_sum")
        if n == "tf.feature_column.linear_model":
            EVENTS.append({"event": "linear_model", "kwargs": {k: enc(v) for k, v in kw.items() if k not in ("features",)}})
            return T("linear_logits")
        if n == "tf.feature_column.input_layer":
            return T("input_layer")
        if n == "tf.layers.dense":
            return T("dense", units=kw.get("units"))
        if n in ("tf.layers.batch_normalization", "tf.layers.dropout"):
            return a[0]
        if n == "tf.variable_scope":
            return T("scope", scope=a[0] if isinstance(a[0], str) else enc(a[0]))
        if n == "tf.group":
            return T("group", n=len(a))
        return T("other:" + n)


class StubModule(types.ModuleType):
    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        r = Rec(self.__name__.replace("tensorflow", "tf") + "." + k)
        setattr(self, k, r)
        return r


for m in ("tensorflow", "tensorflow.python", "tensorflow.python.estimator", "tensorflow.python.estimator.canned"):
    sys.modules[m] = StubModule(m)
tf = sys.modules["tensorflow"]
train = Rec("tf.train")
train.Optimizer = Optimizer
train.SyncReplicasOptimizer = type("SyncReplicasOptimizer", (Optimizer,), {})
for cls in ("AdagradOptimizer", "AdamOptimizer", "FtrlOptimizer", "RMSPropOptimizer", "GradientDescentOptimizer"):
    setattr(train, cls, type(cls, (Optimizer,), {}))
tf.train = train
vgg = types.ModuleType("lib.cnn.vgg")
vgg.Vgg16 = Rec("lib.cnn.vgg.Vgg16")
cnn = types.ModuleType("lib.cnn")
cnn.__path__ = []
sys.modules["lib.cnn"], sys.modules["lib.cnn.vgg"] = cnn, vgg
sys.path.insert(0, "/root/reference/python")
from lib import joint as RJ  # noqa: E402


class Head(object):
    logits_dimension = 1

    def create_estimator_spec(self, features, mode, labels, train_op_fn, logits):
        EVENTS.append({"event": "head", "mode": repr(mode), "logits": enc(logits)})
        train_op_fn(T("loss"))
        return "spec"


if __name__ == "__main__":
    model = yaml.safe_load(open("/root/reference/conf/model.yaml"))
    out = {"_source": __doc__.split("\n\n")[0], "constants": {
        "linear_init_learning_rate": RJ._linear_init_learning_rate, "dnn_init_learning_rate": RJ._dnn_init_learning_rate,
        "linear_decay_rate": RJ._linear_decay_rate, "dnn_decay_rate": RJ._dnn_decay_rate, "decay_steps": RJ.decay_steps},
        "model_types": {}}
    for mt in ("wide", "deep", "wide_deep"):
        del EVENTS[:]
        RJ._wide_deep_combined_model_fn(
            {"f": T("feature")}, T("labels"), tf.estimator.ModeKeys.TRAIN, Head(), mt,
            linear_feature_columns=["wide_cols"], linear_optimizer=model["linear_optimizer"],
            dnn_feature_columns=["deep_cols"], dnn_optimizer=model["dnn_optimizer"],
            dnn_hidden_units=model["dnn_hidden_units"], dnn_connected_mode=model["dnn_connected_mode"], config=None)
        out["model_types"][mt] = list(EVENTS)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_joint.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(out["constants"]))
    for mt, ev in out["model_types"].items():
        print(mt)
        for e in ev:
            print("   ", json.dumps(e)[:260])
