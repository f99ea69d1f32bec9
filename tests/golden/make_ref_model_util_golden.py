#!/usr/bin/env python
"""Golden behaviour of the reference's optimizer / activation selection, produced by EXECUTING
python/lib/utils/model_util.py (`get_optimizer_instance` :62-105, `activation_fn` :28-59) with a recording stub in place of
TensorFlow: `tf.train.Optimizer` is a real base class, every `tf.train.*Optimizer` a subclass that records its constructor
arguments, so the reference's own `eval(opt)` + isinstance check run unchanged.
Output: tests/golden/ref_model_util.json, replayed by tests/test_ref_model_util_golden.py against
wide_deep_amd.build_estimator.parse_optimizer / opt_tuple and the engine's activation table.
Run in the build container only (/root/reference does not exist on the GPU box)."""
import json
import os
import sys
import types


class Optimizer(object):
    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = list(args), dict(kwargs)


class _Fn(object):
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return self.name


tf = types.ModuleType("tensorflow")
tf.train = types.SimpleNamespace(Optimizer=Optimizer, SyncReplicasOptimizer=type("SyncReplicasOptimizer", (Optimizer,), {}))
for cls in ("AdagradOptimizer", "AdamOptimizer", "FtrlOptimizer", "RMSPropOptimizer", "GradientDescentOptimizer",
            "MomentumOptimizer", "AdadeltaOptimizer", "ProximalAdagradOptimizer"):
    setattr(tf.train, cls, type(cls, (Optimizer,), {}))
tf.nn = types.SimpleNamespace(**{n: _Fn("tf.nn." + n) for n in ("relu", "relu6", "leaky_relu", "crelu", "elu", "selu", "softplus", "softsign")})
tf.sigmoid, tf.tanh = _Fn("tf.sigmoid"), _Fn("tf.tanh")
sys.modules["tensorflow"] = tf
sys.path.insert(0, "/root/reference/python/lib/utils")
import model_util as MU  # noqa: E402

OPTS = [
    ("Adagrad", 0.05), ("Adam", 0.05), ("Ftrl", 0.2), ("RMSProp", 0.02), ("SGD", 0.05), ("Adagrad", None),
    ("tf.train.FtrlOptimizer(learning_rate=0.1,l1_regularization_strength=0.5,l2_regularization_strength=1)", 0.05),
    ("tf.train.FtrlOptimizer(learning_rate=0.1,l1_regularization_strength=0.5,l2_regularization_strength=1)", None),
    ("tf.train.AdagradOptimizer(learning_rate=0.05, initial_accumulator_value=0.1)", None),
    ("tf.train.AdamOptimizer(beta1=0.8, epsilon=1e-6)", 0.05),
    ("tf.train.RMSPropOptimizer(0.1, decay=0.5, momentum=0.3)", 0.05),
    ("tf.train.GradientDescentOptimizer(learning_rate=0.3)", 0.05),
    ("tf.train.ProximalAdagradOptimizer(learning_rate=0.1, l1_regularization_strength=0.001)", None),
    ("Adadelta", 0.1), ("Nadam", 0.1), ("adagrad", 0.1), ("tf.train.NoSuchOptimizer(0.1)", 0.1), ("tf.nn.relu", 0.1), ("0.5", 0.1),
]
ACTS = ["sigmoid", "tanh", "relu", "relu6", "leaky_relu", "crelu", "elu", "selu", "softplus", "softsign", "gelu", "ReLU", None]

if __name__ == "__main__":
    # the model conf strings the reference ships
    import yaml
    model = yaml.safe_load(open("/root/reference/conf/model.yaml"))
    shipped = []
    for k in ("linear_optimizer", "dnn_optimizer"):
        lr = model.get(k.replace("optimizer", "initial_learning_rate"))
        shipped.append((model[k], lr))
    out = {"_source": __doc__.split("\n\n")[0], "optimizers": [], "activations": []}
    for opt, lr in shipped + OPTS:
        try:
            o = MU.get_optimizer_instance(opt, lr)
            rec = {"class": type(o).__name__, "args": o.args, "kwargs": o.kwargs}
        except Exception as e:      # noqa: BLE001
            rec = {"exception": type(e).__name__, "message": str(e)}
        out["optimizers"].append({"opt": opt, "learning_rate": lr, "result": rec})
    for a in ACTS:
        try:
            rec = {"fn": repr(MU.activation_fn(a))}
        except Exception as e:      # noqa: BLE001
            rec = {"exception": type(e).__name__, "message": str(e)}
        out["activations"].append({"name": a, "result": rec})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_model_util.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for r in out["optimizers"] + out["activations"]:
        print(str(r)[:200])
