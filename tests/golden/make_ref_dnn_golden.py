#!/usr/bin/env python
"""Golden tower wiring produced by EXECUTING the reference's `_dnn_logit_fn` and `multidnn_logit_fn_builder`
(python/lib/dnn.py:43-275) with a recording stub in place of TensorFlow: every tf.layers.dense / dropout /
batch_normalization / tf.concat / tf.add_n call is captured as a dataflow graph whose leaves are the input layer, for each
connected mode, with and without dropout / batch normalisation, in TRAIN and EVAL mode, and for a two-tower multi-DNN.
Output: tests/golden/ref_dnn_graphs.json; tests/test_ref_dnn_golden.py interprets the graphs in numpy with seeded weights and
compares with oracle.tower_forward -- i.e. the oracle's restatement of dnn.py is checked against dnn.py itself.

Environment shims only: PyYAML Loader default, `unicode`, stub modules for tensorflow (+ tensorflow.python.estimator.canned).
DROPOUT / BATCH_NORM are module-level constants of dnn.py read from conf/model.yaml at import; the script sets them per case
(configuration, not code).  Run in the build container only."""
import builtins
import json
import os
import sys
import types

import yaml

_load = yaml.load
yaml.load = lambda stream, Loader=None: _load(stream, Loader=Loader or yaml.SafeLoader)
builtins.unicode = str

NODES = []


class T(object):
    """symbolic tensor / scope: result of a recorded call"""

    def __init__(self, op, inputs=(), attrs=None):
        self.id, self.op, self.inputs, self.attrs = len(NODES), op, list(inputs), dict(attrs or {})
        self.name = "%s_%d" % (op, self.id)
        NODES.append(self)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


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

    def __mro_entries__(self, bases):       # `class MultiDNNClassifier(tf.estimator.Estimator)` at dnn.py:298
        return (object,)

    def __call__(self, *a, **kw):
        n = self._name
        if n == "tf.layers.dense":
            scope = kw.get("name")
            return T("dense", [a[0]], {"units": kw["units"], "activation": repr(kw["activation"]) if kw.get("activation") is not None else None,
                                       "use_bias": kw.get("use_bias", True), "scope": scope.attrs.get("scope") if isinstance(scope, T) else None})
        if n == "tf.layers.dropout":
            return T("dropout", [a[0]], {"rate": kw["rate"], "training": kw["training"]})
        if n == "tf.layers.batch_normalization":
            return T("batch_normalization", [a[0]], {k: v for k, v in kw.items()})
        if n == "tf.concat":
            return T("concat", list(a[0]), {"axis": kw.get("axis", a[1] if len(a) > 1 else None)})
        if n == "tf.add_n":
            return T("add_n", list(a[0]))
        if n == "tf.feature_column.input_layer":
            return T("input_layer")
        if n == "tf.variable_scope":
            return T("scope", [], {"scope": a[0], "reuse": repr(kw.get("reuse"))})
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
sys.path.insert(0, "/root/reference/python")
from lib import dnn as RD  # noqa: E402


def graph_of(out):
    """nodes reachable from `out`, as JSON (scopes dropped)"""
    seen, order = {}, []

    def walk(t):
        if t.id in seen:
            return
        for i in t.inputs:
            walk(i)
        seen[t.id] = len(order)
        order.append(t)
    walk(out)
    return [{"op": t.op, "inputs": [seen[i.id] for i in t.inputs], "attrs": t.attrs} for t in order]


if __name__ == "__main__":
    TRAIN = tf.estimator.ModeKeys.TRAIN
    EVAL = tf.estimator.ModeKeys.EVAL
    cases = []
    for mode in ("simple", "first_dense", "last_dense", "lase_dense", "dense", "resnet"):
        for hidden in ([5, 3, 2], [4, 6, 3, 2]) if mode not in ("last_dense", "lase_dense") else ([5, 3, 2],):
            for dropout, bn in ((None, True), (0.25, True), (0.25, False), (None, False)):
                for run_mode, run_name in ((TRAIN, "train"), (EVAL, "eval")):
                    RD.DROPOUT, RD.BATCH_NORM = dropout, bn
                    del NODES[:]
                    rec = {"connected_mode": mode, "hidden_units": hidden, "dropout": dropout, "batch_norm": bn, "mode": run_name}
                    try:
                        out = RD._dnn_logit_fn({"f": T("feature")}, run_mode, 1, 1, hidden, mode, ["cols"], None)
                        rec["graph"] = graph_of(out)
                    except Exception as e:      # noqa: BLE001 -- e.g. the 'lase_dense' typo of dnn.py:77
                        rec["exception"] = {"class": type(e).__name__, "message": str(e)}
                    cases.append(rec)
    # multi-DNN: logits of the towers are added (dnn.py:260-274)
    RD.DROPOUT, RD.BATCH_NORM = None, True
    del NODES[:]
    fn = RD.multidnn_logit_fn_builder(1, [[5, 3], [4, 2, 2]], ["simple", "dense"], ["cols"], None)
    out = fn({"f": T("feature")}, TRAIN)
    multi = {"hidden_units": [[5, 3], [4, 2, 2]], "connected_mode": ["simple", "dense"], "graph": graph_of(out)}
    shipped = {"activation": repr(RD.ACTIVATION_FN), "dnn_l1": RD.DNN_L1, "dnn_l2": RD.DNN_L2, "reg": RD.REG.op if RD.REG is not None else None}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_dnn_graphs.json")
    json.dump({"_source": __doc__.split("\n\n")[0], "cases": cases, "multi": multi, "shipped": shipped}, open(dst, "w"), sort_keys=True)
    ok = sum(1 for c in cases if "graph" in c)
    print("wrote", dst, len(cases), "cases,", ok, "graphs; exceptions:", sorted({(c["connected_mode"], c["exception"]["message"]) for c in cases if "exception" in c}))
    print(shipped)
