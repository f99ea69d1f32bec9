#!/usr/bin/env python
"""Golden column wiring produced by EXECUTING the reference's `_build_model_columns` (python/lib/build_estimator.py:49-169)
on its shipped conf/*.yaml, with the `tensorflow` module replaced by a RECORDING stub: every `tf.feature_column.*` call the
reference makes is captured with its arguments (hash bucket sizes, embedding dimensions, vocabularies, boundaries, identity
ranges, crossed keys and their bucket sizes, which columns go to the wide / deep side), the normalizer lambdas are probed on
sample values, and the wide / deep input dimensions the function logs are kept.  TensorFlow itself is not needed for this
part of the path: the function only BUILDS column objects.

Environment shims only (no reference code is changed): PyYAML Loader default, `unicode`, and stub modules for `tensorflow`,
`lib.joint`, `lib.utils.model_util` (which import TF internals at module level).
Output: tests/golden/ref_columns.json; replayed by tests/test_ref_columns_golden.py against wide_deep_amd.build_estimator.
Run in the build container only (/root/reference does not exist on the GPU box)."""
import builtins
import json
import os
import sys
import types

import yaml

_load = yaml.load
yaml.load = lambda stream, Loader=None: _load(stream, Loader=Loader or yaml.SafeLoader)
builtins.unicode = str

LOG = []


class Node(object):
    """result of a recorded call"""

    def __init__(self, fn, args, kwargs):
        self.fn, self.args, self.kwargs = fn, args, kwargs

    def to_json(self):
        return {"fn": self.fn, "args": [enc(a) for a in self.args], "kwargs": {k: enc(v) for k, v in sorted(self.kwargs.items())}}


class Rec(object):
    """attribute chain under the stub module: calling it records (dotted name, args, kwargs)"""

    def __init__(self, name):
        self._name = name

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        r = Rec(self._name + "." + k)
        setattr(self, k, r)
        return r

    def __call__(self, *a, **kw):
        fix = lambda v: list(v) if isinstance(v, map) else v          # Python 2 `map` returned a list (build_estimator.py:103)
        a, kw = tuple(fix(v) for v in a), {k: fix(v) for k, v in kw.items()}
        n = Node(self._name, a, kw)
        LOG.append(n)
        return n

    def __mro_entries__(self, bases):
        return (object,)

    def __repr__(self):
        return self._name


def enc(v):
    if isinstance(v, Node):
        return v.to_json()
    if isinstance(v, Rec):
        return repr(v)
    if isinstance(v, (map, tuple, list)):
        return [enc(x) for x in v]
    if callable(v):      # normalizer_fn lambdas: probe them (tf.log is recorded as a call on the stub)
        out = []
        for x in (0.0, 1.0, 25.0, 100.0):
            y = v(x)
            out.append([x, y.to_json() if isinstance(y, Node) else float(y)])
        return {"lambda_probe": out}
    if hasattr(v, "item"):
        return v.item()
    return v


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


sys.modules["tensorflow"] = StubModule("tensorflow")
for name, attrs in (("lib.joint", ["WideAndDeepClassifier"]), ("lib.utils", []), ("lib.utils.model_util", ["activation_fn"])):
    m = types.ModuleType(name)
    m.__path__ = []
    for a in attrs:
        setattr(m, a, Rec(name + "." + a))
    sys.modules[name] = m
sys.path.insert(0, "/root/reference/python")
import lib  # noqa: E402
lib.joint, lib.utils = sys.modules["lib.joint"], sys.modules["lib.utils"]
from lib import build_estimator as RB  # noqa: E402

if __name__ == "__main__":
    del LOG[:]
    wide, deep = RB._build_model_columns()
    logs = [n.args[0] for n in LOG if n.fn == "tf.logging.info"]
    out = {"_source": __doc__.split("\n\n")[0], "wide": [enc(c) for c in wide], "deep": [enc(c) for c in deep], "logged": logs}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_columns.json")
    json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
    print("wrote", dst, len(wide), "wide", len(deep), "deep")
    print("\n".join(logs))
