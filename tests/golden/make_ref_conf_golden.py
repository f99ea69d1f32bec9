#!/usr/bin/env python
"""Golden vectors produced by EXECUTING the reference's own configuration reader (python/lib/read_conf.py, imported from
/root/reference -- it is plain Python + PyYAML, no TensorFlow) on the reference's shipped conf/*.yaml and on mutated copies.

Two environment shims, no change to reference code: PyYAML >= 6 needs an explicit Loader for `yaml.load(f)`, and Python 3
has no `unicode` builtin (read_conf.py:163).  `Config.get_feature_name` cannot run on Python 3 (`dict.values().remove`,
read_conf.py:266-267) and is not recorded.

Output: tests/golden/ref_conf.json
  "default":  what every reader method returns for the shipped configuration;
  "errors":   [mutation, exception class, message] -- the reference's error behaviour for invalid configurations.
tests/test_ref_conf_golden.py replays both against wide_deep_amd/read_conf.py.  Run in the build container only
(/root/reference does not exist on the GPU box)."""
import builtins
import copy
import json
import os
import sys
import tempfile

import yaml

_load = yaml.load
yaml.load = lambda stream, Loader=None: _load(stream, Loader=Loader or yaml.SafeLoader)
builtins.unicode = str
sys.path.insert(0, "/root/reference/python")
from lib import read_conf as R  # noqa: E402

REF_CONF = "/root/reference/conf"

# (file, path of keys into the parsed YAML, new value | "__delete__" | ("__rename__", new key))
MUTATIONS = [
    ("feature", ["os", "type"], None),
    ("feature", ["os", "type"], "categorical"),
    ("feature", ["os"], ("__rename__", "not_in_schema")),
    ("feature", ["os"], ("__rename__", "OS")),
    ("feature", ["os", "transform"], "nonsense"),
    ("feature", ["os", "transform"], None),
    ("feature", ["os", "parameter"], 7),
    ("feature", ["os", "parameter"], "wifi"),
    ("feature", ["adplan_id", "parameter"], "12"),
    ("feature", ["adplan_id", "parameter"], 12.5),
    ("feature", ["adplan_id", "parameter"], None),
    ("feature", ["idea_type", "parameter"], "3"),
    ("feature", ["age", "transform"], "sqrt"),
    ("feature", ["age", "transform"], None),
    ("feature", ["age", "transform"], "log"),
    ("feature", ["age", "transform"], "standard"),
    ("feature", ["age", "parameter", "normalization"], [1]),
    ("feature", ["age", "parameter", "normalization"], 5),
    ("feature", ["age", "parameter", "normalization"], ["a", 2]),
    ("feature", ["age", "parameter", "normalization"], [3, 2]),
    ("feature", ["age", "parameter", "normalization"], None),
    ("feature", ["age", "parameter", "boundaries"], 5),
    ("feature", ["age", "parameter", "boundaries"], [1, "x"]),
    ("feature", ["age", "parameter", "boundaries"], None),
    ("feature", ["age", "parameter"], [1, 2]),
    ("cross_feature", ["age&ugender", "hash_bucket_size"], None),
    ("cross_feature", ["age&ugender", "hash_bucket_size"], 0),
    ("cross_feature", ["age&ugender", "hash_bucket_size"], "10"),
    ("cross_feature", ["age&ugender", "hash_bucket_size"], 2.5),
    ("cross_feature", ["age&ugender", "is_deep"], 2),
    ("cross_feature", ["age&ugender", "is_deep"], None),
    ("cross_feature", ["age&ugender", "is_deep"], 0),
    ("cross_feature", ["age&ugender"], ("__rename__", "age")),
    ("cross_feature", ["age&ugender"], ("__rename__", "nosuch&ugender")),
    ("cross_feature", ["age&ugender"], ("__rename__", " age &  ugender ")),
    ("cross_feature", ["age&ugender"], ("__rename__", "ugender&age")),
]


def mutate(doc, path, value):
    doc = copy.deepcopy(doc)
    d = doc
    for k in path[:-1]:
        d = d[k]
    if isinstance(value, (tuple, list)) and len(value) == 2 and value[0] == "__rename__":
        d[value[1]] = d.pop(path[-1])
    elif value == "__delete__":
        del d[path[-1]]
    else:
        d[path[-1]] = value
    return doc


def jsonable(x):
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    return x


if __name__ == "__main__":
    c = R.Config()
    out = {"_source": __doc__.split("\n\n")[0],
           "default": {"schema": jsonable(c.read_schema()), "feature": jsonable(c.read_feature_conf()),
                       "cross_feature": jsonable(c.read_cross_feature_conf()), "train": jsonable(c.train),
                       "distribution": jsonable(c.distribution), "runconfig": jsonable(c.runconfig),
                       "model": jsonable(c.model), "config": jsonable(c.config)},
           "errors": []}
    docs = {k: yaml.safe_load(open(os.path.join(REF_CONF, k + ".yaml"))) for k in ("feature", "cross_feature")}
    with tempfile.TemporaryDirectory() as tmp:
        for which, path, value in MUTATIONS:
            doc = mutate(docs[which], path, value)
            p = os.path.join(tmp, which + ".yaml")
            yaml.safe_dump(doc, open(p, "w"))
            cfg = R.Config(**{which + "_conf_file": p})
            try:
                res = cfg.read_feature_conf() if which == "feature" else cfg.read_cross_feature_conf()
                rec = {"ok": jsonable(res) if which == "cross_feature" else True}
            except Exception as e:      # noqa: BLE001 -- the exception IS the recorded behaviour
                rec = {"exception": type(e).__name__, "message": str(e)}
            out["errors"].append({"file": which, "path": path, "value": jsonable(value), "result": rec})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_conf.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, len(out["errors"]), "mutations")
    for e in out["errors"]:
        print(e["file"], e["path"], e["value"], "->", str(e["result"])[:110])
