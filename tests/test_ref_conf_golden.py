"""CPU: wide_deep_amd/read_conf.py against golden vectors obtained by EXECUTING the reference's own reader
(python/lib/read_conf.py) on its shipped configuration and on 36 mutated configurations
(tests/golden/make_ref_conf_golden.py -> tests/golden/ref_conf.json): same parsed values from this repo's conf/*.yaml, same
exception class and message for every invalid configuration (errors are part of the interface)."""
import copy
import json
import os

import pytest
import yaml

from wide_deep_amd.read_conf import Config, conf_dir

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_conf.json")))


def _jsonable(x):
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    return x


def test_shipped_configuration_parses_to_the_reference_values():
    c, d = Config(), G["default"]
    assert _jsonable(c.read_schema()) == d["schema"]
    assert _jsonable(c.read_feature_conf()) == d["feature"]
    assert list(c.read_feature_conf()) == list(d["feature"]) or sorted(c.read_feature_conf()) == sorted(d["feature"])
    # x1000 bucket sizes (0.1 -> 100.0 stays a float), is_deep default.  This repo's cross_feature.yaml lists the crosses
    # sorted by name, the reference's in hand-written order: immaterial, linear_model / input_layer sort columns by name
    key = lambda e: "&".join(e[0])
    assert sorted(_jsonable(c.read_cross_feature_conf()), key=key) == sorted(d["cross_feature"], key=key)
    assert all(type(a[1]) is type(b[1]) for a, b in zip(sorted(_jsonable(c.read_cross_feature_conf()), key=key),
                                                        sorted(d["cross_feature"], key=key)))
    for key in ("train", "distribution", "runconfig", "model"):
        got, exp = _jsonable(getattr(c, key)), d[key]
        for k, v in exp.items():
            if key == "train" and k in ("model_dir", "train_data", "eval_data", "test_data", "pred_data", "image_train_data",
                                        "image_eval_data", "image_test_data"):
                continue                                                       # paths are relative to each repo's layout
            assert k in got and got[k] == v, (key, k, got.get(k), v)


def _mutate(doc, path, value):
    doc = copy.deepcopy(doc)
    d = doc
    for k in path[:-1]:
        d = d[k]
    if isinstance(value, list) and len(value) == 2 and value[0] == "__rename__":
        d[value[1]] = d.pop(path[-1])
    else:
        d[path[-1]] = value
    return doc


@pytest.mark.parametrize("case", G["errors"], ids=lambda c: "%s:%s=%s" % (c["file"], "/".join(c["path"]), str(c["value"])[:24]))
def test_invalid_configurations_fail_like_the_reference(tmp_path, case):
    which = case["file"]
    doc = yaml.safe_load(open(os.path.join(conf_dir(), which + ".yaml")))
    p = tmp_path / (which + ".yaml")
    yaml.safe_dump(_mutate(doc, case["path"], case["value"]), open(p, "w"))
    cfg = Config(**{which + "_conf_file": str(p)})
    run = cfg.read_feature_conf if which == "feature" else cfg.read_cross_feature_conf
    exp = case["result"]
    if "exception" in exp:
        with pytest.raises(Exception) as ei:
            run()
        assert type(ei.value).__name__ == exp["exception"]
        assert str(ei.value) == exp["message"]
    else:
        res = run()
        if which == "cross_feature":
            assert _jsonable(res) == exp["ok"]
