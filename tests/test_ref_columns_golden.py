"""CPU: the column wiring of wide_deep_amd.build_estimator.build_model_spec against golden vectors obtained by EXECUTING the
reference's `_build_model_columns` (python/lib/build_estimator.py:49-169) with a recording stub in place of TensorFlow
(tests/golden/make_ref_columns_golden.py -> tests/golden/ref_columns.json): every tf.feature_column call the reference
makes on its shipped configuration -- 70 wide columns, 70 deep columns, wide dimension 12,714,809, deep dimension 734."""
import json
import os

import numpy as np

from wide_deep_amd import build_estimator as BE
from wide_deep_amd.features import _normalize
from wide_deep_amd.read_conf import Config

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_columns.json")))
FC = "tf.feature_column."


def _probe_ok(normalizer, probe):
    for x, y in probe["lambda_probe"]:
        if isinstance(y, dict):                      # tf.log(x) recorded as a call on the stub
            assert y["fn"] == "tf.log" and normalizer and normalizer[0] == "log"
        else:
            got = _normalize(np.asarray([x], np.float32), normalizer)[0]
            assert abs(float(got) - y) <= 1e-6 * max(1.0, abs(y)), (normalizer, x, got, y)


def _leaf_name(k):
    if isinstance(k, str):
        return k
    if k["fn"] == FC + "bucketized_column":
        return k["args"][0]["args"][0] + "_bucketized"
    return k["args"][0]


def _check_categorical(slot, col):
    fn, kw = col["fn"][len(FC):], col["kwargs"]
    if fn == "categorical_column_with_hash_bucket":
        assert (slot.kind, slot.feature, slot.num_buckets, slot.name) == ("hash", col["args"][0], kw["hash_bucket_size"], col["args"][0])
        assert kw["dtype"] == "tf.string"
    elif fn == "categorical_column_with_vocabulary_list":
        assert (slot.kind, slot.feature, slot.name) == ("vocab", col["args"][0], col["args"][0])
        assert list(slot.vocab) == kw["vocabulary_list"] and slot.num_buckets == len(kw["vocabulary_list"])
        assert kw["default_value"] == -1 and kw["num_oov_buckets"] == 0          # out of vocabulary -> dropped
    elif fn == "categorical_column_with_identity":
        assert (slot.kind, slot.feature, slot.num_buckets, slot.name) == ("identity", col["args"][0], kw["num_buckets"], col["args"][0])
        assert kw["default_value"] == 0                                          # out of range -> bucket 0
    elif fn == "bucketized_column":
        num = col["args"][0]
        assert num["fn"] == FC + "numeric_column" and num["kwargs"]["shape"] == [1] and num["kwargs"]["default_value"] == 0
        assert (slot.kind, slot.feature, slot.name) == ("bucket", num["args"][0], num["args"][0] + "_bucketized")
        assert [float(b) for b in slot.boundaries] == [float(b) for b in kw["boundaries"]]
        assert slot.num_buckets == len(kw["boundaries"]) + 1
        if num["kwargs"].get("normalizer_fn") is not None:
            _probe_ok(slot.normalizer, num["kwargs"]["normalizer_fn"])           # wide bucketize sees the NORMALISED value
        else:
            assert not slot.normalizer
    elif fn == "crossed_column":
        keys, size = col["args"]
        assert slot.kind == "cross" and slot.num_buckets == int(size) and slot.hash_key == 0xDECAFCAFFE
        assert slot.name == "_X_".join(sorted(_leaf_name(k) for k in keys))
        assert len(slot.cross_keys) == len(keys)
        for ck, k in zip(slot.cross_keys, keys):                                 # key ORDER is part of the hash
            if isinstance(k, str):
                assert (ck.kind, ck.feature) == ("string", k)
            elif k["fn"] == FC + "categorical_column_with_identity":
                assert (ck.kind, ck.feature, ck.num_buckets) == ("identity", k["args"][0], k["kwargs"]["num_buckets"])
            else:
                num = k["args"][0]
                assert k["fn"] == FC + "bucketized_column" and "normalizer_fn" not in num["kwargs"]   # RAW value (quirk C.5)
                assert (ck.kind, ck.feature) == ("bucket", num["args"][0])
                assert [float(b) for b in ck.boundaries] == [float(b) for b in k["kwargs"]["boundaries"]]
    else:
        raise AssertionError("unexpected wide column " + fn)


def _col_name(col):
    fn = col["fn"][len(FC):]
    if fn == "crossed_column":
        return "_X_".join(sorted(_leaf_name(k) for k in col["args"][0]))
    if fn == "bucketized_column":
        return col["args"][0]["args"][0] + "_bucketized"
    return col["args"][0]


def test_wide_and_deep_columns_equal_the_reference_wiring():
    spec = BE.build_model_spec(Config(), "wide_deep")
    slots = {s.name: s for s in spec.slots}
    wide = {_col_name(c): c for c in G["wide"]}
    assert len(wide) == len(G["wide"]) == 70
    assert sorted(n for n, s in slots.items() if s.wide) == sorted(wide)
    for name, col in wide.items():
        _check_categorical(slots[name], col)
    assert sum(s.num_buckets for s in spec.slots if s.wide) == 12714809
    assert "Wide input dimension is: 12714809.0" in G["logged"] and "Deep input dimension is: 734" in G["logged"]

    deep_dim, n_emb, n_ind, n_num = 0, 0, 0, 0
    dense = {d.name: d for d in spec.dense_cols}
    for col in G["deep"]:
        fn, kw = col["fn"][len(FC):], col["kwargs"]
        if fn == "embedding_column":
            s = slots[_col_name(col["args"][0])]
            _check_categorical(s, col["args"][0])
            assert s.deep == "embedding" and s.dim == kw["dimension"] and kw.get("combiner", "mean") == "mean"
            deep_dim += s.dim
            n_emb += 1
        elif fn == "indicator_column":
            s = slots[_col_name(col["args"][0])]
            assert s.deep == "indicator"
            deep_dim += s.num_buckets
            n_ind += 1
        else:
            assert fn == "numeric_column" and kw["shape"] == [1] and kw["default_value"] == 0
            d = dense[col["args"][0]]
            kind = {0: None, 1: "min_max", 2: "standard", 3: "log"}[d.kind]
            if kw.get("normalizer_fn") is not None:
                _probe_ok((kind, d.p0, d.p1) if kind else None, kw["normalizer_fn"])
            else:
                assert kind is None
            deep_dim += 1
            n_num += 1
    assert (n_emb, n_ind, n_num) == (16 + 31, 20, 3) and deep_dim == 734
    assert n_num == len(spec.dense_cols)
    assert sum(1 for s in spec.slots if s.deep == "embedding") == n_emb and sum(1 for s in spec.slots if s.deep == "indicator") == n_ind


def test_input_layer_concat_order_follows_sorted_tf_column_names():
    """tf.feature_column.input_layer concatenates the deep columns sorted by `column.name` (`<cat>_embedding`,
    `<cat>_indicator`, the numeric key): FeaturePlan.tf_deep_cols / tf_input_perm (the row order of the first kernel in a
    checkpoint) must list exactly the reference's recorded deep columns, in that order, with their widths."""
    from wide_deep_amd.plan import FeaturePlan
    plan = FeaturePlan(BE.build_model_spec(Config(), "wide_deep"))
    exp = []
    for col in G["deep"]:
        fn = col["fn"][len(FC):]
        if fn == "embedding_column":
            exp.append((_col_name(col["args"][0]) + "_embedding", col["kwargs"]["dimension"]))
        elif fn == "indicator_column":
            inner = col["args"][0]
            n = len(inner["kwargs"]["vocabulary_list"]) if "vocabulary_list" in inner["kwargs"] else inner["kwargs"]["num_buckets"]
            exp.append((_col_name(inner) + "_indicator", n))
        else:
            exp.append((col["args"][0], 1))
    exp.sort(key=lambda t: t[0])
    assert [(n, w) for n, _, w in plan.tf_deep_cols] == exp
    assert plan.tf_deep_dim == sum(w for _, w in exp) == 734 and len(set(plan.tf_input_perm.tolist())) == 734
    assert plan.deep_dim >= 734 and plan.deep_dim % 4 == 0          # internal width: alignment holes / slab padding only
