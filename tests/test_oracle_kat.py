"""CPU: pin the oracle's integer path against the upstream-TF known answers (tests/golden/kat_hash.json)
and cross-check the C restatement against the independent Python transcription on every length branch."""
import json
import os
import random

import numpy as np

from oracle import farmhash_py as F
from oracle import oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_hash.json")))


def _csr(lists, conv):
    vals, offs = [], [0]
    for row in lists:
        vals += [conv(v) for v in row]
        offs.append(len(vals))
    return np.asarray(vals, dtype=np.uint64), np.asarray(offs, dtype=np.int32)


def test_fingerprint64_kat():
    for s, exp in GOLD["fingerprint64"].items():
        assert O.fingerprint64(s.encode()) == exp
        assert F.fingerprint64(s.encode()) == exp


def test_fingerprint64_le32_bytes_against_independent_cityhash_build():
    """farmhashna::Hash64 == CityHash64 v1.1 for len <= 32 (shared HashLen0to16 / HashLen17to32): vectors produced by abseil's
    compiled CityHash64 (tests/golden/make_city_vectors.py) pin the 17..32-byte branch that no TF known answer covers."""
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_city_le32.json")))["vectors"]
    lens = set()
    for hx, exp in g:
        s = bytes.fromhex(hx)
        lens.add(len(s))
        assert O.fingerprint64(s) == exp, len(s)
        assert F.fingerprint64(s) == exp, len(s)
    assert lens >= set(range(0, 33))
    toks = [bytes.fromhex(hx) for hx, _ in g]
    data, offs = O.pack_tokens(toks)
    assert O.fingerprint64_batch(data, offs).tolist() == [e for _, e in g]


def test_fingerprint64_published_answers_cover_every_length_branch():
    """Guava FarmHashFingerprint64Test + BigQuery FARM_FINGERPRINT doc examples (tests/helpers.py): 4 / 32 / 256-byte
    strings and the 3200-message chain over lengths 0..3200 -- the 33..64-byte and > 64-byte branches included."""
    from tests.helpers import GUAVA_SIMPLE, BIGQUERY_DOC, GUAVA_MULTIPLE_LENGTHS, guava_multiple_lengths
    for s, exp in GUAVA_SIMPLE + BIGQUERY_DOC:
        assert O.fingerprint64(s) == exp and F.fingerprint64(s) == exp
    rec = []
    assert guava_multiple_lengths(O.fingerprint64, rec) == GUAVA_MULTIPLE_LENGTHS
    lens = {len(m) for m in rec}
    assert len(rec) == 3200 and lens >= set(range(0, 800)) and max(lens) == 3199
    # the independent Python transcription on the same messages (batch C path too)
    data, offs = O.pack_tokens(rec)
    fps = O.fingerprint64_batch(data, offs).tolist()
    assert all(F.fingerprint64(m) == f for m, f in list(zip(rec, fps))[::7])
    it = iter(fps)
    assert guava_multiple_lengths(lambda m: next(it)) == GUAVA_MULTIPLE_LENGTHS


def test_hash_bucket_kat():
    for toks, exp in GOLD["hash_bucket_10"]:
        assert O.hash_bucket(toks, 10).tolist() == exp


def test_cross_kat():
    g = GOLD["cross_3keys"]
    cols = [_csr([[t]], lambda s: O.fingerprint64(s.encode())) for t in g["tokens"]]
    assert O.cross_hash(cols, 0)[0].tolist() == [g["buckets0"]]
    assert O.cross_hash(cols, 100)[0].tolist() == [g["buckets100"]]


def test_crossed_column_int_and_string_keys():
    g = GOLD["crossed_column_key5"]
    fp = lambda s: O.fingerprint64(s.encode())
    a = _csr(g["a_ids"], int)
    c, d1, d2 = _csr(g["c"], fp), _csr(g["d1"], fp), _csr(g["d2"], fp)
    ids, offs = O.cross_hash([a, c], 5, hash_key=5)
    assert ids.tolist() == g["a_c_5"] and offs.tolist() == [0, 2, 6]
    ids, offs = O.cross_hash([a, c, d1, d2], 15, hash_key=5)
    assert ids.tolist() == g["a_c_d1_d2_15"] and offs.tolist() == [0, 2, 18]


def test_c_matches_python_on_all_length_branches():
    rnd = random.Random(7)
    for n in list(range(0, 260)) + [511, 512, 513, 1000, 4097]:
        s = bytes(rnd.getrandbits(8) for _ in range(n))
        assert O.fingerprint64(s) == F.fingerprint64(s), n
    for _ in range(200):
        a, b = rnd.getrandbits(64), rnd.getrandbits(64)
        assert O.fingerprint_cat64(a, b) == F.fingerprint_cat64(a, b)


def test_cross_empty_key_gives_empty_row():
    a = (np.asarray([3, 4], dtype=np.uint64), np.asarray([0, 2, 2], dtype=np.int32))   # row1 empty
    b = (np.asarray([7, 8, 9], dtype=np.uint64), np.asarray([0, 1, 3], dtype=np.int32))
    ids, offs = O.cross_hash([a, b], 50)
    assert offs.tolist() == [0, 2, 2]
    exp = [O.fingerprint_cat64(O.fingerprint_cat64(0xDECAFCAFFE, 3), 7) % 50,
           O.fingerprint_cat64(O.fingerprint_cat64(0xDECAFCAFFE, 4), 7) % 50]
    assert ids.tolist() == exp


def test_bucketize_boundaries():
    # tf bucketized_column: buckets (-inf,b0),[b0,b1),...,[bn-1,inf)  
    assert O.bucketize([-1.0, 0.0, 0.5, 1.0, 2.0], [0.0, 1.0]).tolist() == [0, 1, 1, 2, 2]


def test_ftrl_and_adagrad_formulas():
    import torch
    w = torch.zeros(4, 1); z = torch.zeros(4, 1); n = torch.full((4, 1), 0.1)
    g = torch.tensor([[2.0], [-3.0]])
    O.ftrl_rows(w, z, n, np.asarray([1, 3]), g, 0.1, 0.5, 1.0)
    # hand evaluation of TF FtrlOptimizer(lr_power=-0.5): row 1, g=2
    n_new = 0.1 + 4.0
    zz = 2.0 - (np.sqrt(np.float32(n_new)) - np.sqrt(np.float32(0.1))) / 0.1 * 0.0
    quad = np.sqrt(np.float32(n_new)) / 0.1 + 2.0
    exp = (0.5 - zz) / quad
    assert abs(float(w[1, 0]) - exp) < 1e-6 and float(w[0, 0]) == 0.0 and abs(float(n[1, 0]) - n_new) < 1e-6
    assert float(w[3, 0]) > 0  # negative gradient -> positive weight
    t = torch.ones(2, 3); acc = torch.full((2, 3), 0.1)
    O.adagrad_rows(t, acc, np.asarray([0]), torch.full((1, 3), 0.5), 0.05)
    assert torch.allclose(t[0], torch.full((3,), 1 - 0.05 * 0.5 / np.sqrt(0.35)), atol=1e-6) and torch.all(t[1] == 1)


def test_optimizer_restatements_hand_computed():
    """tf.train.{GradientDescent,RMSProp,Adam} (dense and sparse forms) on numbers small enough to follow by hand."""
    import numpy as np
    import torch
    from oracle import oracle as O
    f = lambda *v: torch.tensor(v, dtype=torch.float32)
    # SGD dense / sparse
    st = {"w": f(1.0, 2.0)}
    O.opt_apply_dense(("SGD", 0.1), st, "w", f(10.0, -10.0), None)
    assert torch.allclose(st["w"], f(0.0, 3.0))
    st = {"t": torch.ones(3, 2)}
    O.opt_apply_rows(("SGD", 0.5), st, "t", np.asarray([2]), f(1.0, 2.0).reshape(1, 2), None)
    assert torch.allclose(st["t"], torch.tensor([[1.0, 1.0], [1.0, 1.0], [0.5, 0.0]]))
    # RMSProp: ms = 1 + (4 - 1) * 0.1 = 1.3; mom = 0.5 * 0 + 0.1 * 2 / sqrt(1.3 + 0) ; var = 1 - mom
    st = {"w": f(1.0), "w/RMSProp": f(1.0), "w/RMSProp_1": f(0.0)}
    O.opt_apply_dense(("RMSProp", 0.1, 0.9, 0.5, 0.0), st, "w", f(2.0), None)
    mom = 0.1 * 2 / np.sqrt(1.3)
    assert abs(float(st["w/RMSProp"]) - 1.3) < 1e-6 and abs(float(st["w/RMSProp_1"]) - mom) < 1e-6
    assert abs(float(st["w"]) - (1 - mom)) < 1e-6
    O.opt_apply_dense(("RMSProp", 0.1, 0.9, 0.5, 0.0), st, "w", f(0.0), None)    # momentum keeps moving it
    assert abs(float(st["w/RMSProp_1"]) - 0.5 * mom) < 1e-6 and abs(float(st["w"]) - (1 - 1.5 * mom)) < 1e-6
    # RMSProp sparse: untouched rows keep their slots and values
    st = {"t": torch.ones(2, 1), "t/RMSProp": torch.ones(2, 1), "t/RMSProp_1": torch.zeros(2, 1)}
    O.opt_apply_rows(("RMSProp", 0.1, 0.9, 0.0, 0.0), st, "t", np.asarray([1]), f(2.0).reshape(1, 1), None)
    assert float(st["t"][0]) == 1.0 and float(st["t/RMSProp"][0]) == 1.0 and abs(float(st["t"][1]) - (1 - mom)) < 1e-6
    # Adam dense, t = 1: m = 0.1 g, v = 0.001 g^2, lr_t = lr sqrt(1 - 0.999) / (1 - 0.9); step = lr * g / |g| (eps -> 0)
    st = {"w": f(1.0), "w/Adam": f(0.0), "w/Adam_1": f(0.0)}
    O.opt_apply_dense(("Adam", 0.01, 0.9, 0.999, 0.0), st, "w", f(3.0), (0.9, 0.999))
    assert abs(float(st["w/Adam"]) - 0.3) < 1e-6 and abs(float(st["w/Adam_1"]) - 0.009) < 1e-7
    assert abs(float(st["w"]) - 0.99) < 1e-6
    # Adam sparse: row 0 hit at t = 1, not at t = 2 -> it still moves at t = 2 (m decays, no new gradient)
    st = {"t": torch.ones(2, 1), "t/Adam": torch.zeros(2, 1), "t/Adam_1": torch.zeros(2, 1)}
    O.opt_apply_rows(("Adam", 0.01, 0.9, 0.999, 1e-8), st, "t", np.asarray([0]), f(3.0).reshape(1, 1), (0.9, 0.999))
    assert abs(float(st["t"][0]) - 0.99) < 1e-6 and float(st["t"][1]) == 1.0       # m = v = 0: 0 / (0 + eps) = 0
    O.opt_apply_rows(("Adam", 0.01, 0.9, 0.999, 1e-8), st, "t", np.asarray([1]), f(-1.0).reshape(1, 1), (0.81, 0.998001))
    m0, v0 = 0.3 * 0.9, 0.009 * 0.999
    lr_t = 0.01 * np.sqrt(1 - 0.998001) / (1 - 0.81)
    assert abs(float(st["t"][0]) - (0.99 - lr_t * m0 / np.sqrt(v0))) < 1e-6
    assert abs(float(st["t"][1]) - (1.0 + lr_t * 0.1 / np.sqrt(0.001))) < 1e-6
    assert O.adam_pow_names(("Adam", 1, .9, .999, 1e-8), ("Adam", 1, .9, .999, 1e-8)) == {
        "dnn": ("beta1_power", "beta2_power"), "linear": ("beta1_power_1", "beta2_power_1")}
    assert O.adam_pow_names(("Adagrad", 1, .1), ("Adam", 1, .9, .999, 1e-8)) == {"linear": ("beta1_power", "beta2_power")}


def test_ftrl_learning_rate_power_and_centered_rmsprop_hand_computed():
    """FtrlCompute's general branch (training_ops.cc) and ApplyCenteredRMSProp on numbers small enough to follow by hand,
    plus the two equivalences TF's own ftrl_test.py checks: lr_power = -0.5 is the default branch, and lr_power = 0 without
    regularisation from zero weights is gradient descent (testEquivGradientDescentwithoutRegularization)."""
    import numpy as np
    import torch
    from oracle import oracle as O
    f = lambda *v: torch.tensor(v, dtype=torch.float32)
    # lr_power = -1: accum^1.  n: 0.1 -> 4.1; z = 0 + 2 - (4.1 - 0.1) / 0.5 * 1 = -6; quad = 4.1 / 0.5 + 2 * 0.25 = 8.7;
    # |z| > l1 = 1: w = (sign(z) * 1 - z) / quad = (-1 + 6) / 8.7
    st = {"w": f(1.0), "w/Ftrl_1": f(0.0), "w/Ftrl": f(0.1)}
    O.opt_apply_dense(("Ftrl", 0.5, 1.0, 0.25, 0.1, -1.0), st, "w", f(2.0), None)
    assert abs(float(st["w/Ftrl"]) - 4.1) < 1e-6 and abs(float(st["w/Ftrl_1"]) + 6.0) < 1e-5
    assert abs(float(st["w"]) - 5.0 / 8.7) < 1e-6
    # sparse form: only the listed row moves, same numbers
    st = {"t": torch.ones(2, 1), "t/Ftrl_1": torch.zeros(2, 1), "t/Ftrl": torch.full((2, 1), 0.1)}
    O.opt_apply_rows(("Ftrl", 0.5, 1.0, 0.25, 0.1, -1.0), st, "t", np.asarray([1]), f(2.0).reshape(1, 1), None)
    assert float(st["t"][0]) == 1.0 and float(st["t/Ftrl"][0]) == np.float32(0.1) and abs(float(st["t"][1]) - 5.0 / 8.7) < 1e-6
    # lr_power = -0.5 given explicitly through the general formula == the default branch
    g = torch.tensor([0.3, -1.2, 2.0, 0.0])
    a = {"w": f(0.5, -0.2, 0.0, 1.0), "w/Ftrl_1": f(0.1, 0.0, -0.3, 0.2), "w/Ftrl": f(0.1, 0.4, 0.1, 2.0)}
    wn, zn, nn = O._ftrl_general(a["w"], a["w/Ftrl_1"], a["w/Ftrl"], g, 0.1, 0.05, 0.5, -0.5)
    O.opt_apply_dense(("Ftrl", 0.1, 0.05, 0.5, 0.1), a, "w", g, None)
    assert torch.allclose(wn, a["w"], rtol=1e-6, atol=1e-7) and torch.allclose(zn, a["w/Ftrl_1"], rtol=1e-6, atol=1e-7)
    assert torch.allclose(nn, a["w/Ftrl"])
    # lr_power = 0, l1 = l2 = 0, w0 = 0: three Ftrl steps == three SGD steps
    st = {"w": f(0.0, 0.0), "w/Ftrl_1": f(0.0, 0.0), "w/Ftrl": f(0.1, 0.1)}
    sg = {"w": f(0.0, 0.0)}
    for gi in (f(0.1, 0.2), f(-0.3, 0.05), f(0.02, -0.04)):
        O.opt_apply_dense(("Ftrl", 3.0, 0.0, 0.0, 0.1, 0.0), st, "w", gi, None)
        O.opt_apply_dense(("SGD", 3.0), sg, "w", gi, None)
    assert torch.allclose(st["w"], sg["w"], rtol=1e-6, atol=1e-7)
    # centered RMSProp, decay 0.9: ms = 1 + (4 - 1) * 0.1 = 1.3; mg = 0 + (2 - 0) * 0.1 = 0.2;
    # mom = 0.5 * 0 + 0.1 * 2 / sqrt(1.3 - 0.04 + 0) ; var = 1 - mom
    opt = ("RMSProp", 0.1, 0.9, 0.5, 0.0, True)
    assert O.slot_names(opt) == ("/RMSProp", "/RMSProp_2", "/RMSProp_1")     # rms, momentum, mg in TF's creation order
    st = {"w": f(1.0), "w/RMSProp": f(1.0), "w/RMSProp_1": f(0.0), "w/RMSProp_2": f(0.0)}
    O.opt_apply_dense(opt, st, "w", f(2.0), None)
    mom = 0.1 * 2 / np.sqrt(1.26)
    assert abs(float(st["w/RMSProp"]) - 1.3) < 1e-6 and abs(float(st["w/RMSProp_1"]) - 0.2) < 1e-7
    assert abs(float(st["w/RMSProp_2"]) - mom) < 1e-6 and abs(float(st["w"]) - (1 - mom)) < 1e-6
    st = {"t": torch.ones(2, 1), "t/RMSProp": torch.ones(2, 1), "t/RMSProp_1": torch.zeros(2, 1), "t/RMSProp_2": torch.zeros(2, 1)}
    O.opt_apply_rows(opt, st, "t", np.asarray([1]), f(2.0).reshape(1, 1), None)
    assert float(st["t"][0]) == 1.0 and float(st["t/RMSProp_1"][0]) == 0.0 and abs(float(st["t"][1]) - (1 - mom)) < 1e-6
    assert abs(float(st["t/RMSProp_1"][1]) - 0.2) < 1e-7


def test_crelu_equals_a_relu_layer_of_twice_the_width_with_tied_halves():
    """What the engine builds for activation `crelu` (plan.FeaturePlan): relu(x [W | -W] + [b | -b]) IS
    tf.nn.crelu(x W + b) = concat(relu(z), relu(-z)) bit for bit, and dL/dW = G'[:, :N] - G'[:, N:]."""
    import torch
    from oracle import oracle as O
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, 19, generator=g)
    W = torch.randn(19, 8, generator=g).requires_grad_(True)
    b = torch.randn(8, generator=g).requires_grad_(True)
    V = torch.randn(16, 3, generator=g)
    y = O.ACTIVATIONS["crelu"](x @ W + b)
    assert y.shape == (37, 16)
    (y @ V).sum().backward()
    W2 = torch.cat([W.detach(), -W.detach()], dim=1).requires_grad_(True)
    b2 = torch.cat([b.detach(), -b.detach()]).requires_grad_(True)
    y2 = torch.relu(x @ W2 + b2)
    assert torch.equal(y.detach(), y2.detach())
    (y2 @ V).sum().backward()
    assert torch.allclose(W.grad, W2.grad[:, :8] - W2.grad[:, 8:], rtol=1e-5, atol=1e-6)
    assert torch.allclose(b.grad, b2.grad[:8] - b2.grad[8:], rtol=1e-5, atol=1e-6)
    # tower_forward: BN (2N features) and the next kernel (2N rows) follow the doubled width
    tw = {"kernel": [W.detach(), torch.randn(16, 4, generator=g)], "bias": [b.detach(), torch.zeros(4)],
          "gamma": [torch.ones(16), torch.ones(8)], "beta": [torch.zeros(16), torch.zeros(8)],
          "logits_kernel": torch.randn(8, 1, generator=g), "logits_bias": torch.zeros(1)}
    assert O.tower_forward(x, tw, "simple", "crelu", True).shape == (37, 1)


def test_batched_column_calls_equal_per_column_calls_bit_for_bit():
    """OracleWideDeep.batched (bench.py's cpu_baseline configuration: all sparse columns of a step through one C call each
    way, columns side by side under OpenMP) must be the SAME arithmetic as the per-column path the parity tests use."""
    import torch
    rng_b = 300
    deep = [{"name": "c%d_embedding" % i, "kind": "embedding", "key": "c%d" % i, "num_buckets": 50 + i, "dim": 8} for i in range(4)]
    deep.append({"name": "x0", "kind": "numeric", "key": "x0", "num_buckets": 0, "dim": 1})
    wide = [{"name": "c%d" % i, "key": "c%d" % i, "num_buckets": 50 + i} for i in range(4)]

    def mk():
        st, g = {}, torch.Generator().manual_seed(1)
        for c in deep[:4]:
            n = O.OracleWideDeep.emb_name(c)
            st[n] = torch.randn(c["num_buckets"], 8, generator=g)
            st[n + "/Adagrad"] = torch.full((c["num_buckets"], 8), 0.1)
        for c in wide:
            n = O.OracleWideDeep.wide_name(c)
            st[n] = torch.zeros(c["num_buckets"], 1)
            st[n + "/Ftrl"] = torch.full((c["num_buckets"], 1), 0.1)
            st[n + "/Ftrl_1"] = torch.zeros(c["num_buckets"], 1)
        b = "linear/linear_model/bias_weights"
        st[b], st[b + "/Ftrl"], st[b + "/Ftrl_1"] = torch.zeros(1), torch.full((1,), 0.1), torch.zeros(1)
        p = "dnn/dnn_1/"
        for l, (k, n) in enumerate([(33, 16), (16, 8)]):
            st[p + "hiddenlayer_%d/kernel" % l] = torch.randn(k, n, generator=g) * 0.2
            st[p + "hiddenlayer_%d/bias" % l] = torch.zeros(n)
            st[p + "hiddenlayer_%d/batch_normalization/gamma" % l] = torch.ones(n)
            st[p + "hiddenlayer_%d/batch_normalization/beta" % l] = torch.zeros(n)
        st[p + "logits/kernel"], st[p + "logits/bias"] = torch.randn(8, 1, generator=g) * 0.2, torch.zeros(1)
        for k in list(st):
            if k.startswith(p):
                st[k + "/Adagrad"] = torch.full_like(st[k], 0.1)
        return st

    def batch(seed):
        r = np.random.default_rng(seed)
        ids = {}
        for i in range(4):
            lens = r.integers(0, 4, size=rng_b)
            offs = np.zeros(rng_b + 1, np.int32)
            offs[1:] = np.cumsum(lens)
            ids["c%d" % i] = (r.integers(-1, 50 + i, size=int(offs[-1])).astype(np.int64), offs)   # -1: pruned ids, empty bags
        return {"ids": ids, "dense": {"x0": r.standard_normal(rng_b).astype(np.float32)},
                "labels": (r.random(rng_b) < 0.3).astype(np.float32), "weights": None}

    a = O.OracleWideDeep("wide_deep", deep, wide, [([16, 8], "simple")], mk())
    b = O.OracleWideDeep("wide_deep", deep, wide, [([16, 8], "simple")], mk())
    b.batched = True
    for s in range(3):
        bt = batch(s)
        la, lga = a.train_step(bt)
        lb, lgb = b.train_step(bt)
        assert la == lb and torch.equal(lga, lgb)
    for k in a.state:
        assert torch.equal(a.state[k], b.state[k]), k
