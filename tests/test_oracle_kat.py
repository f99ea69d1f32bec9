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
