"""CPU: oracle.tower_forward (the restatement of python/lib/dnn.py every GPU tower test is checked against) versus dnn.py
ITSELF: tests/golden/ref_dnn_graphs.json holds the dataflow graphs recorded while executing the reference's `_dnn_logit_fn`
/ multi-DNN builder against a stub tensorflow (tests/golden/make_ref_dnn_golden.py).  The graphs are interpreted here in
numpy with seeded weights; logits must equal the oracle's -- concat order of every connected mode, dropout between the
activation and BN in TRAIN mode only, BN called without `training` (=> inference affine, quirk C.1), logits layer linear,
towers of a multi-DNN added."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_dnn_graphs.json")))
B, K0 = 7, 6


def _interpret(graph, x, seed, dropout_masks):
    """returns (output, tower weights in oracle layout per model id)"""
    rng = np.random.default_rng(seed)
    vals, towers, n_drop = [], {}, 0
    inv = np.float32(1.0) / np.sqrt(np.float32(1.0) + np.float32(1e-3))
    for node in graph:
        op, ins, at = node["op"], [vals[i] for i in node["inputs"]], node["attrs"]
        if op == "input_layer":
            v = x
        elif op == "dense":
            tid, lname = at["scope"].split("/")
            tw = towers.setdefault(tid, {"kernel": [], "bias": [], "gamma": [], "beta": [], "pending_bn": None})
            W = rng.standard_normal((ins[0].shape[1], at["units"])).astype(np.float32) * 0.5
            b = rng.standard_normal(at["units"]).astype(np.float32) * 0.1
            assert at["use_bias"] is True
            v = ins[0] @ W + b
            if lname == "logits":
                assert at["activation"] is None and at["units"] == 1
                tw["logits_kernel"], tw["logits_bias"] = W, b
            else:
                assert lname == "hiddenlayer_%d" % len(tw["kernel"]) and at["activation"] == "tf.nn.relu"
                v = np.maximum(v, 0)
                tw["kernel"].append(W)
                tw["bias"].append(b)
                tw["last"] = tid
            node["_tid"] = tid
        elif op == "dropout":
            assert at["training"] is True
            m = dropout_masks[n_drop]
            n_drop += 1
            v = ins[0] / np.float32(1.0 - at["rate"]) * m[:, : ins[0].shape[1]]
        elif op == "batch_normalization":
            assert at == {}                      # no `training=True`: moving statistics (0, 1) -> an affine (quirk C.1)
            n = ins[0].shape[1]
            gamma = (1.0 + 0.3 * rng.standard_normal(n)).astype(np.float32)
            beta = (0.2 * rng.standard_normal(n)).astype(np.float32)
            tid = [t for t, tw in towers.items() if len(tw["gamma"]) < len(tw["kernel"])][0]
            towers[tid]["gamma"].append(gamma)
            towers[tid]["beta"].append(beta)
            v = ins[0] * (gamma * inv) + beta
        elif op == "concat":
            assert at["axis"] == 1
            v = np.concatenate(ins, axis=1)
        elif op == "add_n":
            v = sum(ins)
        else:
            raise AssertionError("unexpected op " + op)
        vals.append(v)
    return vals[-1], towers, n_drop


def _to_torch(tw):
    t = lambda a: torch.as_tensor(a)
    return {"kernel": [t(a) for a in tw["kernel"]], "bias": [t(a) for a in tw["bias"]], "gamma": [t(a) for a in tw["gamma"]],
            "beta": [t(a) for a in tw["beta"]], "logits_kernel": t(tw["logits_kernel"]), "logits_bias": t(tw["logits_bias"])}


CASES = [c for c in G["cases"] if "graph" in c]


@pytest.mark.parametrize("k", range(len(CASES)), ids=lambda k: "%s-%s-do%s-bn%d-%s" % (
    CASES[k]["connected_mode"], "x".join(map(str, CASES[k]["hidden_units"])), CASES[k]["dropout"], CASES[k]["batch_norm"], CASES[k]["mode"]))
def test_oracle_tower_equals_the_recorded_reference_graph(k):
    c = CASES[k]
    rng = np.random.default_rng(k)
    x = rng.standard_normal((B, K0)).astype(np.float32)
    masks = [(rng.random((B, 16)) < 0.75).astype(np.float32) for _ in c["hidden_units"]]
    out, towers, n_drop = _interpret(c["graph"], x, 100 + k, masks)
    train = c["mode"] == "train"
    assert n_drop == (len(c["hidden_units"]) if (c["dropout"] is not None and train) else 0)     # dropout: TRAIN mode only
    tw = towers["dnn_1"]
    assert len(tw["kernel"]) == len(c["hidden_units"]) and [w.shape[1] for w in tw["kernel"]] == c["hidden_units"]
    assert len(tw["gamma"]) == (len(c["hidden_units"]) if c["batch_norm"] else 0)
    use_do = c["dropout"] if (c["dropout"] is not None and train) else None
    got = O.tower_forward(torch.as_tensor(x), _to_torch(tw), c["connected_mode"], "relu", c["batch_norm"], use_do,
                          [m[:, :h] for m, h in zip(masks, c["hidden_units"])] if use_do else None)
    np.testing.assert_allclose(got.numpy(), out, rtol=2e-5, atol=2e-5)


def test_multi_dnn_adds_the_tower_logits_and_shares_the_input_layer():
    m = G["multi"]
    x = np.random.default_rng(5).standard_normal((B, K0)).astype(np.float32)
    out, towers, _ = _interpret(m["graph"], x, 77, [])
    assert sorted(towers) == ["dnn_1", "dnn_2"]                       # variable scopes dnn_<model id>/hiddenlayer_<l>, /logits
    assert sum(1 for n in m["graph"] if n["op"] == "input_layer") == 2  # built per tower under AUTO_REUSE: same variables
    tot = sum(O.tower_forward(torch.as_tensor(x), _to_torch(towers["dnn_%d" % (i + 1)]), mode, "relu", True)
              for i, mode in enumerate(m["connected_mode"]))
    np.testing.assert_allclose(tot.numpy(), out, rtol=2e-5, atol=2e-5)


def test_reference_rejects_last_dense_as_shipped():
    """dnn.py:77 spells the accepted name 'lase_dense': `last_dense` fails the assert, and the misspelt name falls into the
    arbitrary-connections branch and crashes.  This engine builds last_dense as documented (dnn.py:55, 135-153)."""
    bad = [c for c in G["cases"] if "exception" in c]
    assert {c["connected_mode"] for c in bad} == {"last_dense", "lase_dense"}
    for c in bad:
        if c["connected_mode"] == "last_dense":
            assert c["exception"] == {"class": "AssertionError", "message": "Invalid connected_mode: last_dense"}
    assert G["shipped"]["activation"] == "tf.nn.relu"
