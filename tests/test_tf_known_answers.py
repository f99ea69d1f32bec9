"""CPU: the fp32 semantics of the path (optimizer formulas, mean / sum combiners, batch-SUM head loss, streaming AUC) against
known answers printed in upstream TensorFlow's own unit tests (tests/golden/kat_tf_fp32.json names each TF test).
TensorFlow is the reference's arithmetic dependency and cannot run here; these constants are what it asserts about itself."""
import json
import os

import numpy as np
import torch

from oracle import oracle as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_tf_fp32.json")))


def _run(opt, var, grad, steps, rows):
    st = {"v": torch.tensor(var, dtype=torch.float32)}
    a, b = O.slot_init_values(opt)
    sa, sb = O.SLOT_NAMES[opt[0]]
    if sa:
        st["v" + sa] = torch.full_like(st["v"], a)
    if sb:
        st["v" + sb] = torch.full_like(st["v"], b)
    g = torch.tensor(grad, dtype=torch.float32)
    for _ in range(steps):
        if rows:      # the sparse apply of the same optimizer: every element its own row
            O.opt_apply_rows(opt, st, "v", np.arange(len(var)), g.reshape(-1, 1), None)
        else:
            O.opt_apply_dense(opt, st, "v", g, None)
    return st["v"].numpy()


def test_adagrad_and_ftrl_formulas_match_tf_optimizer_tests():
    a = G["adagrad"]
    for rows in (False, True):
        for c in a["cases"]:
            got = _run(("Adagrad", a["lr"], a["init"]), c["var"], c["grad"], a["steps"], rows)
            np.testing.assert_allclose(got, c["expect"], rtol=2e-6)
        for f in G["ftrl"]:
            for c in f["cases"]:
                got = _run(("Ftrl", f["lr"], f["l1"], f["l2"], f["init"]), c["var"], c["grad"], f["steps"], rows)
                np.testing.assert_allclose(got, c["expect"], rtol=2e-6)


def test_embedding_mean_linear_sum_and_head_loss_match_tf_tests():
    e = G["embedding_mean"]
    out = O.embag_fwd(torch.tensor(e["table"]), np.asarray(e["ids"], np.int64), np.asarray(e["offs"], np.int32), mean=True)
    assert out.tolist() == e["expect"]                          # empty bag -> zero vector
    l = G["linear_cross_sum"]
    w = torch.tensor(l["weights"]).reshape(-1, 1)
    out = O.embag_fwd(w, np.asarray(l["ids"], np.int64), np.asarray(l["offs"], np.int32), mean=False).reshape(-1) + l["bias"]
    np.testing.assert_allclose(out.numpy(), l["expect"], rtol=1e-6)
    for c in G["head_loss"]["cases"]:
        w = torch.tensor(c["weights"]) if "weights" in c else None
        loss, dl, p = O.bce_sum(torch.tensor(c["logits"]), torch.tensor(c["labels"]), w)
        assert abs(loss - c["expect"]) < 1e-4
        ww = np.asarray(c.get("weights", [1.0] * len(c["logits"])))
        np.testing.assert_allclose(dl.numpy(), ww * (p.numpy() - np.asarray(c["labels"])), atol=1e-6)   # not divided by B


def test_streaming_auc_matches_tf_metrics_tests():
    from wide_deep_amd.estimator import binary_head_metrics
    for c in G["auc"]["cases"]:
        p = np.asarray(c["pred"], np.float64)
        y = np.asarray(c["label"], np.float64)
        w = np.asarray(c.get("weight", np.ones_like(p)), np.float64)
        pc = np.clip(p, 1e-9, 1 - 1e-9)
        m = binary_head_metrics(p, np.log(pc / (1 - pc)), y, w, 0, len(p))
        got = m["auc"] if c["curve"] == "ROC" else m["auc_precision_recall"]
        assert abs(got - c["expect"]) < 1e-3, c["name"]       # TF's own delta
