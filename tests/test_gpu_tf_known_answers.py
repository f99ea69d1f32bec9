"""GPU: the HIP kernels, through the C ABI, against known answers printed in upstream TensorFlow's own unit tests
(tests/golden/kat_tf_fp32.json; the same vectors pin the CPU oracle in tests/test_tf_known_answers.py)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_tf_fp32.json")))


def _opt_dense(opt, var, grad, steps, each_step=None):
    from wide_deep_amd import capi
    from wide_deep_amd.capi import call, ptr
    from wide_deep_amd.plan import opt_params, opt_slot_init
    import ctypes
    o = capi.WdOpt()
    o.kind, o.lr = capi.WD_OPT_KINDS[opt[0]], float(opt[1])
    o.p0, o.p1, o.p2, o.p3 = opt_params(opt)
    ia, ib = opt_slot_init(opt)
    w = torch.tensor(var, dtype=torch.float32, device="cuda")
    a = torch.full_like(w, 0.0 if ia is None else ia)
    b = torch.full_like(w, 0.0 if ib is None else ib)
    g = torch.tensor(grad, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    if opt[0] == "Adam":
        pw = torch.tensor([opt[2], opt[3]], dtype=torch.float32, device="cuda")     # beta^t of the step being applied
        o.pow = pw.data_ptr()
    for t in range(steps):
        call("wd_opt_dense", ptr(w), ptr(a), ptr(b), ptr(g), w.numel(), ctypes.byref(o), st)
        if opt[0] == "Adam":
            call("wd_adam_tick", ptr(pw), float(opt[2]), float(opt[3]), st)
        if each_step is not None:
            each_step(t + 1, w.cpu().numpy())
    return w.cpu().numpy()


def test_dense_adagrad_and_ftrl_kernels_match_tf_optimizer_tests():
    a = G["adagrad"]
    for c in a["cases"]:
        np.testing.assert_allclose(_opt_dense(("Adagrad", a["lr"], a["init"]), c["var"], c["grad"], a["steps"]), c["expect"], rtol=3e-6)
    for f in G["ftrl"]:
        for c in f["cases"]:
            opt = ("Ftrl", f["lr"], f["l1"], f["l2"], f["init"]) + ((-0.5, f["l2_shrinkage"]) if "l2_shrinkage" in f else ())
            got = _opt_dense(opt, c["var"], c["grad"], f["steps"])
            np.testing.assert_allclose(got, c["expect"], rtol=3e-6)


def test_head_kernel_matches_tf_head_test():
    from wide_deep_amd.capi import call, ptr
    for c in G["head_loss"]["cases"]:
        n = len(c["logits"])
        x = torch.tensor(c["logits"], device="cuda")
        zero = torch.zeros(n, device="cuda")
        y = torch.tensor(c["labels"], device="cuda")
        w = torch.tensor(c["weights"], device="cuda") if "weights" in c else None
        logit = torch.zeros(n, device="cuda"); p = torch.zeros(n, device="cuda"); dl = torch.zeros(n, device="cuda")
        loss = torch.zeros(1, device="cuda")
        call("wd_bce_sum_fwd_bwd", ptr(x), ptr(zero), ptr(y), ptr(w) if w is not None else None, n, ptr(logit), ptr(p), ptr(dl),
             ptr(loss), torch.cuda.current_stream().cuda_stream)
        assert abs(float(loss) - c["expect"]) < 1e-4


def test_engine_mean_combiner_and_wide_sum_match_tf_feature_column_tests():
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.plan import criteo_spec
    # embedding_column, combiner='mean', an empty bag in the middle (table padded from 2 to 4 columns with zeros)
    e = G["embedding_mean"]
    eng = WideDeepEngine(criteo_spec(n_dense=1, n_sparse=1, buckets=3, dim=4, hidden=(8,)), max_batch=64)
    st = eng.export_state()
    nm = "dnn/input_from_feature_columns/input_layer/%s/embedding_weights" % eng.plan.slots[0].deep_name
    tab = torch.zeros(3, 4)
    tab[:, :2] = torch.tensor(e["table"])
    st[nm] = tab
    eng.import_state(st)
    lens = np.diff(np.asarray(e["offs"])).reshape(-1, 1).astype(np.int64)
    B = len(lens)
    hb = {"B": B, "lens": lens, "raw": np.asarray(e["ids"], np.int64), "dense": np.zeros((B, 1), np.float32),
          "labels": np.zeros(B, np.float32)}
    eng.forward(synth.to_device_ids(eng.plan, hb))
    torch.cuda.synchronize()
    c0 = eng.plan.out_col[0]
    x = eng.towers[0]["act"][:B, c0:c0 + 4].cpu()
    assert x[:, :2].tolist() == e["expect"] and float(x[:, 2:].abs().max()) == 0.0
    # linear_model, sparse_combiner='sum', + bias
    l = G["linear_cross_sum"]
    eng = WideDeepEngine(criteo_spec(n_dense=0, n_sparse=1, buckets=5, dim=4, hidden=(8,), model_type="wide"), max_batch=64)
    st = eng.export_state()
    st["linear/linear_model/%s/weights" % eng.plan.slots[0].name] = torch.tensor(l["weights"]).reshape(-1, 1)
    st["linear/linear_model/bias_weights"] = torch.tensor([l["bias"]])
    eng.import_state(st)
    lens = np.diff(np.asarray(l["offs"])).reshape(-1, 1).astype(np.int64)
    B = len(lens)
    hb = {"B": B, "lens": lens, "raw": np.asarray(l["ids"], np.int64), "dense": None, "labels": np.zeros(B, np.float32)}
    logit = eng.forward(synth.to_device_ids(eng.plan, hb))
    np.testing.assert_allclose(logit.cpu().numpy(), l["expect"], rtol=1e-6)


def test_dense_sgd_rmsprop_adam_kernels_match_tf_optimizer_tests():
    s, r, a = G["sgd"], G["rmsprop"], G["adam"]
    for c in s["cases"]:
        np.testing.assert_allclose(_opt_dense(("SGD", s["lr"]), c["var"], c["grad"], s["steps"]), c["expect"], rtol=1e-6)
    for c in r["cases"]:
        got = _opt_dense(("RMSProp", r["lr"], r["decay"], r["momentum"], r["epsilon"]), c["var"], c["grad"], r["steps"])
        np.testing.assert_allclose(got, c["expect"], atol=1e-5)
    for c in a["cases"]:
        ref = {"p": np.asarray(c["var"], np.float64), "m": 0.0, "v": 0.0}
        g = np.asarray(c["grad"], np.float64)

        def check(t, w):          # adam_update_numpy of TF's adam_test.py, step t
            alpha_t = a["lr"] * np.sqrt(1 - a["beta2"] ** t) / (1 - a["beta1"] ** t)
            ref["m"] = a["beta1"] * ref["m"] + (1 - a["beta1"]) * g
            ref["v"] = a["beta2"] * ref["v"] + (1 - a["beta2"]) * g * g
            ref["p"] = ref["p"] - alpha_t * ref["m"] / (np.sqrt(ref["v"]) + a["epsilon"])
            np.testing.assert_allclose(w, ref["p"], rtol=1e-6)
        _opt_dense(("Adam", a["lr"], a["beta1"], a["beta2"], a["epsilon"]), c["var"], c["grad"], a["steps"], check)
