"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (the checker, never the product).

CPU restatement of the reference's train-step hot path.  The integer / sparse
pieces call the plain-C restatement in ``wd_oracle.c``; the dense tower is a
plain fp32 PyTorch-CPU autograd graph (so the hand-written HIP backward is
checked against an independently derived gradient).

What it follows in the reference (Lapis-Hong/wide_deep):
  * column wiring             python/lib/build_estimator.py:49-169
  * wide logits               python/lib/linear.py:20-36
  * deep input layer + tower  python/lib/dnn.py:43-275
  * combine + head + train op python/lib/joint.py:81-269
  * optimizers/activations    python/lib/utils/model_util.py:28-105
and the TensorFlow-1.x semantics those calls select,
including the quirks of Appendix C (BN always inference-mode affine, batch-SUM
loss, no lr decay, regularizers never added to the loss).

Only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and
``__graft_entry__.smoke()`` may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libwd_oracle.so")
_lib = None

BN_EPS = 1e-3  # tf.layers.batch_normalization default epsilon


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["sh", os.path.join(_HERE, "build.sh")])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.wdo_fingerprint64.restype = ctypes.c_uint64
        L.wdo_fingerprint64.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.wdo_fingerprint_cat64.restype = ctypes.c_uint64
        L.wdo_fingerprint_cat64.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
        L.wdo_embag_row_grads.restype = ctypes.c_int64
        L.wdo_bce_sum.restype = ctypes.c_double
        L.wdo_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


class _Ptr(ctypes.c_void_p):
    """address of an array / tensor that also OWNS a reference to it: `_p(x.contiguous())` or `_p(np.ascontiguousarray(x))`
    may be handed a fresh copy that nothing else refers to -- it has to live until the C call returns."""


def _p(a):
    if a is None:
        return None
    p = _Ptr(a.data_ptr() if isinstance(a, torch.Tensor) else a.ctypes.data)
    p.keep = a
    return p


# ---------------------------------------------------------------------------
# integer path
# ---------------------------------------------------------------------------
def fingerprint64(s: bytes) -> int:
    return int(lib().wdo_fingerprint64(s, len(s)))


def fingerprint_cat64(a: int, b: int) -> int:
    return int(lib().wdo_fingerprint_cat64(ctypes.c_uint64(a), ctypes.c_uint64(b)))


def pack_tokens(tokens):
    """list[bytes|str] -> (uint8 bytes, int64 offs[n+1])"""
    bs = [t.encode() if isinstance(t, str) else t for t in tokens]
    offs = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs])
    data = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return data, offs


def fingerprint64_batch(data, offs):
    n = len(offs) - 1
    out = np.zeros(n, dtype=np.uint64)
    data = np.ascontiguousarray(data)
    if data.size == 0:
        data = np.zeros(1, np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.int64)      # named: the converted copy must outlive the call
    lib().wdo_fingerprint64_batch(_p(data), _p(offs), ctypes.c_int64(n), _p(out))
    return out


def hash_bucket(tokens, num_buckets):
    """string_to_hash_bucket_fast over a list of tokens -> int64 ids."""
    data, offs = pack_tokens(tokens)
    fp = fingerprint64_batch(data, offs)
    return (fp % np.uint64(num_buckets)).astype(np.int64)


def cross_hash(cols, num_buckets, hash_key=0xDECAFCAFFE):
    """SparseCross (hashed).  cols: list of (vals uint64[nnz_k], offs int32[B+1]).
    String keys pass Fingerprint64(token); int keys pass the id itself.
    Returns (ids int64, offs int32[B+1]); last key varies fastest."""
    nk = len(cols)
    assert 1 <= nk <= 16
    B = len(cols[0][1]) - 1
    vals = [np.ascontiguousarray(np.asarray(c[0]).astype(np.uint64)) for c in cols]
    vals = [v if v.size else np.zeros(1, np.uint64) for v in vals]
    offs = [np.ascontiguousarray(c[1], dtype=np.int32) for c in cols]
    VP = (ctypes.c_void_p * nk)(*[v.ctypes.data for v in vals])
    OP = (ctypes.c_void_p * nk)(*[o.ctypes.data for o in offs])
    out_offs = np.zeros(B + 1, dtype=np.int32)
    lib().wdo_cross_offsets(OP, ctypes.c_int(nk), ctypes.c_int64(B), _p(out_offs))
    out = np.zeros(max(int(out_offs[-1]), 1), dtype=np.int64)
    lib().wdo_cross_hash(VP, OP, ctypes.c_int(nk), ctypes.c_int64(B), ctypes.c_uint64(hash_key),
                         ctypes.c_uint64(int(num_buckets)), _p(out_offs), _p(out))
    return out[: int(out_offs[-1])], out_offs


def bucketize(x, boundaries):
    """tf bucketized_column: number of boundaries <= x."""
    return np.searchsorted(np.asarray(boundaries, dtype=np.float32), np.asarray(x, dtype=np.float32), side="right").astype(np.int64)


# ---------------------------------------------------------------------------
# sparse float pieces (one column at a time, like TF)
# ---------------------------------------------------------------------------
def embag_fwd(table, ids, offs, mean):
    V, D = table.shape
    B = len(offs) - 1
    out = torch.zeros(B, D, dtype=torch.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    if ids.size == 0:
        ids = np.zeros(1, np.int64)
    lib().wdo_embag_fwd(_p(table), ctypes.c_int64(D), _p(ids), _p(np.ascontiguousarray(offs, dtype=np.int32)),
                        ctypes.c_int64(B), ctypes.c_int(1 if mean else 0), _p(out), ctypes.c_int64(D))
    return out


def embag_row_grads(D, ids, offs, grad_out, mean):
    B = len(offs) - 1
    nnz = int(offs[-1])
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    if ids.size == 0:
        ids = np.zeros(1, np.int64)
    uniq = np.zeros(max(nnz, 1), dtype=np.int64)
    rg = torch.zeros(max(nnz, 1), D, dtype=torch.float32)
    g = grad_out.contiguous()
    n = lib().wdo_embag_row_grads(ctypes.c_int64(D), _p(ids), _p(np.ascontiguousarray(offs, dtype=np.int32)),
                                  ctypes.c_int64(B), ctypes.c_int(1 if mean else 0), _p(g), ctypes.c_int64(D),
                                  _p(uniq), _p(rg))
    return uniq[:n], rg[:n]


def adagrad_rows(table, accum, uniq, row_grad, lr):
    D = table.shape[1]
    lib().wdo_adagrad_rows(_p(table), _p(accum), ctypes.c_int64(D), _p(np.ascontiguousarray(uniq)),
                           ctypes.c_int64(len(uniq)), _p(row_grad.contiguous()), ctypes.c_float(lr))


def adagrad_dense(w, accum, g, lr):
    assert w.is_contiguous() and accum.is_contiguous()
    lib().wdo_adagrad_dense(_p(w), _p(accum), _p(g.contiguous()), ctypes.c_int64(w.numel()), ctypes.c_float(lr))


def ftrl_rows(w, z, n, uniq, row_grad, lr, l1, l2):
    D = w.shape[1] if w.dim() > 1 else 1
    lib().wdo_ftrl_rows(_p(w), _p(z), _p(n), ctypes.c_int64(D), _p(np.ascontiguousarray(uniq)),
                        ctypes.c_int64(len(uniq)), _p(row_grad.contiguous()), ctypes.c_float(lr), ctypes.c_float(l1),
                        ctypes.c_float(l2))


def ftrl_dense(w, z, n, g, lr, l1, l2):
    lib().wdo_ftrl_dense(_p(w), _p(z), _p(n), _p(g.contiguous()), ctypes.c_int64(w.numel()), ctypes.c_float(lr),
                         ctypes.c_float(l1), ctypes.c_float(l2))


def _ptr_array(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*[p if isinstance(p, int) else (p.value if p is not None else None) for p in ptrs])


def embag_fwd_cols(tables, ids_offs, B, mean, outs, ld_outs):
    """wdo_embag_fwd for ALL columns in one call, columns side by side (bit-identical to the per-column calls).
    tables: list of [V, D] tensors; ids_offs: list of (ids int64, offs int32); outs: list of data pointers (int)."""
    n = len(tables)
    if n == 0:
        return
    keep = []
    idp, ofp = [], []
    for ids, offs in ids_offs:
        i = np.ascontiguousarray(ids, dtype=np.int64)
        if i.size == 0:
            i = np.zeros(1, np.int64)
        o = np.ascontiguousarray(offs, dtype=np.int32)
        keep += [i, o]
        idp.append(i.ctypes.data)
        ofp.append(o.ctypes.data)
    D = np.asarray([t.shape[1] if t.dim() > 1 else 1 for t in tables], dtype=np.int64)
    ld = np.asarray(ld_outs, dtype=np.int64)
    lib().wdo_embag_fwd_cols(ctypes.c_int(n), _ptr_array([t.data_ptr() for t in tables]), _p(D), _ptr_array(idp),
                             _ptr_array(ofp), ctypes.c_int64(B), ctypes.c_int(1 if mean else 0), _ptr_array(list(outs)),
                             _p(ld))


def sparse_apply_cols(tables, slot_a, slot_b, ids_offs, B, mean, grad_ptrs, ld_grads, kind, lr, l1=0.0, l2=0.0):
    """Row gradients (duplicates summed) + Adagrad (kind 0) / Ftrl (kind 1) sparse apply for ALL columns in one call."""
    n = len(tables)
    if n == 0:
        return
    keep, idp, ofp = [], [], []
    for ids, offs in ids_offs:
        i = np.ascontiguousarray(ids, dtype=np.int64)
        if i.size == 0:
            i = np.zeros(1, np.int64)
        o = np.ascontiguousarray(offs, dtype=np.int32)
        keep += [i, o]
        idp.append(i.ctypes.data)
        ofp.append(o.ctypes.data)
    D = np.asarray([t.shape[1] if t.dim() > 1 else 1 for t in tables], dtype=np.int64)
    ld = np.asarray(ld_grads, dtype=np.int64)
    pa = _ptr_array([t.data_ptr() if t is not None else None for t in slot_a])
    lib().wdo_sparse_apply_cols(ctypes.c_int(n), _p(D), _ptr_array(idp), _ptr_array(ofp), ctypes.c_int64(B),
                                ctypes.c_int(1 if mean else 0), _ptr_array(list(grad_ptrs)), _p(ld),
                                _ptr_array([t.data_ptr() for t in tables]), pa,
                                _ptr_array([t.data_ptr() for t in slot_b]), ctypes.c_int(kind), ctypes.c_float(lr),
                                ctypes.c_float(l1), ctypes.c_float(l2))


# ---------------------------------------------------------------------------
# the other optimizers the reference accepts (python/lib/utils/model_util.py:84-90), restated from the TF 1.x kernels:
#   GradientDescent  training_ops ApplyGradientDescent / scatter_sub          var -= lr g
#   RMSProp          ApplyRMSProp / SparseApplyRMSProp (touched rows only)    ms += (g^2 - ms)(1 - decay);
#                    mom = mom * momentum + lr g / sqrt(ms + eps); var -= mom     (slots: rms init 1, momentum init 0)
#   Adam             dense: ApplyAdam  m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); var -= lr_t m / (sqrt(v) + eps)
#                    sparse: AdamOptimizer._apply_sparse_shared  m = m b1 (WHOLE variable); m[rows] += (1 - b1) g; same for v;
#                    var -= lr_t m / (sqrt(v) + eps) (WHOLE variable);  lr_t = lr sqrt(1 - b2^t) / (1 - b1^t)
# No golden vectors exist for SGD / RMSProp / Adam (parity unpinned: SURVEY 8c); they are checked against hand-computed
# cases in tests/test_oracle_kat.py.  Adagrad and Ftrl ARE pinned by the constants of TF's own optimizer tests
# (tests/golden/kat_tf_fp32.json, tests/test_tf_known_answers.py).
#   Ftrl, learning_rate_power != -0.5   FtrlCompute's general branch: pow(accum, -lr_power) in place of sqrt(accum)
#   RMSProp centered  ApplyCenteredRMSProp / SparseApplyCenteredRMSProp: mg += (g - mg)(1 - decay);
#                    mom = mom * momentum + lr g / sqrt(ms - mg^2 + eps)   (slot mg init 0; TF names the slots in creation
#                    order rms, mg, momentum = /RMSProp, /RMSProp_1, /RMSProp_2)
# Optimizer tuples: ("SGD", lr) ("Adagrad", lr, init) ("Ftrl", lr, l1, l2, init[, lr_power])
#                   ("RMSProp", lr, decay, momentum, eps[, centered]) ("Adam", lr, beta1, beta2, eps)
# ---------------------------------------------------------------------------
SLOT_NAMES = {"SGD": (None, None), "Adagrad": (None, "/Adagrad"), "Ftrl": ("/Ftrl_1", "/Ftrl"),
              "RMSProp": ("/RMSProp", "/RMSProp_1"), "Adam": ("/Adam", "/Adam_1")}
SLOT_INIT = {"SGD": (None, None), "Adagrad": (None, "init"), "Ftrl": (0.0, "init"), "RMSProp": (1.0, 0.0),
             "Adam": (0.0, 0.0)}


def slot_init_values(opt):
    """(initial value of slot a, of slot b) or None where the optimizer has no such slot."""
    a, b = SLOT_INIT[opt[0]]
    init = float(opt[2]) if opt[0] == "Adagrad" else (float(opt[4]) if opt[0] == "Ftrl" else None)
    return (init if a == "init" else a), (init if b == "init" else b)


def _centered(opt):
    return opt[0] == "RMSProp" and len(opt) > 5 and bool(opt[5])


def slot_names(opt):
    """(slot a, slot b, slot c) checkpoint suffixes; slot c only exists for centered RMSProp (mean gradient, init 0)."""
    if _centered(opt):
        return "/RMSProp", "/RMSProp_2", "/RMSProp_1"
    return SLOT_NAMES[opt[0]] + (None,)


def _ftrl_shrink(opt):
    return float(opt[6]) if (opt[0] == "Ftrl" and len(opt) > 6) else 0.0


def _ftrl_is_general(opt):
    """an Ftrl tuple that leaves the default kernel: learning_rate_power != -0.5 or l2_shrinkage != 0"""
    return opt[0] == "Ftrl" and ((len(opt) > 5 and opt[5] != -0.5) or _ftrl_shrink(opt) != 0.0)


def _ftrl_general(w, z, n, g, lr, l1, l2, lr_power, l2_shrink=0.0):
    """FtrlCompute (TF training_ops.cc) with lr_power != -0.5 and / or l2_shrinkage on torch tensors (elementwise, fp32): the
    linear slot takes g + 2 l2_shrinkage var, the accumulator the plain g^2 (pinned by ftrl_test.testFtrlWithL1_L2_L2Shrinkage)."""
    n_new = n + g * g
    if lr_power == -0.5:
        pn, po = torch.sqrt(n_new), torch.sqrt(n)
    else:
        pn, po = torch.pow(n_new, -lr_power), torch.pow(n, -lr_power)
    z_new = z + (g + 2.0 * l2_shrink * w) - (pn - po) / lr * w
    quad = pn / lr + 2.0 * l2
    w_new = torch.where(z_new.abs() > l1, (torch.sign(z_new) * l1 - z_new) / quad, torch.zeros_like(w))
    return w_new, z_new, n_new


def opt_apply_dense(opt, state, nm, g, pow_):
    kind = opt[0]
    w = state[nm]
    sa, sb, sc = slot_names(opt)
    if kind == "SGD":
        w -= opt[1] * g
    elif kind == "Adagrad":
        adagrad_dense(w, state[nm + sb], g, opt[1])
    elif _ftrl_is_general(opt):
        wn, zn, nn = _ftrl_general(w, state[nm + sa], state[nm + sb], g.reshape(w.shape), opt[1], opt[2], opt[3], opt[5],
                                   _ftrl_shrink(opt))
        w.copy_(wn); state[nm + sa].copy_(zn); state[nm + sb].copy_(nn)
    elif kind == "Ftrl":
        ftrl_dense(w, state[nm + sa], state[nm + sb], g, opt[1], opt[2], opt[3])
    elif kind == "RMSProp":
        lr, decay, mom, eps = opt[1:5]
        ms, mo = state[nm + sa], state[nm + sb]
        ms += (g * g - ms) * (1.0 - decay)
        if sc is not None:
            mg = state[nm + sc]
            mg += (g - mg) * (1.0 - decay)
            mo.mul_(mom).add_(lr * g / torch.sqrt(ms - mg * mg + eps))
        else:
            mo.mul_(mom).add_(lr * g / torch.sqrt(ms + eps))
        w -= mo
    elif kind == "Adam":
        _, lr, b1, b2, eps = opt
        m, v = state[nm + sa], state[nm + sb]
        lr_t = np.float32(lr) * np.sqrt(np.float32(1.0) - np.float32(pow_[1])) / (np.float32(1.0) - np.float32(pow_[0]))
        m += (g - m) * (1.0 - b1)
        v += (g * g - v) * (1.0 - b2)
        w -= float(lr_t) * m / (torch.sqrt(v) + eps)
    else:
        raise ValueError(kind)


def opt_apply_rows(opt, state, nm, uniq, rg, pow_):
    """Sparse apply of the summed per-row gradients rg [U, D] at rows uniq [U]."""
    kind = opt[0]
    w = state[nm]
    sa, sb, sc = slot_names(opt)
    idx = torch.as_tensor(np.asarray(uniq, dtype=np.int64))
    w2 = w.reshape(w.shape[0], -1)
    if len(uniq) == 0 and kind != "Adam":     # no row has a gradient (an all-empty column); Adam still decays / moves
        return
    rg = rg.reshape(len(uniq), w2.shape[1])
    if kind == "SGD":
        w2[idx] -= opt[1] * rg
    elif kind == "Adagrad":
        adagrad_rows(w2, state[nm + sb].reshape(w2.shape), uniq, rg, opt[1])
    elif _ftrl_is_general(opt):
        z_t, n_t = state[nm + sa].reshape(w2.shape), state[nm + sb].reshape(w2.shape)
        wn, zn, nn = _ftrl_general(w2[idx], z_t[idx], n_t[idx], rg, opt[1], opt[2], opt[3], opt[5], _ftrl_shrink(opt))
        w2[idx] = wn
        z_t[idx] = zn
        n_t[idx] = nn
    elif kind == "Ftrl":
        ftrl_rows(w, state[nm + sa], state[nm + sb], uniq, rg, opt[1], opt[2], opt[3])
    elif kind == "RMSProp":
        lr, decay, mom, eps = opt[1:5]
        ms_t, mo_t = state[nm + sa].reshape(w2.shape), state[nm + sb].reshape(w2.shape)
        ms = ms_t[idx]
        ms = ms + (rg * rg - ms) * (1.0 - decay)
        if sc is not None:
            mg_t = state[nm + sc].reshape(w2.shape)
            mg = mg_t[idx]
            mg = mg + (rg - mg) * (1.0 - decay)
            mg_t[idx] = mg
            mo = mo_t[idx] * mom + lr * rg / torch.sqrt(ms - mg * mg + eps)
        else:
            mo = mo_t[idx] * mom + lr * rg / torch.sqrt(ms + eps)
        ms_t[idx] = ms
        mo_t[idx] = mo
        w2[idx] -= mo
    elif kind == "Adam":
        _, lr, b1, b2, eps = opt
        m, v = state[nm + sa].reshape(w2.shape), state[nm + sb].reshape(w2.shape)
        lr_t = np.float32(lr) * np.sqrt(np.float32(1.0) - np.float32(pow_[1])) / (np.float32(1.0) - np.float32(pow_[0]))
        m *= b1
        m[idx] += (1.0 - b1) * rg
        v *= b2
        v[idx] += (1.0 - b2) * rg * rg
        w2 -= float(lr_t) * m / (torch.sqrt(v) + eps)
    else:
        raise ValueError(kind)


def adam_pow_names(dnn_opt, lin_opt, has_deep=True, has_wide=True):
    """Names of the non-slot beta-power variables: the first Adam optimizer built by python/lib/joint.py:224-262 (dnn, then
    linear) owns beta1_power / beta2_power, a second one gets the _1 suffix (TF name uniquification)."""
    out, n = {}, 0
    for scope, opt, on in (("dnn", dnn_opt, has_deep), ("linear", lin_opt, has_wide)):
        if on and opt[0] == "Adam":
            suf = "" if n == 0 else "_%d" % n
            out[scope] = ("beta1_power" + suf, "beta2_power" + suf)
            n += 1
    return out


def bce_sum(logits, labels, weights=None):
    n = logits.numel()
    dl = torch.zeros(n, dtype=torch.float32)
    pr = torch.zeros(n, dtype=torch.float32)
    loss = lib().wdo_bce_sum(_p(logits.contiguous()), _p(labels.contiguous()),
                             _p(weights.contiguous()) if weights is not None else None, ctypes.c_int64(n), _p(dl), _p(pr))
    return float(loss), dl, pr


# ---------------------------------------------------------------------------
# activations (python/lib/utils/model_util.py:44-55)
# ---------------------------------------------------------------------------
ACTIVATIONS = {
    "sigmoid": torch.sigmoid,
    "tanh": torch.tanh,
    "relu": torch.relu,
    "relu6": torch.nn.functional.relu6,
    "leaky_relu": lambda x: torch.nn.functional.leaky_relu(x, 0.2),  # tf.nn.leaky_relu default alpha=0.2
    "elu": torch.nn.functional.elu,
    "selu": torch.selu,
    "softplus": torch.nn.functional.softplus,
    "softsign": torch.nn.functional.softsign,
    # tf.nn.crelu: concat(relu(x), relu(-x)) on the last axis -- the next layer (and this layer's BN) sees 2N features
    "crelu": lambda x: torch.cat([torch.relu(x), torch.relu(-x)], dim=-1),
}


def connection_map(mode):
    """python/lib/dnn.py:195-205: ['0-1', '0-3', '1-2'] (or '0-1,0-3,1-2', or (i, j) pairs) -> {j: [i, ...]}.  The reference
    line `connected_mapping[j] = connected_mapping[j].append(i)` (dnn.py:203) stores None for a second edge into j and
    `connected_mapping[layer_id + 1]` (dnn.py:220) raises KeyError for a layer nothing connects to: this is
    what the docstring (dnn.py:65-66) says the list means -- every edge kept, unconnected layers read their predecessor.
    (Under Python 2, which `map(...)` being subscripted-by-unpacking at dnn.py:196-199 assumes, the comprehension at :220 also
    rebinds `net` to the last collected tensor, so the layer's own output would be dropped; the branch cannot have been run.)
    No golden vector exists for this mode: parity is pinned by construction only -- the chain list 0-1, 1-2, ... must
    reproduce `dense` (same concat, same order) and the result must equal torch autograd of this function."""
    if isinstance(mode, str):
        mode = [p for p in mode.replace(" ", "").split(",") if p]
    cm = {}
    for e in mode:
        i, j = (int(v) for v in (e.split("-") if isinstance(e, str) else e))
        if i not in cm.setdefault(j, []):
            cm[j].append(i)
    return cm


def is_connection_list(mode):
    return not isinstance(mode, str) or (len(mode) >= 3 and mode.replace(" ", "").replace(",", "").replace("-", "").isdigit())


def tower_forward(x, tw, mode, act, batch_norm, dropout=None, masks=None):
    """python/lib/dnn.py:92-234 for one tower.  tw: dict with lists 'kernel','bias','gamma','beta' (hidden
    layers) and 'logits_kernel','logits_bias'.  BN is the inference-mode affine of SURVEY App. C.1.
    dropout (rate) + masks (per layer [B, N] of 0 / 1, the keep decisions): tf.layers.dropout in TRAIN mode, between
    the activation and BN (dnn.py:111-114): x / keep_prob * keep.  The random draw itself is the caller's."""
    f = ACTIVATIONS[act]
    cmap = None
    if is_connection_list(mode):
        cmap, mode = connection_map(mode), "list"
    inv = 1.0 / float(np.sqrt(np.float32(1.0) + np.float32(BN_EPS)))
    input_layer = x
    net = x
    coll = [x]
    for l in range(len(tw["kernel"])):
        h = f(net @ tw["kernel"][l] + tw["bias"][l])
        if dropout and masks is not None:
            h = h / np.float32(1.0 - dropout) * torch.as_tensor(masks[l])
        if batch_norm:
            h = h * (tw["gamma"][l] * inv) + tw["beta"][l]
        if mode == "simple":
            net = h
        elif mode == "first_dense":
            net = torch.cat([h, input_layer], dim=1)
        elif mode == "last_dense":
            coll.append(h)
            net = h
        elif mode == "dense":
            coll.append(h)
            net = torch.cat(coll, dim=1)
        elif mode == "resnet":
            net = torch.cat([h, coll[l]], dim=1)
            coll.append(net)
        elif mode == "list":    # dnn.py:218-223: [net_collections[idx] for idx in the mapping, in index order] + [this layer]
            net = torch.cat([coll[i] for i in range(len(coll)) if i in cmap.get(l + 1, ())] + [h], dim=1)
            coll.append(net)
        else:
            raise ValueError(mode)
    if mode == "last_dense":
        net = torch.cat(coll, dim=1)
    return net @ tw["logits_kernel"] + tw["logits_bias"]


class OracleWideDeep:
    """Whole-model oracle.  Columns are described neutrally:

    deep_cols: list of dicts {name, kind: 'embedding'|'indicator'|'numeric', key, num_buckets, dim}
    wide_cols: list of dicts {name, key, num_buckets}
    towers:    list of (hidden_units, mode)
    state:     dict TF-variable-name -> torch.float32 CPU tensor (see wide_deep_amd/checkpoint naming)

    batch: {'ids': {key: (ids int64, offs int32[B+1])}, 'dense': {key: float32[B]},
            'labels': float32[B], 'weights': float32[B] or None}
    """

    def __init__(self, model_type, deep_cols, wide_cols, towers, state, act="relu", batch_norm=True,
                 dnn_opt=("Adagrad", 0.05, 0.1), lin_opt=("Ftrl", 0.1, 0.5, 1.0, 0.1), dropout=None):
        self.model_type = model_type
        # both tf.feature_column.input_layer and linear_model sort columns by name
        self.deep_cols = sorted(deep_cols, key=lambda c: c["name"])
        self.wide_cols = sorted(wide_cols, key=lambda c: c["name"])
        self.towers = towers
        self.state = state
        self.act = act
        self.batch_norm = batch_norm
        self.dnn_opt = dnn_opt
        self.lin_opt = lin_opt
        self.dropout = dropout      # rate; the keep masks of a step come with the batch ("dropout_masks")
        # batched: all sparse columns of a step go through ONE C call each way, columns side by side (same per-column
        # arithmetic, bit-identical results; tests/test_oracle_kat.py) -- the CPU-baseline configuration of bench.py
        self.batched = False

    def _can_batch(self):
        return (self.batched and self.dnn_opt[0] == "Adagrad" and self.lin_opt[0] == "Ftrl" and len(self.lin_opt) <= 5
                and all(c["kind"] != "indicator" for c in self.deep_cols))

    # -- Adam beta powers (non-slot variables, one pair per optimizer instance) ---------------------
    def _pow_names(self):
        return adam_pow_names(self.dnn_opt, self.lin_opt, self.model_type in ("deep", "wide_deep"),
                              self.model_type in ("wide", "wide_deep"))

    def _pow(self, scope):
        names = self._pow_names().get(scope)
        if names is None:
            return None
        opt = self.dnn_opt if scope == "dnn" else self.lin_opt
        for nm, b in zip(names, (opt[2], opt[3])):
            if nm not in self.state:
                self.state[nm] = torch.tensor(float(b), dtype=torch.float32)
        return float(self.state[names[0]]), float(self.state[names[1]])

    def _tick(self, scope):
        names = self._pow_names().get(scope)
        if names is not None:
            opt = self.dnn_opt if scope == "dnn" else self.lin_opt
            self.state[names[0]] = self.state[names[0]] * np.float32(opt[2])
            self.state[names[1]] = self.state[names[1]] * np.float32(opt[3])

    # -- variable names ----------------------------------------------------
    @staticmethod
    def emb_name(col):
        return "dnn/input_from_feature_columns/input_layer/%s/embedding_weights" % col["name"]

    @staticmethod
    def wide_name(col):
        return "linear/linear_model/%s/weights" % col["name"]

    def tower_params(self, t):
        hidden, _ = self.towers[t]
        p = "dnn/dnn_%d/" % (t + 1)
        tw = {"kernel": [], "bias": [], "gamma": [], "beta": []}
        for l in range(len(hidden)):
            tw["kernel"].append(self.state[p + "hiddenlayer_%d/kernel" % l])
            tw["bias"].append(self.state[p + "hiddenlayer_%d/bias" % l])
            if self.batch_norm:
                tw["gamma"].append(self.state[p + "hiddenlayer_%d/batch_normalization/gamma" % l])
                tw["beta"].append(self.state[p + "hiddenlayer_%d/batch_normalization/beta" % l])
        tw["logits_kernel"] = self.state[p + "logits/kernel"]
        tw["logits_bias"] = self.state[p + "logits/bias"]
        return tw

    # -- forward -------------------------------------------------------------
    def input_layer(self, batch):
        B = len(batch["labels"]) if "labels" in batch else batch["batch_size"]
        if self._can_batch():
            width = sum(c["dim"] if c["kind"] == "embedding" else 1 for c in self.deep_cols)
            x = torch.zeros(B, width, dtype=torch.float32)
            tabs, io, outs, col0 = [], [], [], 0
            for c in self.deep_cols:
                if c["kind"] == "embedding":
                    tabs.append(self.state[self.emb_name(c)])
                    io.append(batch["ids"][c["key"]])
                    outs.append(x.data_ptr() + 4 * col0)
                    col0 += c["dim"]
                else:
                    x[:, col0] = torch.as_tensor(np.asarray(batch["dense"][c["key"]], dtype=np.float32))
                    col0 += 1
            embag_fwd_cols(tabs, io, B, True, outs, [width] * len(tabs))
            return x
        parts = []
        for c in self.deep_cols:
            if c["kind"] == "embedding":
                ids, offs = batch["ids"][c["key"]]
                parts.append(embag_fwd(self.state[self.emb_name(c)], ids, offs, mean=True))
            elif c["kind"] == "indicator":
                ids, offs = batch["ids"][c["key"]]
                o = torch.zeros(B, c["num_buckets"], dtype=torch.float32)
                for b in range(B):
                    for j in range(offs[b], offs[b + 1]):
                        if ids[j] >= 0:
                            o[b, int(ids[j])] += 1.0
                parts.append(o)
            else:
                parts.append(torch.as_tensor(np.asarray(batch["dense"][c["key"]], dtype=np.float32)).reshape(B, 1))
        return torch.cat(parts, dim=1) if parts else torch.zeros(B, 0)

    def wide_logits(self, batch):
        B = len(batch["labels"]) if "labels" in batch else batch["batch_size"]
        out = torch.zeros(B, dtype=torch.float32)
        if self._can_batch() and self.wide_cols:
            per = torch.zeros(len(self.wide_cols), B, dtype=torch.float32)
            embag_fwd_cols([self.state[self.wide_name(c)] for c in self.wide_cols],
                           [batch["ids"][c["key"]] for c in self.wide_cols], B, False,
                           [per.data_ptr() + 4 * B * j for j in range(len(self.wide_cols))], [1] * len(self.wide_cols))
            for j in range(len(self.wide_cols)):          # same column order as the per-column path
                out += per[j]
            return out + self.state["linear/linear_model/bias_weights"][0]
        for c in self.wide_cols:
            ids, offs = batch["ids"][c["key"]]
            out += embag_fwd(self.state[self.wide_name(c)], ids, offs, mean=False)[:, 0]
        return out + self.state["linear/linear_model/bias_weights"][0]

    def forward(self, batch, need_grad=False):
        B = len(batch["labels"]) if "labels" in batch else batch["batch_size"]
        logits = torch.zeros(B, dtype=torch.float32)
        cache = {}
        if self.model_type in ("deep", "wide_deep"):
            x = self.input_layer(batch)
            x.requires_grad_(need_grad)
            dnn = None
            params = []
            for t, (hidden, mode) in enumerate(self.towers):
                tw = self.tower_params(t)
                if need_grad:
                    for k in ("kernel", "bias", "gamma", "beta"):
                        for v in tw[k]:
                            v.requires_grad_(True)
                    tw["logits_kernel"].requires_grad_(True)
                    tw["logits_bias"].requires_grad_(True)
                dm = batch.get("dropout_masks") if need_grad else None      # TRAIN mode only
                lg = tower_forward(x, tw, mode, self.act, self.batch_norm, self.dropout if dm is not None else None,
                                   dm[t] if dm is not None else None)[:, 0]
                dnn = lg if dnn is None else dnn + lg
                params.append(tw)
            cache.update(x=x, dnn=dnn, params=params)
            logits = logits + dnn.detach()
        if self.model_type in ("wide", "wide_deep"):
            wl = self.wide_logits(batch)
            cache["wide"] = wl
            logits = logits + wl
        return logits, cache

    def predict(self, batch):
        with torch.no_grad():
            logits, _ = self.forward(batch, need_grad=False)
        return logits, torch.sigmoid(logits)

    # -- one train step (joint.py:224-262 with TF optimizer semantics) --------
    def train_step(self, batch):
        labels = torch.as_tensor(np.asarray(batch["labels"], dtype=np.float32))
        weights = batch.get("weights")
        weights = torch.as_tensor(np.asarray(weights, dtype=np.float32)) if weights is not None else None
        logits, cache = self.forward(batch, need_grad=True)
        loss, dlogit, _ = bce_sum(logits, labels, weights)

        if "dnn" in cache:
            cache["dnn"].backward(dlogit)
            opt = self.dnn_opt
            pw = self._pow("dnn")
            dx = cache["x"].grad
            # dense tower params
            with torch.no_grad():
                for t, tw in enumerate(cache["params"]):
                    names = []
                    p = "dnn/dnn_%d/" % (t + 1)
                    for l in range(len(tw["kernel"])):
                        names += [p + "hiddenlayer_%d/kernel" % l, p + "hiddenlayer_%d/bias" % l]
                        if self.batch_norm:
                            names += [p + "hiddenlayer_%d/batch_normalization/gamma" % l,
                                      p + "hiddenlayer_%d/batch_normalization/beta" % l]
                    names += [p + "logits/kernel", p + "logits/bias"]
                    for nm in names:
                        v = self.state[nm]
                        g = v.grad
                        v.requires_grad_(False)
                        opt_apply_dense(opt, self.state, nm, g, pw)
                        v.grad = None
            # embedding rows
            col0 = 0
            if self._can_batch():
                dxc = dx.contiguous()
                tabs, accs, io, gp = [], [], [], []
                for c in self.deep_cols:
                    if c["kind"] == "embedding":
                        nm = self.emb_name(c)
                        tabs.append(self.state[nm])
                        accs.append(self.state[nm + "/Adagrad"])
                        io.append(batch["ids"][c["key"]])
                        gp.append(dxc.data_ptr() + 4 * col0)
                        col0 += c["dim"]
                    else:
                        col0 += 1
                sparse_apply_cols(tabs, [None] * len(tabs), accs, io, len(labels), True, gp, [dxc.shape[1]] * len(tabs),
                                  0, opt[1])
            for c in ([] if self._can_batch() else self.deep_cols):
                if c["kind"] == "embedding":
                    D = c["dim"]
                    ids, offs = batch["ids"][c["key"]]
                    uniq, rg = embag_row_grads(D, ids, offs, dx[:, col0:col0 + D], mean=True)
                    opt_apply_rows(opt, self.state, self.emb_name(c), uniq, rg, pw)
                    col0 += D
                elif c["kind"] == "indicator":
                    col0 += c["num_buckets"]
                else:
                    col0 += 1
            self._tick("dnn")
        if "wide" in cache:
            opt = self.lin_opt
            pw = self._pow("linear")
            if self._can_batch() and self.wide_cols:
                dl = dlogit.contiguous()
                names = [self.wide_name(c) for c in self.wide_cols]
                sparse_apply_cols([self.state[n] for n in names], [self.state[n + "/Ftrl_1"] for n in names],
                                  [self.state[n + "/Ftrl"] for n in names], [batch["ids"][c["key"]] for c in self.wide_cols],
                                  len(labels), False, [dl.data_ptr()] * len(names), [1] * len(names), 1, opt[1], opt[2],
                                  opt[3])
            for c in ([] if self._can_batch() else self.wide_cols):
                ids, offs = batch["ids"][c["key"]]
                uniq, rg = embag_row_grads(1, ids, offs, dlogit.reshape(-1, 1), mean=False)
                opt_apply_rows(opt, self.state, self.wide_name(c), uniq, rg, pw)
            opt_apply_dense(opt, self.state, "linear/linear_model/bias_weights", dlogit.sum().reshape(1), pw)
            self._tick("linear")
        return loss, logits
