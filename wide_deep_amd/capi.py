"""ctypes binding of the C-ABI hot-path library (include/wd_hip.h -> _lib/libwd_hip.so).

The product path has NO CPU fallback: if the HIP library is missing or a call
fails, this module raises.  (The CPU oracle under oracle/ is test infrastructure
and is never imported from here.)
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch's bundled HIP runtime first so libwd_hip.so binds to the same one)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libwd_hip.so")

WD_MAX_CROSS_KEYS = 8
WD_FOLD_PARTS = 16

SLOT_NONE, SLOT_EMBEDDING, SLOT_INDICATOR = 0, 1, 2
SLOT_F_SMALL = 1              # wd_slot_t.flags (include/wd_hip.h)
SMALL_MAX_FLOATS = 8192       # WD_SMALL_MAX_FLOATS

ACT_IDS = {
    None: 0, "none": 0, "relu": 1, "sigmoid": 2, "tanh": 3, "relu6": 4, "leaky_relu": 5, "elu": 6, "selu": 7,
    "softplus": 8, "softsign": 9,
}


class WdSlot(ctypes.Structure):
    _fields_ = [
        ("emb_off", ctypes.c_int64),
        ("row_base", ctypes.c_int64),
        ("num_buckets", ctypes.c_int32),
        ("dim", ctypes.c_int32),
        ("out_col", ctypes.c_int32),
        ("kind", ctypes.c_int32),
        ("wide", ctypes.c_int32),
        ("bucket_shift", ctypes.c_int32),
        ("bucket_base", ctypes.c_int32),
        ("flags", ctypes.c_int32),
    ]


class WdDenseCol(ctypes.Structure):
    _fields_ = [("p0", ctypes.c_float), ("p1", ctypes.c_float), ("kind", ctypes.c_int32), ("out_col", ctypes.c_int32)]


class WdMlpLayer(ctypes.Structure):
    _fields_ = [
        ("w_off", ctypes.c_int64), ("b_off", ctypes.c_int64), ("K", ctypes.c_int64), ("N", ctypes.c_int64),
        ("gamma_idx", ctypes.c_void_p), ("beta_idx", ctypes.c_void_p),
        ("Wf", ctypes.c_void_p), ("bf", ctypes.c_void_p), ("s", ctypes.c_void_p), ("t", ctypes.c_void_p),
        ("Gpart", ctypes.c_void_p), ("nsplit", ctypes.c_int32), ("pad_", ctypes.c_int32),
        ("WfT_h", ctypes.c_void_p), ("ld_wft_h", ctypes.c_int64), ("cat_off", ctypes.c_void_p), ("wcat", ctypes.c_void_p),
    ]


WD_CHAIN_MAX_LAYERS = 6
WD_TN_GROUP_MAX = 20


WD_OPT_KINDS = {"SGD": 0, "Adagrad": 1, "Ftrl": 2, "RMSProp": 3, "Adam": 4}
WD_OPT_RMSPROP_CENTERED = 6


class WdOpt(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("lr", ctypes.c_float), ("p0", ctypes.c_float), ("p1", ctypes.c_float),
                ("p2", ctypes.c_float), ("p3", ctypes.c_float), ("pow", ctypes.c_void_p), ("slot_c", ctypes.c_void_p)]


class WdChainInput(ctypes.Structure):
    _fields_ = [
        ("emb", ctypes.c_void_p), ("slots", ctypes.c_void_p), ("ids", ctypes.c_void_p), ("wide", ctypes.c_void_p),
        ("wide_bias", ctypes.c_void_p), ("wide_out", ctypes.c_void_p), ("dense", ctypes.c_void_p), ("cols", ctypes.c_void_p),
        ("x_out", ctypes.c_void_p), ("ld_dense", ctypes.c_int64), ("S", ctypes.c_int32), ("slot0", ctypes.c_int32),
        ("ngroup", ctypes.c_int32), ("dim", ctypes.c_int32), ("ncols", ctypes.c_int32), ("row_stride", ctypes.c_int32),
        ("wide_in_row", ctypes.c_int32), ("pad_", ctypes.c_int32),
    ]


WD_CHAIN_MAX_SLOTS = 128


class WdChainOpts(ctypes.Structure):
    _fields_ = [("input", ctypes.c_void_p), ("loss_part", ctypes.c_void_p), ("stamps", ctypes.c_void_p),
                ("tile_stamps", ctypes.c_void_p), ("row_tile", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("wide_vals", ctypes.c_void_p), ("wide_bias", ctypes.c_void_p), ("wide_out", ctypes.c_void_p),
                ("wide_S", ctypes.c_int32), ("pad_", ctypes.c_int32), ("windows", ctypes.c_void_p)]


class WdChainWindows(ctypes.Structure):
    _fields_ = [("seg_col", ctypes.c_int32 * (WD_CHAIN_MAX_LAYERS + 1)), ("in_col", ctypes.c_int32 * (WD_CHAIN_MAX_LAYERS + 1)),
                ("k_logits", ctypes.c_int32), ("cols", ctypes.c_int32)]


class WdApplyNext(ctypes.Structure):
    _fields_ = [("bucket_start", ctypes.c_void_p), ("pairs", ctypes.c_void_p), ("x", ctypes.c_void_p),
                ("wide_vals", ctypes.c_void_p), ("ldx", ctypes.c_int64), ("unsorted_buckets", ctypes.c_int32),
                ("pad_", ctypes.c_int32)]


class WdTnJob(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("Cpart", ctypes.c_void_p),
        ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("M", ctypes.c_int64), ("N", ctypes.c_int64),
        ("K", ctypes.c_int64), ("nsplit", ctypes.c_int32), ("append_ones", ctypes.c_int32),
    ]


class WdChainLayer(ctypes.Structure):
    _fields_ = [
        ("Wpk", ctypes.c_void_p), ("WTpk", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("gamma", ctypes.c_void_p),
        ("beta", ctypes.c_void_p), ("a_out", ctypes.c_void_p), ("dz_out", ctypes.c_void_p), ("db_part", ctypes.c_void_p),
        ("dgamma_part", ctypes.c_void_p), ("dbeta_part", ctypes.c_void_p), ("K", ctypes.c_int32), ("N", ctypes.c_int32),
    ]


WD_TAIL_GRAD, WD_TAIL_UPDATE, WD_TAIL_PACK = 1, 2, 4


class WdTailLayer(ctypes.Structure):
    _fields_ = [
        ("w_off", ctypes.c_int64), ("b_off", ctypes.c_int64), ("gamma_off", ctypes.c_int64), ("beta_off", ctypes.c_int64),
        ("K", ctypes.c_int64), ("N", ctypes.c_int64), ("Gpart", ctypes.c_void_p), ("db_sum", ctypes.c_void_p),
        ("dgamma_sum", ctypes.c_void_p), ("dbeta_sum", ctypes.c_void_p), ("Wpk", ctypes.c_void_p), ("WTpk", ctypes.c_void_p),
        ("nsplit", ctypes.c_int32), ("pk_tile", ctypes.c_int32),
        ("nseg", ctypes.c_int32), ("pad_", ctypes.c_int32),
        ("seg_k0", ctypes.c_int32 * (WD_CHAIN_MAX_LAYERS + 1)), ("seg_w", ctypes.c_int32 * (WD_CHAIN_MAX_LAYERS + 1)),
        ("seg_red0", ctypes.c_int32 * (WD_CHAIN_MAX_LAYERS + 1)), ("seg_kred", ctypes.c_int32 * (WD_CHAIN_MAX_LAYERS + 1)),
        ("seg_wt", ctypes.c_void_p * (WD_CHAIN_MAX_LAYERS + 1)),
    ]


class WdCrossKeys(ctypes.Structure):
    _fields_ = [
        ("vals", ctypes.c_void_p * WD_MAX_CROSS_KEYS),
        ("offs", ctypes.c_void_p * WD_MAX_CROSS_KEYS),
        ("nkeys", ctypes.c_int32),
    ]


WD_FEAT_KINDS = {"hash": 0, "vocab": 1, "identity": 2, "bucket": 3, "cross": 4}
WD_FEAT_KEY_KINDS = {"string": 0, "identity": 1, "bucket": 2}


class WdFeatKey(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("src", ctypes.c_int32), ("num_buckets", ctypes.c_int32), ("nbound", ctypes.c_int32),
                ("bound_off", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class WdFeatSlot(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("src", ctypes.c_int32), ("num_buckets", ctypes.c_int32), ("nbound", ctypes.c_int32),
                ("bound_off", ctypes.c_int32), ("norm_kind", ctypes.c_int32), ("p0", ctypes.c_float), ("p1", ctypes.c_float),
                ("nkeys", ctypes.c_int32), ("pad_", ctypes.c_int32), ("hash_key", ctypes.c_uint64),
                ("keys", WdFeatKey * WD_MAX_CROSS_KEYS)]


class WdFeatBatch(ctypes.Structure):
    _fields_ = [("fp", ctypes.c_void_p), ("tok_val", ctypes.c_void_p), ("ex_offs", ctypes.c_void_p),
                ("tok_base", ctypes.c_void_p), ("lmax", ctypes.c_void_p), ("ints", ctypes.c_void_p),
                ("floats", ctypes.c_void_p), ("bounds", ctypes.c_void_p), ("batch", ctypes.c_int64), ("S", ctypes.c_int32),
                ("empty_index", ctypes.c_int32), ("ntok_dev", ctypes.c_void_p)]


class WdFeatVocab(ctypes.Structure):
    _fields_ = [("bytes", ctypes.c_void_p), ("offs", ctypes.c_void_p), ("nvocab", ctypes.c_int32), ("pad_", ctypes.c_int32)]


P = ctypes.c_void_p
I32, I64, U64, F32, SZ = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float, ctypes.c_size_t

# name -> argtypes; every function returns int status unless listed in _RESTYPES
_PROTOS = {
    "wd_abi_version": [],
    "wd_build_stamp": [],
    "wd_fingerprint64": [P, P, I64, P, P],
    "wd_fingerprint64_dyn": [P, P, I64, P, P, P],
    "wd_feat_vocab_lookup_all": [P, P, I64, P, P, P, I32, P, P, P],
    "wd_hash_bucket": [P, P, I64, P, I64, P, I32, P, P],
    "wd_emit_hash_slot": [P, P, I64, U64, P, I32, I32, P, P],
    "wd_emit_int_slot": [P, P, I64, P, I32, I32, P, P],
    "wd_cross_hash": [ctypes.POINTER(WdCrossKeys), I64, U64, U64, P, I32, I32, P, P],
    "wd_feat_vocab_lookup": [P, P, I64, I64, P, P, I32, P, P],
    "wd_feat_lens": [P, ctypes.POINTER(WdFeatBatch), P, P, P],
    "wd_feat_offsets_workspace_bytes": [I64],
    "wd_feat_offsets": [P, P, I64, P, I64, P, P],
    "wd_feat_emit": [P, ctypes.POINTER(WdFeatBatch), P, P, I64, P],
    "wd_embag_fwd": [P, P, I32, P, I32, I32, P, P, I64, P, I64, P],
    "wd_embag_fwd_range": [P, P, I32, I32, I32, I32, P, P, I64, P, I64, P],
    "wd_input_layer_fwd": [P, P, I32, I32, I32, I32, P, P, I32, I64, P, I64, P, I64, P, I32, P, P, P, P],
    "wd_indicator_fwd": [P, I32, P, I32, P, P, I64, P, I64, P],
    "wd_dense_fwd": [P, I64, P, I32, I64, P, I64, P],
    "wd_wide_fwd": [P, I32, P, P, I32, P, P, I64, P, P],
    "wd_small_tables_ws_floats": [I32, I32, I32, I64],
    "wd_embag_fwd_wide": [P, I64, P, I32, P, I32, I32, P, P, I64, P, I64, P, P],
    "wd_wide_sum": [P, P, P, I32, I64, P, P],
    "wd_small_tables_fwd": [P, P, P, I32, P, I32, I32, I32, P, P, I64, P, I64, P, I32, P],
    "wd_small_tables_bwd": [P, P, P, P, I32, P, I32, I32, I32, P, P, I64, P, I64, P, F32, F32, F32, F32, P, I64, I32, P],
    "wd_small_tables_grad": [P, I32, P, I32, I32, I32, P, P, I64, P, I64, P, P, I64, P, P],
    "wd_small_tables_apply": [P, P, P, P, I32, P, I32, I32, I32, P, F32, F32, F32, F32, I32, P],
    "wd_bce_sum_fwd_bwd": [P, P, P, P, I64, P, P, P, P, P],
    "wd_bce_loss_sum": [P, P, P, I64, P, P],
    "wd_sort_workspace_bytes": [I64, I32],
    "wd_build_sort_keys": [P, I32, P, P, I64, I64, P, P, P],
    "wd_sort_pairs": [P, P, P, P, I64, I32, P, SZ, P],
    "wd_embag_bwd_adagrad": [P, P, P, I32, I32, P, P, I64, P, P, I64, F32, P],
    "wd_wide_bwd_ftrl": [P, P, I32, P, P, I64, P, F32, F32, F32, P],
    "wd_bias_ftrl": [P, P, I64, F32, F32, F32, P],
    "wd_bucket_max": [],
    "wd_bucket_chunks": [],
    "wd_sparse_bwd_fused": [P, P, P, P, P, I32, P, P, I64, I64, P, I64, P, I64, F32, F32, F32, F32, P, P, P, P, I32, P],
    "wd_sparse_bucketize": [P, I32, P, P, I64, I64, P, P, P, P, I32, P],
    "wd_sparse_apply": [P, P, P, P, P, I32, P, I64, P, I64, P, I64, F32, F32, F32, F32, P, P, I32, P],
    "wd_sparse_apply_rec": [P, I32, I32, P, P, P, I32, P, I64, P, I64, P, I64, F32, F32, F32, F32, P, P, I32, P, P],
    "wd_bucket_onehot": [P, I32, P, I32, I64, P, P, I32, I32, P, P, P],
    "wd_bucket_sort": [P, P, I32, P, I32, P, I64, I32, P, P, P, P],
    "wd_row_update": [P, I32, I32, P, P, P, I32, I64, P, I64, P, F32, F32, F32, F32, P, P, I32, P, P, P],
    "wd_bucket_sort_ragged": [P, P, I32, P, I32, P, I64, I32, I64, P],
    "wd_row_update_ragged": [P, I32, I32, P, P, P, I32, I64, P, P, I64, P, I64, F32, F32, F32, F32, P, I64, P, P, I32, P],
    "wd_hash_bucket_cols": [P, P, I64, P, I32, P, P, P],
    "wd_prefetch_onehot_blocks": [I64, I32, I32, I32],
    "wd_prefetch_onehot": [P, I32, I32, P, I32, P, I64, P, I64, P, P, I64, P, I32, P, P],
    "wd_route_chunks": [],
    "wd_route_build": [P, I32, I32, P, P, I64, I32, P, P, P, P, P, P],
    "wd_owner_gather": [P, I64, I32, P, P, I64, P, I32, P],
    "wd_owner_gather_rec": [P, I32, I32, P, I64, P, I32, P],
    "wd_grad_pack": [P, I32, P, P, I64, P, I64, P, I32, I32, P, P],
    "wd_fill_i32": [P, I32, I64, P],
    "wd_embag_fwd_strided": [P, I64, P, I32, P, I32, I32, P, P, I64, P, I64, P],
    "wd_gemm_nn_bias_act": [P, I64, P, I64, P, I32, I32, P, I64, I64, I64, I64, P],
    "wd_gemm_nt": [P, I64, P, I64, P, I64, I64, I64, I64, I32, P],
    "wd_gemm_nt_actbwd": [P, I64, P, I64, P, I64, I64, I64, I64, P, I64, I32, P],
    "wd_gemm_tn_splitk": [P, I64, P, I64, P, I64, I64, I64, I32, I32, P],
    "wd_fold_affine_all": [P, P, I32, I64, F32, P, I64, P, I64, P],
    "wd_mlp_finalize_all": [P, I32, I64, P, F32, P, P],
    "wd_crelu_tie": [P, I64, I64, I64, I64, P],
    "wd_mlp_finalize_adagrad_all": [P, I32, I64, P, P, F32, P, F32, P],
    "wd_logits_head_blocks": [I64, I64],
    "wd_gemm_tn_splitk_group": [P, I32, P],
    "wd_sparse_apply_opt": [P, P, P, P, P, P, I32, P, I64, P, I64, P, I64, P, P, P, P, I32, P, P],
    "wd_opt_dense": [P, P, P, P, I64, P, P],
    "wd_adam_untouched": [P, P, P, P, P, I32, I64, I64, P, P, P, P],
    "wd_adam_tick": [P, F32, F32, P],
    "wd_tower_chain_lds_bytes": [I32, P, I32, I32],
    "wd_tower_chain_windows_lds_bytes": [P, I32, P, I32],
    "wd_tower_chain_blocks": [I64, I32],
    "wd_tower_chain": [P, I64, I32, P, I32, I32, F32, P, P, P, P, P, I64, P, P, P, P, P, P, P, I64, I32, P, P],
    "wd_chain_tail": [P, I32, P, P, P, F32, F32, I32, P],
    "wd_route_unique": [P, I64, I32, I32, P, P, P, P, P, P],
    "wd_row_grad_presum": [P, I32, I64, P, I64, P, I32, P, P, I32, P, P, I32, P],
    "wd_logits_head_h": [P, I64, I64, P, P, I32, P, P, P, I64, P, P, P, P, P, P, I64, I32, P, P],
    "wd_hgemm_nn": [P, I64, P, I64, P, I32, I32, P, I64, P, I64, I64, I64, I64, P],
    "wd_hgemm_nt": [P, I64, P, I64, I64, I64, I64, P, I64, I32, P, I64, P, I64, P, I64, I32, P],
    "wd_hgemm_tn_splitk": [P, I64, P, I64, P, I64, I64, I64, I32, P],
    "wd_cast_transpose_h": [P, I64, I64, I64, P, I64, I32, P, I64, P, I64, P],
    "wd_logits_head": [P, I64, I64, P, P, I32, P, P, P, I64, P, P, P, P, P, P, I64, I32, P, P],
    "wd_fold_affine": [P, I64, I64, P, P, F32, P, P, P, P, I64, I64, P],
    "wd_act_bwd": [P, I64, P, I64, I32, P, I64, I64, I64, P],
    "wd_dropout_fwd": [P, I64, I64, I64, F32, P, I32, P],
    "wd_act_bwd_dropout": [P, I64, P, I64, I32, P, I64, I64, I64, F32, P, I32, P],
    "wd_counter_tick": [P, P],
    "wd_mlp_finalize": [P, I32, P, I64, I64, P, P, P, P, F32, P, I64, I64, P],
    "wd_adagrad_dense": [P, P, P, I64, F32, P],
    "wd_fill_f32": [P, F32, I64, P],
    "wd_diag_gather64": [P, P, I64, I32, P, P],
    "wd_diag_gather_modes": [P, I64, P, I64, I32, P, P],
    "wd_diag_access": [P, P, P, I32, P, I64, I32, I32, P, P],
}
_RESTYPES = {"wd_build_stamp": ctypes.c_char_p, "wd_prefetch_onehot_blocks": I64, "wd_feat_offsets_workspace_bytes": I64, "wd_sort_workspace_bytes": SZ, "wd_logits_head_blocks": I64, "wd_tower_chain_lds_bytes": I64, "wd_tower_chain_windows_lds_bytes": I64, "wd_tower_chain_blocks": I64, "wd_bucket_max": I32, "wd_bucket_chunks": I32, "wd_route_chunks": I32, "wd_small_tables_ws_floats": I64}

EXPORTED_SYMBOLS = sorted(list(_PROTOS) + ["wd_last_error"])

_lib = None


class WdError(RuntimeError):
    pass


def source_stamp():
    """sha256 over csrc/*.hip, csrc/*.h and include/wd_hip.h in name order -- what csrc/build.sh embeds as wd_build_stamp().
    None when the sources are not there (an installed library without its tree)."""
    import glob
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    names = sorted([os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))]
                   + ["../../include/wd_hip.h"])
    if len(names) < 3:
        return None
    h = hashlib.sha256()
    try:
        for n in names:
            with open(os.path.join(csrc, n), "rb") as f:
                h.update(f.read())
    except OSError:
        return None
    return h.hexdigest()


def load():
    """Load libwd_hip.so (fails loudly when the HIP extension has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("WD_HIP_LIB", LIB_PATH)      # diagnostics: an experiment build of the same ABI
    if not os.path.exists(path):
        raise WdError(
            "HIP extension %s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(wide_deep_amd/csrc/build.sh).  There is no CPU fallback for the hot path." % path)
    lib = ctypes.CDLL(path)
    lib.wd_last_error.restype = ctypes.c_char_p
    lib.wd_last_error.argtypes = []
    if "WD_HIP_LIB" not in os.environ:
        # the prebuilt library ships with the tree (gpurun copies it): it must have been built from THESE sources
        want = source_stamp()
        try:
            lib.wd_build_stamp.restype = ctypes.c_char_p
            got = lib.wd_build_stamp().decode()
        except AttributeError:
            got = "(none: built before the stamp existed)"
        if want is not None and got != want:
            raise WdError("stale HIP library %s: built from sources with stamp %s, the tree has %s -- run "
                          "`python -c 'import __graft_entry__ as g; g.build()'` (wide_deep_amd/csrc/build.sh)" % (path, got[:16], want[:16]))
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    _lib = lib
    return lib


def ptr(t):
    """device/host pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


NCALLS = 0      # launches issued through call() (dist._Segments uses it to skip empty graph segments)


def call(name, *args):
    global NCALLS
    lib = load()
    NCALLS += 1
    rc = getattr(lib, name)(*args)
    if name in _RESTYPES:
        return rc
    if rc != 0:
        raise WdError("%s failed (%d): %s" % (name, rc, lib.wd_last_error().decode()))
    return rc
