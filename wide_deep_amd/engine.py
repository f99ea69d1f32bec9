"""WideDeepEngine -- the MI355X train step behind the reference's model_fn.

One engine instance = what the reference builds in ``_wide_deep_combined_model_fn``
(python/lib/joint.py:81-269): wide logits (python/lib/linear.py:20-36) + deep logits
(python/lib/dnn.py:43-275) -> sigmoid-CE head -> per-scope optimizers (Ftrl for `linear`,
Adagrad for `dnn`).  All arithmetic runs in hand-written gfx950 kernels reached through the C ABI
(include/wd_hip.h); torch is used for device memory, streams and graph capture only.

There is no CPU fallback: constructing an engine without a GPU or without the built HIP library
raises.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import capi, hipgraph
from .capi import call, ptr
from .plan import (BN_EPS, FeaturePlan, ModelSpec, OPT_SLOT_ODD, adam_pow_names, bucket_geometry, ftrl_l2_shrinkage, ftrl_lr_power, opt_params,
                   opt_slot_init, opt_slot_names, rmsprop_centered, small_table_slots)


class DeviceBatch:
    """One batch, resident in HBM, in the example-major bag-CSR layout (include/wd_hip.h)."""

    def __init__(self, B, ids, bag_offs, dense=None, labels=None, weights=None, nnz=None, one_hot=False):
        self.one_hot = bool(one_hot)   # every bag holds exactly one id (bag_offs == 0,1,2,...): kernels may skip the CSR
        self.B = int(B)
        self.ids = ids              # int32 [nnz]
        self.bag_offs = bag_offs    # int32 [B*S + 1]
        self.dense = dense          # float32 [B, n_dense] or None
        self.labels = labels        # float32 [B] or None
        self.weights = weights      # float32 [B] or None
        self.nnz = int(nnz if nnz is not None else ids.numel())
        self.ids_cols = None        # optional slot-major copy of a one-id-per-bag batch: ids_cols[s * B + b] (wd_hash_bucket_cols)
        self.ids_cols_valid = False   # set by whoever fills ids_cols together with ids (synth.hash_tokens); a caller that writes
                                      # `ids` itself leaves it False and the bucketing reads `ids`


def _stream():
    return torch.cuda.current_stream().cuda_stream


class WideDeepEngine:
    def __init__(self, spec: ModelSpec, max_batch=8192, max_nnz=None, device="cuda", seed=0, expected_nnz=None,
                 tower_dtype="fp32", table_seed=None, row_records=None):
        if not torch.cuda.is_available():
            raise capi.WdError("WideDeepEngine needs a GPU (MI355X / gfx950); there is no CPU fallback")
        capi.load()
        # the engine's OWN copy of the spec: set_learning_rates rewrites the optimizer tuples the launches read their rates from,
        # and a second engine built from the caller's spec must still see the initial ones
        import dataclasses
        spec = dataclasses.replace(spec)
        self.lr0 = {"dnn": float(spec.dnn_opt[1]) if spec.has_deep else None, "linear": float(spec.lin_opt[1]) if spec.has_wide else None}
        self.dropout = float(spec.dropout) if (spec.dropout and spec.has_deep) else 0.0
        if not 0.0 <= self.dropout < 1.0:
            raise ValueError("dnn_dropout must be in [0, 1), got %r" % (spec.dropout,))
        if self.dropout and tower_dtype == "fp16":
            raise NotImplementedError("tower_dtype='fp16': dnn_dropout runs on the fp32 tower")
        for opt in ([spec.dnn_opt] if spec.has_deep else []) + ([spec.lin_opt] if spec.has_wide else []):
            if opt[0] not in capi.WD_OPT_KINDS:
                raise ValueError("unsupported optimizer %r (supported: %s)" % (opt, sorted(capi.WD_OPT_KINDS)))
        # the reference's defaults (conf/model.yaml: Adagrad on the dnn scope, Ftrl on the linear scope) take the
        # specialised kernels; any other tf.train optimizer of model_util.py:84-90 the generic ones (wd_opt_t)
        self.default_opts = ((not spec.has_deep or spec.dnn_opt[0] == "Adagrad") and
                             (not spec.has_wide or (spec.lin_opt[0] == "Ftrl" and ftrl_lr_power(spec.lin_opt) == -0.5
                                                    and ftrl_l2_shrinkage(spec.lin_opt) == 0.0)))
        if tower_dtype not in ("fp32", "fp16"):
            raise ValueError("tower_dtype must be 'fp32' (exact fp32 MFMA) or 'fp16' (half operands, fp32 accumulate)")
        self.half = tower_dtype == "fp16"     # BASELINE configs[4]: fp16 MFMA dense path, fp32 embeddings
        self.spec = spec
        self.plan = plan = FeaturePlan(spec)
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        self.max_nnz = int(max_nnz) if max_nnz else self.max_batch * max(plan.S, 1) * 8
        self.inv = 1.0 / math.sqrt(1.0 + BN_EPS)
        self.crelu = plan.crelu      # a relu layer of twice the width with tied halves (plan.py)
        # Row-record layout (Criteo-shaped models: every categorical column is embedded with ONE width <= 16 AND feeds the
        # linear part, default optimizers, one GPU): the embedding row and the wide line {w, z, n, -} of a fused row share
        # ONE 32 / 64 / 128-byte record, so that the forward finds the wide weight in the line it fetched for the row and the
        # update touches two random lines per row (record + accumulator) instead of three (profiles/r2z_layouts.txt).
        # row_records=None: on when eligible (WD_ROW_RECORDS=0 turns it off); True: required; False: separate tables.
        # Round 6: the small tables of csrc/small_tables.hip (crossed columns) may be narrower than the record -- they sit in records
        # of the big columns' width ([emb d <= D | pad | w z n - | pad]; the small-table kernels take the record stride), so a model
        # like BASELINE configs[3] (26 columns of width 16 + two crossed columns of width 4) is record-shaped too: its multi-hot
        # gather and its row update touch one line less per row (profiles/r2z_layouts.txt: 3 tables -> 2, -17 % on the update).
        forced = getattr(self, "_small_forced", None)
        small0 = (list(forced) if forced is not None else
                  small_table_slots(plan.slots, spec.has_deep, spec.has_wide, capi.SMALL_MAX_FLOATS, os.environ.get("WD_SMALL_TABLES", "cross"))
                  if type(self) is WideDeepEngine else [])
        big = [i for i in range(plan.S) if i not in small0]
        dims = sorted({int(plan.slots[i].dim) for i in big if plan.slots[i].deep == "embedding"}) if spec.has_deep else []
        # (the row-sharded engine lays its LOCAL rows out the same way: dist.py sets _records_ok before it gets here)
        eligible = (spec.has_deep and spec.has_wide and self.default_opts
                    and getattr(self, "_records_ok", type(self) is WideDeepEngine)
                    and len(dims) == 1 and dims[0] in (4, 8, 16) and plan.S > 0 and len(big) > 0
                    and all(plan.slots[i].deep == "embedding" and plan.slots[i].wide for i in big)
                    and all(plan.slots[i].wide and (plan.slots[i].deep in ("embedding", None))
                            and (plan.slots[i].deep is None or int(plan.slots[i].dim) <= dims[0]) for i in small0)
                    and (not small0 or os.environ.get("WD_SMALL_RECORDS", "1") != "0"))
        if row_records and not eligible:
            raise ValueError("row_records=True: the model is not record-shaped (one embedding width in {4, 8, 16} on every "
                             "categorical column, all of them wide columns too, Adagrad + Ftrl)")
        if row_records is None:
            row_records = os.environ.get("WD_ROW_RECORDS", "1") != "0"
        self.rec_stride = {4: 8, 8: 16, 16: 32}[dims[0]] if (row_records and eligible) else 0
        self.rec = None
        self.flat_ragged = os.environ.get("WD_FLAT_RAGGED", "1") != "0"       # multi-hot batches on records: sort ahead, flat update
        self.act_id = capi.ACT_IDS["relu" if self.crelu else spec.activation]
        self.global_step = 0
        self._primed = None             # pipeline.StepGraph: (batch, bucket set, activation buffer, global step) whose input work is in place
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)

        # ---- slot descriptors --------------------------------------------------------------
        S = plan.S
        # bucket geometry of the fused sparse backward (wd_sparse_bucketize): ~64 occurrences per row-range bucket,
        # one bucket per row for vocabularies too small to be cut (plan.bucket_geometry)
        exp_nnz = int(expected_nnz) if expected_nnz else self.max_batch * max(S, 1)
        # ~64 occurrences per bucket at BASELINE batch sizes; a small batch (the reference's own 64-512: conf/train.yaml:47,
        # BASELINE configs[0]) gets smaller buckets -- enough of them to fill the chip (k_bucket_update is one workgroup per bucket:
        # at batch 512 of the shipped conf 423 buckets, the slowest holding six rows with > 32 occurrences each, reduced one after the
        # other by its workgroup, took 140 us for 50 k occurrences; profiles/r6_c1_kernel_stats_b512_before.md)
        target = float(os.environ.get("WD_BUCKET_TARGET", "0")) or min(64.0, max(8.0, exp_nnz / 2048.0))
        shifts, bases, self.n_buckets = bucket_geometry(
            [s.num_buckets for s in plan.slots], exp_nnz / max(S, 1), int(call("wd_bucket_max")), target)
        self.bucket_shifts = shifts
        arr = (capi.WdSlot * max(S, 1))()
        for i, s in enumerate(plan.slots):
            is_emb = s.deep == "embedding" and spec.has_deep
            is_ind = s.deep == "indicator" and spec.has_deep
            arr[i].emb_off = plan.emb_off[i]
            arr[i].row_base = plan.row_base[i]
            arr[i].num_buckets = int(s.num_buckets)
            arr[i].dim = int(s.dim) if is_emb else 0
            arr[i].out_col = plan.out_col[i]
            arr[i].kind = capi.SLOT_EMBEDDING if is_emb else (capi.SLOT_INDICATOR if is_ind else capi.SLOT_NONE)
            arr[i].wide = 1 if (spec.has_wide and s.wide) else 0
            arr[i].bucket_shift, arr[i].bucket_base = shifts[i], bases[i]
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self.slots_dev = torch.from_numpy(raw).to(dev)
        self.group_slots = {d: torch.tensor(v, **i32) for d, v in plan.emb_groups.items()}
        # Small tables with long bags (csrc/small_tables.hip): crossed columns (python/lib/build_estimator.py:138-155 -- a bag holds
        # the PRODUCT of its keys' counts, hashed into a few hundred buckets) whose table + one partial record per row fit in
        # LDS.  Multi-hot batches take them out of the general path (wd_wide_fwd / the embedding-bag groups / the bucketed
        # update skip a slot whose descriptor carries WD_SLOT_F_SMALL) and through wd_small_tables_fwd / _bwd.  Separate tables,
        # the reference's default optimizers, one GPU.  WD_SMALL_TABLES=0: off; =all: every column that fits, crossed or not.
        # (row-sharded ranks, dist.py: these columns are REPLICATED instead of exchanged -- the set is fixed before the local
        # model exists, `_small_forced`)
        self.small_idx = small0
        self.slots_small_dev = self.small_idx_dev = self.small_ws = None
        if self.small_idx:
            for i in self.small_idx:
                arr[i].flags = capi.SLOT_F_SMALL
            self.slots_small_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
            for i in self.small_idx:
                arr[i].flags = 0
            self.small_idx_dev = torch.tensor(self.small_idx, **i32)
            self.small_ws = None      # partial sums of wd_small_tables_bwd, allocated by the first (eager) step that needs them
            self.small_rows = max(int(plan.slots[i].num_buckets) for i in self.small_idx)
            self.small_dim = max(int(arr[i].dim) for i in self.small_idx)
            # embedding-bag groups without the small columns (a group may lose its contiguity: the slot-list kernel takes it)
            self.group_lists_big = {d: [i for i in v if i not in self.small_idx] for d, v in plan.emb_groups.items()
                                    if any(i not in self.small_idx for i in v)}
            self.group_slots_big = {d: torch.tensor(v, **i32) for d, v in self.group_lists_big.items()}
        self.ind_slots_dev = torch.tensor(plan.ind_slots, **i32) if plan.ind_slots else None
        if plan.dense_cols:
            darr = (capi.WdDenseCol * len(plan.dense_cols))()
            for j, d in enumerate(plan.dense_cols):
                darr[j].p0, darr[j].p1, darr[j].kind, darr[j].out_col = d.p0, d.p1, d.kind, plan.dense_out_col[j]
            self.dense_cols_dev = torch.from_numpy(np.frombuffer(bytes(darr), dtype=np.uint8).copy()).to(dev)
        else:
            self.dense_cols_dev = None

        # ---- sparse state ------------------------------------------------------------------
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        self.gen = g
        # table_seed (row-sharded ranks): the embedding tables and the dropout stream of THIS rank draw from their own seed
        # -- with the shared one, local row j would start identical on every rank -- the dense parameters keep `seed`
        gt = g
        if table_seed is not None:
            gt = torch.Generator(device=dev)
            gt.manual_seed(int(table_seed))
        dseed = int(seed if table_seed is None else table_seed)
        # dropout: device {seed, step}; the keep mask is a function of it (include/wd_hip.h), never stored
        self.drop_seed = torch.tensor([dseed * 0x9E3779B1 + 12345, 0], dtype=torch.int64, device=dev) if self.dropout else None
        dnn_a, dnn_b = opt_slot_init(spec.dnn_opt) if spec.has_deep else (None, None)
        lin_a, lin_b = opt_slot_init(spec.lin_opt) if spec.has_wide else (None, None)
        if self.rec_stride:
            # records [total_rows][rec_stride]; self.emb / self.wide are strided VIEWS of it ([rows, D] / [rows, 4])
            self.rec = torch.zeros(max(plan.total_rows, 1), self.rec_stride, **f32)
            rarr = (capi.WdSlot * S)()
            ctypes.memmove(rarr, arr, ctypes.sizeof(rarr))
            for i in range(S):
                rarr[i].emb_off = plan.row_base[i] * self.rec_stride       # row of id = rec + emb_off + id * rec_stride
            self.rslots_dev = torch.from_numpy(np.frombuffer(bytes(rarr), dtype=np.uint8).copy()).to(dev)
        if spec.has_deep:
            ne = max(plan.emb_elems, 4)
            self.emb = torch.zeros(ne, **f32) if self.rec is None else self.rec[:, : dims[0]]
            self.emb_a = torch.full((ne,), dnn_a, **f32) if dnn_a is not None else None      # optimizer slot a
            self.emb_acc = torch.full((ne,), dnn_b, **f32) if dnn_b is not None else None    # optimizer slot b
            self.emb_c = torch.zeros(ne, **f32) if rmsprop_centered(spec.dnn_opt) else None  # slot c (mean gradient)
            for i, s in enumerate(plan.slots):
                if plan.emb_off[i] >= 0:
                    std = 1.0 / math.sqrt(s.dim)
                    # embedding_column initializer: truncated_normal(0, 1/sqrt(dim))  (SURVEY App. A.6); drawn into a
                    # contiguous buffer so that both table layouts start from the same numbers
                    v = torch.empty(s.num_buckets, s.dim, **f32)
                    torch.nn.init.trunc_normal_(v, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gt)
                    self._emb_view(self.emb, i).copy_(v)
                    del v
        else:
            self.emb = self.emb_a = self.emb_acc = self.emb_c = None
        if spec.has_wide:
            # {w, slot a, slot b, slot c}  (Ftrl: {w, z, n, -}; slot c = centered RMSProp's mean gradient, starts at 0)
            self.wide = (torch.zeros(max(plan.total_rows, 1), 4, **f32) if self.rec is None
                         else self.rec[:, dims[0]: dims[0] + 4])
            self.bias = torch.zeros(4, **f32)
            for col, v in ((1, lin_a), (2, lin_b)):
                if v is not None:
                    self.wide[:, col] = v
                    self.bias[col] = v
        else:
            self.wide = self.bias = None
        # generic optimizer descriptors (wd_opt_t) + Adam's beta powers (device: a captured graph must see them advance)
        self.pow_names = adam_pow_names(spec.dnn_opt, spec.lin_opt, spec.has_deep, spec.has_wide)
        self.pow, self.opt_c = {}, {}
        for scope, opt, on in (("dnn", spec.dnn_opt, spec.has_deep), ("linear", spec.lin_opt, spec.has_wide)):
            if not on:
                continue
            o = capi.WdOpt()
            o.kind, o.lr = capi.WD_OPT_KINDS[opt[0]], float(opt[1])
            if rmsprop_centered(opt):
                o.kind = capi.WD_OPT_RMSPROP_CENTERED
                if scope == "dnn":
                    o.slot_c = self.emb_c.data_ptr()
            o.p0, o.p1, o.p2, o.p3 = opt_params(opt)
            if opt[0] == "Adam":
                self.pow[scope] = torch.tensor([opt[2], opt[3]], **f32)
                o.pow = self.pow[scope].data_ptr()
            self.opt_c[scope] = o
        self.touched = torch.zeros((max(plan.total_rows, 1) + 31) // 32, **i32) if self.pow else None

        # ---- dense state -------------------------------------------------------------------
        B = self.max_batch
        self.towers = []
        if spec.has_deep:
            n = plan.dense_param_elems
            self.P = torch.zeros(n, **f32)
            self.Pa = torch.full((n,), dnn_a, **f32) if dnn_a is not None else None       # optimizer slot a
            self.Pacc = torch.full((n,), dnn_b, **f32) if dnn_b is not None else None     # optimizer slot b
            self.Pc = torch.zeros(n, **f32) if rmsprop_centered(spec.dnn_opt) else None   # slot c
            # the dnn-scope optimizer as applied to the flat dense buffer (its slot c is Pc, the tables' is emb_c)
            od = capi.WdOpt()
            ctypes.memmove(ctypes.byref(od), ctypes.byref(self.opt_c["dnn"]), ctypes.sizeof(od))
            od.slot_c = self.Pc.data_ptr() if self.Pc is not None else None
            self.opt_c["dnn_dense"] = od
            self.G = torch.zeros(n, **f32)
            for ti, tl in enumerate(plan.towers):
                metas = plan.layer_meta[ti]
                L = len(tl.hidden)
                tw = {"layout": tl, "metas": metas, "L": L}
                tw["act"] = torch.zeros(B, tl.ld, **f32)
                tw["dact"] = torch.zeros(B, tl.ld, **f32)
                maxN = max([m["N"] for m in metas] + [metas[L]["K"]])
                tw["dz"] = [torch.zeros(B * maxN, **f32), torch.zeros(B * maxN, **f32)]   # ping-pong (fused act')
                tw["logit"] = torch.zeros(B, **f32)
                tw["Wf"], tw["bf"], tw["s"], tw["t"], tw["gidx"], tw["bidx"], tw["nsplit"] = [], [], [], [], [], [], []
                tw["Gpart"] = []
                head_blocks = int(call("wd_logits_head_blocks", B, metas[L]["K"]))
                for l, m in enumerate(metas):
                    K, N = m["K"], m["N"]
                    tw["Wf"].append(torch.zeros(K * N, **f32))
                    tw["bf"].append(torch.zeros(capi.WD_FOLD_PARTS * N, **f32))
                    tw["s"].append(torch.zeros(K, **f32))
                    tw["t"].append(torch.zeros(K, **f32))
                    tw["gidx"].append(torch.from_numpy(m["gamma_idx"]).to(dev))
                    tw["bidx"].append(torch.from_numpy(m["beta_idx"]).to(dev))
                    if l == L:     # logits layer: partials come from wd_logits_head, one per 64-example block
                        ns = head_blocks
                    else:          # split-K over the batch: ~512 workgroups, >= 4 reduction slabs (of 64) each
                        tsz = 128 if (self.half and N > 128) else 64
                        tiles = math.ceil((K + 1) / tsz) * math.ceil(N / tsz)
                        ns = max(1, min(math.ceil(512 / tiles), 64, max(1, B // 256)))
                        if not self.half:    # fp32 towers: <= 16 partials per element for the finalize / dense tail to sum
                            ns = min(ns, int(os.environ.get("WD_TN_SPLIT_CAP", "16")))
                    tw["nsplit"].append(ns)
                    tw["Gpart"].append(torch.zeros(ns * (K + 1) * N, **f32))
                    # tf.glorot_uniform_initializer kernel, zero bias, gamma 1, beta 0  
                    Ktf, Ntf = len(plan.tf_rows_of_layer(ti, l)), m["N_tf"]
                    lim = math.sqrt(6.0 / (Ktf + Ntf))
                    W = self.P[m["w_off"]: m["w_off"] + K * N].view(K, N)
                    rows = torch.from_numpy(plan.tf_rows_of_layer(ti, l)).to(dev)
                    Wtf = (torch.rand(Ktf, Ntf, generator=g, **f32) * 2 - 1) * lim
                    W[rows, :Ntf] = Wtf
                    if Ntf != N:       # crelu: mirrored half
                        W[rows, Ntf:] = -Wtf
                    if "gamma_off" in m:
                        self.P[m["gamma_off"]: m["gamma_off"] + N] = 1.0
                if self.half:
                    if len(plan.towers) != 1:
                        raise NotImplementedError("tower_dtype='fp16': one tower")
                    if tl.copies:
                        raise NotImplementedError("tower_dtype='fp16': connected_mode first_dense / connection lists run on "
                                                  "the fp32 tower")
                    f16 = dict(dtype=torch.float16, device=dev)
                    r8 = lambda v: (v + 7) // 8 * 8

                    def hzeros(*shape):
                        """half buffer with 64 halfs of slack behind its last row: the GEMM loaders issue unconditional
                        16-byte loads at addresses clamped to the last vector of a row (csrc/mlp_half.hip)"""
                        n = int(np.prod(shape))
                        return torch.zeros(n + 64, **f16)[:n].view(*shape)

                    Bp = (B + 63) // 64 * 64
                    tw["Bp"] = Bp
                    tw["act_h"] = hzeros(B, tl.ld)            # activations, example-major
                    tw["actT_h"] = hzeros(tl.ld, Bp)          # and column-major (batch contiguous) for TN
                    tw["WfT_h"] = [hzeros(metas[l]["N"], r8(metas[l]["K"])) for l in range(L)]
                    # Z = [dlogit | dz_{L-1} | ... | dz_0] (blocks 8-aligned), example-major and transposed.  The gradient
                    # of segment j is PULLED in one GEMM  Z[:, zbeg_j:zend_j] . Wcat_j^T  over all its consumer layers
                    # (dense / resnet: every later layer and the logits), so nothing is accumulated in HBM.
                    zoff, z = {L: 0}, 8
                    for l in range(L - 1, -1, -1):
                        zoff[l] = z
                        z += r8(metas[l]["N"])
                    tw["zoff"], tw["pZ"] = zoff, z
                    tw["Z_h"] = hzeros(B, z)
                    tw["ZT_h"] = hzeros(z, Bp)
                    pulls, base = {}, 0
                    for j in range(L + 1):
                        cons = [c for c in range(L + 1) if j in tl.in_segs[c]]
                        zb = min(zoff[c] for c in cons)
                        ze = max(zoff[c] + metas[c]["N"] for c in cons)
                        pitch = r8(ze - zb)
                        pulls[j] = dict(zbeg=zb, kred=ze - zb, pitch=pitch, base=base, width=tl.seg_width[j])
                        base += tl.seg_width[j] * pitch
                        base = r8(base)
                    tw["pulls"] = pulls
                    tw["wcat"] = hzeros(max(base, 8))
                    tw["cat_off"] = []
                    for c in range(L + 1):
                        co = np.full(metas[c]["K"], -1, dtype=np.int64)
                        for k, (seg, u) in enumerate(tl.window_cols(c)):
                            if seg >= 0:
                                pj = pulls[seg]
                                co[k] = pj["base"] + u * pj["pitch"] + (zoff[c] - pj["zbeg"])
                        tw["cat_off"].append(torch.from_numpy(co).to(dev))
                self.towers.append(tw)
            self._setup_chain()
            # descriptor table of every layer of every tower (wd_fold_affine_all / wd_mlp_finalize_all)
            nl = sum(len(tw["metas"]) for tw in self.towers)
            larr = (capi.WdMlpLayer * nl)()
            i = 0
            self.max_layer_n = self.max_layer_k = 1
            for tw in self.towers:
                for l, m in enumerate(tw["metas"]):
                    d = larr[i]
                    d.w_off, d.b_off, d.K, d.N = m["w_off"], m["b_off"], m["K"], m["N"]
                    d.gamma_idx, d.beta_idx = tw["gidx"][l].data_ptr(), tw["bidx"][l].data_ptr()
                    d.Wf, d.bf, d.s, d.t = (tw["Wf"][l].data_ptr(), tw["bf"][l].data_ptr(), tw["s"][l].data_ptr(),
                                            tw["t"][l].data_ptr())
                    d.Gpart, d.nsplit = tw["Gpart"][l].data_ptr(), tw["nsplit"][l]
                    if self.half:
                        d.cat_off, d.wcat = tw["cat_off"][l].data_ptr(), tw["wcat"].data_ptr()
                        if l < tw["L"]:
                            d.WfT_h, d.ld_wft_h = tw["WfT_h"][l].data_ptr(), tw["WfT_h"][l].shape[1]
                    self.max_layer_n = max(self.max_layer_n, m["N"])
                    self.max_layer_k = max(self.max_layer_k, m["K"])
                    i += 1
            self.n_layers = nl
            self.layers_dev = torch.from_numpy(np.frombuffer(bytes(larr), dtype=np.uint8).copy()).to(dev)
            # one launch finalises all layers iff every BN gamma/beta has a single consumer layer
            self.all_simple = len(self.towers) == 1 and self.towers[0]["layout"].mode == "simple"
            self.dnn_logit = torch.zeros(B, **f32)
        else:
            self.P = self.Pa = self.Pacc = self.Pc = self.G = None
            self.dnn_logit = None

        # ---- per-step buffers --------------------------------------------------------------
        self.wide_logit = torch.zeros(B, **f32) if spec.has_wide else None
        self.logit = torch.zeros(B, **f32)
        self.prob = torch.zeros(B, **f32)
        self.dlogit = torch.zeros(B, **f32)
        self.loss = torch.zeros(1, **f32)
        M = self.max_nnz
        self.keys = torch.zeros(M, dtype=torch.int32, device=dev)
        self.vals = torch.zeros(M, **i32)
        self.keys_sorted = torch.zeros(M, dtype=torch.int32, device=dev)
        self.vals_sorted = torch.zeros(M, **i32)
        # rocPRIM picks its algorithm by size: take the max over the sizes we may be called with
        sizes, m = [], M
        while m >= 1:
            sizes.append(m)
            m //= 2
        qs = [int(call("wd_sort_workspace_bytes", n, plan.key_bits)) for n in sizes]
        if min(qs) == 0:
            raise capi.WdError("wd_sort_workspace_bytes failed")
        self.sort_ws_bytes = max(qs)
        self.sort_ws = torch.zeros(self.sort_ws_bytes, dtype=torch.uint8, device=dev)
        # three sets of bucketing scratch: in a pipelined multi-step graph (pipeline.StepGraph) the occurrences of step t+1
        # are bucketed and sorted on a branch of their own while the updates of steps t-1 and t still read their sets
        self._bucket_sets = []
        self.max_slot_buckets = max([((int(sl.num_buckets) + (1 << sh) - 1) >> sh) for sl, sh in zip(plan.slots, shifts)] + [1])
        for _ in range(3):
            self._bucket_sets.append(dict(
                ticket=torch.zeros(4, **i32), unsorted=False,      # wd_bucket_onehot: last-workgroup ticket; arrival-order pairs
                sorted=False,                                      # pairs sorted in place by wd_bucket_sort (flat row update)
                long_list=torch.zeros(2 * (M // 32 + 2) + 2, **i32), big_list=torch.zeros(self.n_buckets, **i32),
                patch=torch.zeros(2 * M, **i32),
                cnt=torch.zeros((2 * int(call("wd_bucket_chunks")) + 1) * self.n_buckets, **i32),
                start=torch.zeros(2 * self.n_buckets + 2, **i32),   # starts [nb+1] + launch order [nb]
                rank=torch.zeros(M, **i32), pairs=torch.zeros(M, dtype=torch.int64, device=dev)))
        b0 = self._bucket_sets[0]
        self.bucket_cnt, self.bucket_start, self.occ_rank, self.pairs = b0["cnt"], b0["start"], b0["rank"], b0["pairs"]
        self._folded = False      # one-launch tower on one GPU: the fold for the NEXT step is launched behind the dense update
        self._graph = None
        # Optional side streams (hipGraph capture turns them into parallel branches):
        #   side 0: bucketing of the occurrences (needs only the ids) under the forward + tower backward
        #   side 1: the split-K weight-gradient GEMM of a layer beside the next input-gradient GEMM
        # Measured on MI355X (ROCm 7.2): every cross-stream edge costs more than it hides at C2's kernel sizes
        # (0.305 ms/step single-stream, 0.317 with the bucketing branch, 0.34 with the TN branch), while the fp16
        # tower at C5 gains 4 % from the TN branch.  Defaults follow the measurements; WD_OVERLAP=none|bucket|tn|both.
        # With the one-launch tower (wd_tower_chain: one wavefront per SIMD for ~75 us, 118 KB of LDS) the bucketing
        # kernels co-reside on the same CUs: 0.2425 -> 0.232 ms/step at C2, so that branch is on by default there.
        # Per-layer towers on big ragged batches (C4: 1.06 M occurrences, bucketing 84 us; C5): both branches, 0.7414 -> 0.7157
        # and 1.0227 -> 1.0121 ms/step (profiles/r4_overlap_c4_c5.txt); small batches keep one stream.
        big = self.max_batch * max(1, self.plan.S) >= (1 << 16)
        mode = os.environ.get("WD_OVERLAP", "bucket" if getattr(self, "chain", False) else
                              ("both" if big else ("tn" if self.half else "none")))
        self.overlap_bucket = mode in ("both", "bucket")
        self.overlap_tn = mode in ("both", "tn")
        self._sides = None
        # sparse forward in one launch when the model has exactly one embedding dim group on a contiguous slot range
        # and no indicator columns (all Criteo-shaped configs); otherwise one launch per piece
        self._fused_input_layer = False
        if spec.has_deep and len(plan.emb_groups) == 1 and not plan.ind_slots:
            (gdim, gsl), = plan.emb_groups.items()
            self._fused_input_layer = (gsl == list(range(gsl[0], gsl[0] + len(gsl))) and gdim in (4, 8, 16, 32, 64, 128)
                                       and len(gsl) <= 128)
        if spec.has_deep:
            self._setup_prefetch()

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def _setup_chain(self):
        """One-launch tower (wd_tower_chain, csrc/mlp_chain.hip): exact-fp32 `simple` towers whose widths are multiples
        of 32 and whose row tile fits the LDS.  WD_CHAIN=0 keeps the per-layer GEMM launches."""
        self.chain = False
        self.prefetch, self._apar, self._prefetched = False, 0, False
        self.flat_update = False
        self._prefetch_span = None       # diagnostics: device pointer of uint64[n_act][prefetch_blocks][2] {start, end} of every workgroup of wd_prefetch_onehot
        self._chain_tile_stamps = None   # diagnostics: device uint64[2 * tiles] realtime-clock stamps (bench.py: in-step gather span)
        self._chain_stamps = None    # diagnostics: device int64[64] for the tower kernel's stage stamps (scripts/bench_chain.py)
        plan = self.plan
        if self.half or self.dropout or self.crelu or len(self.towers) != 1 or os.environ.get("WD_CHAIN", "1") == "0":
            return
        tw = self.towers[0]
        tl, metas, L = tw["layout"], tw["metas"], tw["L"]
        self.chain_rt = 32           # examples per workgroup of the one-launch tower (csrc/mlp_chain8.hip)
        # concatenating towers (python/lib/dnn.py:155-193: 'dense', 'resnet') take the same launch: the row tile in LDS mirrors the
        # activation row, whose segment order already makes every layer's input one contiguous window (plan.TowerLayout)
        windows = (tl.mode in ("dense", "resnet") and not tl.copies and self._chain_windows_ok()
                   and os.environ.get("WD_CHAIN_WINDOWS", "1") != "0")
        if (tl.mode != "simple" and not windows) or L < 1 or L > capi.WD_CHAIN_MAX_LAYERS or tl.in_start[0] % 4 or tl.ld % 4:
            return
        dims = [int(metas[l]["N"]) for l in range(L)]
        K0 = int(metas[0]["K"])
        rt = 32
        tw["windows"] = None
        if windows:
            w = capi.WdChainWindows()
            for sg in range(L + 1):
                w.seg_col[sg], w.in_col[sg] = int(tl.seg_start[sg]), int(tl.in_start[sg])
            w.k_logits, w.cols = int(metas[L]["K"]), (int(tl.ld) + 31) // 32 * 32
            if int(call("wd_tower_chain_windows_lds_bytes", ctypes.byref(w), K0, (ctypes.c_int32 * L)(*dims), L)) <= 0:
                return
            tw["windows"] = w
        elif int(call("wd_tower_chain_lds_bytes", K0, (ctypes.c_int32 * L)(*dims), L, rt)) <= 0:
            return
        dev, B = self.device, self.max_batch
        f32 = dict(dtype=torch.float32, device=dev)
        # the kernels in MFMA-fragment order, forward and transposed (wd_chain_layer_t.Wpk / WTpk: written by wd_chain_tail)
        tw["Wpk"] = [torch.zeros(metas[l]["N"] * metas[l]["K"], **f32) for l in range(L)]
        if windows:
            # the PULL operand of segment l (x for l = 0, else hidden layer l - 1's output): [N_l + .. + N_{L-1}] x [segment width]
            tw["WTpk"] = [torch.zeros(sum(dims[l:]) * int(tl.seg_width[l]), **f32) for l in range(L)]
        else:
            tw["WTpk"] = [torch.zeros(metas[l]["N"] * metas[l]["K"], **f32) for l in range(L)]
        tw["dzl"] = [torch.zeros(B * metas[l]["N"], **f32) for l in range(L)]
        # bias / BN gradients: per-row-tile column sums from the tower kernel (dz; d(bn) * a; d(bn)), reduced by column-sum jobs
        # of the grouped weight-gradient launch -- the products need no appended ones row (449 = 7 x 64 + 1 rows cost an 8th
        # tile row) and the dense tail no row reductions
        ntile = int(call("wd_tower_chain_blocks", B, rt))
        for nm in ("db", "dg", "dbeta"):
            tw[nm + "_part"] = [torch.zeros(ntile * metas[l]["N"], **f32) for l in range(L)]
            tw[nm + "_sum"] = [torch.zeros(metas[l]["N"], **f32) for l in range(L)]
        # logits-layer gradient partials: one per row tile
        ns = ntile
        tw["nsplit"][L] = ns
        tw["Gpart"][L] = torch.zeros(ns * (metas[L]["K"] + 1) * metas[L]["N"], **f32)
        # ... summed over the tiles by a column-sum job of the grouped weight-gradient launch: the dense tail reads ONE value per
        # element of the logits layer (summing the 256 per-tile partials itself was 16 dependent round trips of its 65 lanes --
        # the long pole of the launch that sits between the products and the next tower)
        tw["Glog_sum"] = torch.zeros((metas[L]["K"] + 1) * metas[L]["N"], **f32)
        pbase = self.P.data_ptr()
        carr = (capi.WdChainLayer * L)()
        tarr = (capi.WdTailLayer * (L + 1))()
        for l in range(L + 1):
            m, t = metas[l], tarr[l]
            t.w_off, t.b_off, t.K, t.N = m["w_off"], m["b_off"], int(m["K"]), int(m["N"])
            t.gamma_off = m["gamma_off"] if "gamma_off" in m else -1
            t.beta_off = m["beta_off"] if "beta_off" in m else -1
            t.Gpart, t.nsplit, t.pk_tile = tw["Gpart"][l].data_ptr(), tw["nsplit"][l], rt
            if l == L:
                t.Gpart, t.nsplit = tw["Glog_sum"].data_ptr(), 1
                continue
            bn = "gamma_off" in m
            t.db_sum = tw["db_sum"][l].data_ptr()
            t.dgamma_sum, t.dbeta_sum = (tw["dg_sum"][l].data_ptr(), tw["dbeta_sum"][l].data_ptr()) if bn else (None, None)
            t.Wpk, t.WTpk = tw["Wpk"][l].data_ptr(), tw["WTpk"][l].data_ptr()
            if windows:      # rows of this layer's kernel by segment of its window -> that segment's pull operand
                segs = list(tl.in_segs[l])
                t.nseg = len(segs)
                for q, sg in enumerate(segs):
                    t.seg_k0[q] = int(tl.seg_start[sg] - tl.in_start[l])
                    t.seg_w[q] = int(tl.seg_width[sg])
                    t.seg_red0[q] = sum(dims[sg:l])          # the dz tiles lie in ascending layer order from dz_sg on
                    t.seg_kred[q] = sum(dims[sg:])
                    t.seg_wt[q] = tw["WTpk"][sg].data_ptr()
            c = carr[l]
            c.Wpk, c.WTpk, c.bias = tw["Wpk"][l].data_ptr(), tw["WTpk"][l].data_ptr(), pbase + 4 * m["b_off"]
            c.gamma, c.beta = (pbase + 4 * m["gamma_off"], pbase + 4 * m["beta_off"]) if bn else (None, None)
            c.a_out = tw["act"].data_ptr() + 4 * tl.seg_start[l + 1]
            c.dz_out = tw["dzl"][l].data_ptr()
            c.db_part = tw["db_part"][l].data_ptr()
            c.dgamma_part, c.dbeta_part = (tw["dg_part"][l].data_ptr(), tw["dbeta_part"][l].data_ptr()) if bn else (None, None)
            c.K, c.N = int(m["K"]), dims[l]
        tw["chain_layers"], tw["tail_layers"] = carr, tarr
        tw["acts"], tw["chain_layers_p"] = [tw["act"]], [carr]
        self.prefetch, self._apar, self._prefetched = False, 0, False
        self.loss_part = torch.zeros(ntile, **f32)   # per-row-tile losses, summed in tile order by the grouped launch
        # gradient columns of x that anyone reads: the embedding columns (the sparse backward), rounded up by the kernel
        emb_cols = 0
        for i, sl in enumerate(plan.slots):
            if plan.emb_off[i] >= 0:
                emb_cols = max(emb_cols, plan.out_col[i] + int(sl.dim))
        tw["dx_cols"] = min(emb_cols, K0)
        self.chain = True

    def _chain_windows_ok(self):
        """Concatenating towers in the one launch (dist.py: the row-sharded engine takes them too)."""
        return type(self) is WideDeepEngine

    def _setup_prefetch(self):
        """Prefetched input layer (one-id-per-bag batches on row records, single GPU): x and the wide weights of a batch are
        gathered by their own launch (wd_prefetch_onehot) -- in a pipelined multi-step graph one step AHEAD, beside the
        previous tower, into the other of two activation buffers; the rows the update in between rewrites are patched by
        that update (wd_apply_next_t).  WD_INPUT_AHEAD=0: the tower kernel gathers its own x tile (wd_chain_input_t)."""
        self._apar = 0                 # activation buffer / wide-weight list the current step uses
        self._prefetched = False       # pipeline.StepGraph: x / wv of the batch forward() is called with are already in place
        self.prefetch = bool(self.spec.has_deep and self.chain and self.towers[0].get("windows") is None and self.rec is not None and type(self) is WideDeepEngine
                             and self._fused_input_layer and os.environ.get("WD_INPUT_AHEAD", "1") != "0")
        if not self.prefetch:
            return
        tw = self.towers[0]
        tl, L, B = tw["layout"], tw["L"], self.max_batch
        f32 = dict(dtype=torch.float32, device=self.device)
        carr = tw["chain_layers"]
        # THREE activation buffers / weight lists: the gather of batch t+1 (sparse branch, behind update(t-1)) overwrites the
        # buffer the products of step t-2 read -- complete before tower(t-1) even started -- so no event has to sit between the
        # products and the tail of a step to release it (one graph node less on the dense chain)
        self.n_act = 3
        for _ in range(self.n_act - 1):
            act2 = torch.zeros(B, tl.ld, **f32)
            carr2 = (capi.WdChainLayer * L)()
            ctypes.memmove(carr2, carr, ctypes.sizeof(carr))
            for l in range(L):
                carr2[l].a_out = act2.data_ptr() + 4 * tl.seg_start[l + 1]
            tw["acts"].append(act2)
            tw["chain_layers_p"].append(carr2)
        self.wv = [torch.zeros(B * self.plan.S, **f32) for _ in range(self.n_act)]
        # Flat row update: every id-only part of the sparse update (bucketing, the sort of every bucket, long segments, which
        # rows the next batch shares) runs with the bucketing -- beside the previous tower in a pipelined graph -- and the
        # update itself is a flat launch over the sorted pairs (wd_row_update).  WD_FLAT_UPDATE=0: wd_sparse_apply_rec.
        self.flat_update = os.environ.get("WD_FLAT_UPDATE", "1") != "0" and self.plan.S <= 128

    def prefetch_blocks(self, B):
        """workgroups of a wd_prefetch_onehot launch (size of its diagnostic stamp array)"""
        (dim, sl), = self.plan.emb_groups.items()
        return int(call("wd_prefetch_onehot_blocks", B, self.plan.S, dim, len(self.plan.dense_cols)))

    def _prefetch_input(self, bt, st, p):
        """Input layer of `bt` into activation buffer p: embedding rows + numeric columns -> x, wide weights -> wv[p]."""
        plan, tw = self.plan, self.towers[0]
        (dim, sl), = plan.emb_groups.items()
        nd = len(plan.dense_cols)
        call("wd_prefetch_onehot", ptr(self.rec), self.rec_stride, dim, ptr(self.rslots_dev), plan.S, ptr(bt.ids), bt.B,
             tw["acts"][p].data_ptr() + 4 * tw["layout"].seg_start[0], tw["layout"].ld, ptr(self.wv[p]),
             ptr(bt.dense) if nd else None, bt.dense.stride(0) if nd else 0, ptr(self.dense_cols_dev) if nd else None, nd,
             self._prefetch_span + 16 * p * self.prefetch_blocks(bt.B) if self._prefetch_span else None, st)

    def _fold(self, train, st):
        """One launch: fold the BN affines of every layer into its consumer's weights (+ the MFMA-fragment-packed copies
        of the one-launch tower); clears the loss accumulator of the per-layer paths (the one-launch tower stores per-tile
        partials instead) and G when the per-layer finalize path accumulates into it."""
        if self.chain:      # nothing is folded on the one-launch tower: (re)write the MFMA-packed copies of the kernels
            self._chain_tail(capi.WD_TAIL_PACK, st)
            return
        zero_g = train and not self.all_simple
        call("wd_fold_affine_all", ptr(self.P), ptr(self.layers_dev), self.n_layers, self.max_layer_n, self.inv,
             None if self.chain else ptr(self.loss), 1, ptr(self.G) if zero_g else None, self.G.numel() if zero_g else 0, st)

    def set_learning_rates(self, dnn=None, linear=None):
        """Learning rates of the two scopes for the launches that follow (eager steps read them at launch time; a captured
        graph has them baked in: capturing with spec.lr_decay set warns).  Estimator.train calls this before every step when
        spec.lr_decay is set, with rates decayed from self.lr0 -- the initial ones; self.spec (the engine's own copy) carries the
        current ones."""
        spec = self.spec
        if dnn is not None and spec.has_deep:
            spec.dnn_opt = (spec.dnn_opt[0], float(dnn)) + tuple(spec.dnn_opt[2:])
            for k in ("dnn", "dnn_dense"):
                if k in self.opt_c:
                    self.opt_c[k].lr = float(dnn)
        if linear is not None and spec.has_wide:
            spec.lin_opt = (spec.lin_opt[0], float(linear)) + tuple(spec.lin_opt[2:])
            if "linear" in self.opt_c:
                self.opt_c["linear"].lr = float(linear)

    def _chain_tail(self, mode, st):
        """wd_chain_tail: gradients from the split-K partials / Adagrad / packed kernel copies, any combination."""
        tw = self.towers[0]
        call("wd_chain_tail", tw["tail_layers"], tw["L"] + 1, ptr(self.P), ptr(self.Pacc), ptr(self.G), self.inv,
             float(self.spec.dnn_opt[1]), mode, st)

    def _fold_at_end(self):
        """The fold of step t+1 depends on nothing but the dense update of step t: launched right behind it, it runs beside
        the sparse update instead of at the head of the next step (profiles/r2d_timeline_before_pipelining.txt: 9 us + a kernel boundary)."""
        return (self.chain and type(self)._reduce_dense_grads is WideDeepEngine._reduce_dense_grads
                and type(self).backward_and_update is WideDeepEngine.backward_and_update
                and os.environ.get("WD_FOLD_AT_END", "1") == "1")

    def _chain_input_ok(self, bt):
        """The one-launch tower can build its x tile itself (input layer fused, wd_chain_opts_t.input): one id per bag,
        one embedding group, no indicator columns -- the Criteo shape.  WD_CHAIN_INPUT=0 keeps wd_input_layer_fwd."""
        if not (self.spec.has_deep and self.chain and self._fused_input_layer and bt.one_hot):
            return False
        if self.towers[0].get("windows") is not None:      # concatenating towers read x (and the wide logit) from HBM
            return False
        if os.environ.get("WD_CHAIN_INPUT", "1") == "0":
            return False
        (dim, sl), = self.plan.emb_groups.items()
        n0 = self.towers[0]["metas"][0]["N"]
        return (self.plan.S <= capi.WD_CHAIN_MAX_SLOTS and dim % 4 == 0 and 256 % (dim // 4) == 0
                and self.chain_rt * self.plan.S <= min((self.chain_rt + 1) * n0, 1024))

    def _sparse_exchange(self, bt, st):
        """Hook: what has to happen before the tower kernel can build its own x tile (dist.py: the row exchange)."""

    def _chain_input(self, bt, tw):
        """wd_chain_input_t of this batch: ids index the tables directly."""
        plan, spec = self.plan, self.spec
        (dim, sl), = plan.emb_groups.items()
        nd = len(plan.dense_cols)
        ci = capi.WdChainInput()
        ci.emb, ci.slots, ci.ids = ptr(self.emb), ptr(self.slots_dev), ptr(bt.ids)
        ci.wide = ptr(self.wide) if spec.has_wide else None
        ci.wide_bias = ptr(self.bias) if spec.has_wide else None
        ci.wide_out = ptr(self.wide_logit) if spec.has_wide else None
        ci.dense = ptr(bt.dense) if nd else None
        ci.cols = ptr(self.dense_cols_dev) if nd else None
        ci.x_out = self._x_ptr(tw)
        ci.ld_dense = bt.dense.stride(0) if nd else 0
        ci.S, ci.slot0, ci.ngroup, ci.dim, ci.ncols = plan.S, sl[0], len(sl), dim, nd
        if self.rec is not None:       # row records: the wide weight sits behind its embedding row, in the same line
            ci.emb, ci.slots, ci.row_stride = ptr(self.rec), ptr(self.rslots_dev), self.rec_stride
            if spec.has_wide:
                ci.wide, ci.wide_in_row = ptr(self.rec), 1
        return ci

    def _tower_chain(self, tw, bt, B, st, train, fuse_in=False):
        tl, metas, L = tw["layout"], tw["metas"], tw["L"]
        has_emb = bool(self.group_slots)
        need_dx = train and has_emb and tw["dx_cols"] > 0
        opts = capi.WdChainOpts()
        pf = fuse_in and self.prefetch       # x / wide weights of this batch are in activation buffer self._apar
        p = self._apar if pf else 0
        ci = self._chain_input(bt, tw) if (fuse_in and not pf) else None      # kept alive until the call returns
        if ci is not None:
            opts.input = ctypes.addressof(ci)
        if pf and self.spec.has_wide:
            opts.wide_vals, opts.wide_bias, opts.wide_out, opts.wide_S = ptr(self.wv[p]), ptr(self.bias), ptr(self.wide_logit), self.plan.S
        if train:
            opts.loss_part = ptr(self.loss_part)
        opts.stamps = self._chain_stamps
        opts.tile_stamps = self._chain_tile_stamps
        opts.row_tile = self.chain_rt
        opts.flags = int(os.environ.get("WD_CHAIN_FLAGS", "0"))
        if tw.get("windows") is not None:
            opts.windows = ctypes.addressof(tw["windows"])
        call("wd_tower_chain", tw["acts"][p].data_ptr() + 4 * tl.in_start[0], tl.ld, int(metas[0]["K"]), tw["chain_layers_p"][p], L,
             self.act_id, self.inv, self.P.data_ptr() + 4 * metas[L]["w_off"], self.P.data_ptr() + 4 * metas[L]["b_off"],
             None if fuse_in else ptr(self.wide_logit),
             ptr(bt.labels) if train else None, ptr(bt.weights) if train else None, B, ptr(tw["logit"]), ptr(self.logit),
             ptr(self.prob), ptr(self.dlogit) if train else None, ptr(self.loss) if train else None,
             ptr(tw["Gpart"][L]) if train else None,
             tw["dact"].data_ptr() + 4 * tl.in_start[0] if need_dx else None, tl.ld, tw["dx_cols"] if need_dx else 0,
             ctypes.byref(opts), st)

    def dropout_masks(self, B):
        """Host copy of the keep masks the NEXT train step will use: [tower][layer] -> float32 [B, N] of 0 / 1 (the
        function of include/wd_hip.h evaluated with numpy; used by the parity tests to drive the oracle)."""
        if not self.dropout:
            return None
        seed, step = (int(v) for v in self.drop_seed.cpu().tolist())
        M64 = (1 << 64) - 1
        out = []
        for tw in self.towers:
            per = []
            for l in range(tw["L"]):
                N = tw["metas"][l]["N"]
                idx = np.arange(B * N, dtype=np.uint64)
                with np.errstate(over="ignore"):
                    z = (np.uint64(seed & M64) + np.uint64((step * 0x632BE59BD9B4E019) & M64)
                         + np.uint64(((self._drop_layer(tw, l) + 1) * 0x9E3779B97F4A7C15) & M64)
                         + idx * np.uint64(0xD1B54A32D192ED03))
                    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                    z = z ^ (z >> np.uint64(31))
                u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
                per.append((u >= np.float32(self.dropout)).astype(np.float32).reshape(B, N))
            out.append(per)
        return out

    def _side(self, i):
        if self._sides is None:
            self._sides = [torch.cuda.Stream(device=self.device) for _ in range(3)]
        return self._sides[i]

    def _check_batch(self, bt):
        if bt.B > self.max_batch:
            raise ValueError("batch %d exceeds max_batch %d" % (bt.B, self.max_batch))
        if bt.nnz > self.max_nnz:
            raise ValueError("nnz %d exceeds max_nnz %d" % (bt.nnz, self.max_nnz))

    def _x_ptr(self, tw):
        return tw["act"].data_ptr() + 4 * tw["layout"].seg_start[0]

    def _sparse_forward(self, bt: DeviceBatch, st):
        """Input layer (embedding bags, indicators, numeric columns) into tower 0's x, and the wide logit."""
        plan, spec = self.plan, self.spec
        B, S = bt.B, plan.S
        if spec.has_deep and self._fused_input_layer and self.rec is None and not self._small_on(bt):
            # one launch: embedding bags + wide sum + numeric columns
            tw0 = self.towers[0]
            (dim, sl), = plan.emb_groups.items()
            nd = len(plan.dense_cols)
            call("wd_input_layer_fwd", ptr(self.emb), ptr(self.slots_dev), S, sl[0], len(sl), dim, ptr(bt.ids),
                 ptr(bt.bag_offs), 1 if bt.one_hot else 0, B, self._x_ptr(tw0), tw0["layout"].ld,
                 ptr(bt.dense) if nd else None, bt.dense.stride(0) if nd else 0, ptr(self.dense_cols_dev) if nd else None, nd,
                 ptr(self.wide) if spec.has_wide else None, ptr(self.bias) if spec.has_wide else None,
                 ptr(self.wide_logit) if spec.has_wide else None, st)
            return
        small = self._small_on(bt)
        slots_dev = self.slots_small_dev if small else self.slots_dev
        # row records, ragged bags, ONE embedding group over every column that is not a small table: the gather leaves each bag's
        # wide sum too (the weight sits in the line it fetched), a short launch adds them up per example -- instead of wd_wide_fwd's
        # second pass over the same 1.06 M lines (configs[3]: 30 of the input layer's 120 us)
        groups = self.group_slots_big if small else self.group_slots
        fuse_wide = (self.rec is not None and spec.has_deep and spec.has_wide and len(groups) == 1
                     and next(iter(groups)) == self.emb.shape[1] and os.environ.get("WD_EMBAG_WIDE", "1") != "0"
                     and all(sl.wide for i, sl in enumerate(plan.slots) if i not in (self.small_idx if small else ())))
        if spec.has_deep:
            tw0 = self.towers[0]
            ld = tw0["layout"].ld
            xp = self._x_ptr(tw0)
            for dim, gs in groups.items():
                if fuse_wide:
                    if getattr(self, "_wv_ragged", None) is None:
                        self._wv_ragged = torch.zeros(self.max_batch * max(S, 1), dtype=torch.float32, device=self.device)
                    call("wd_embag_fwd_wide", ptr(self.rec), self.rec_stride, ptr(self.rslots_dev), S, ptr(gs), gs.numel(), dim,
                         ptr(bt.ids), ptr(bt.bag_offs), B, xp, ld, ptr(self._wv_ragged), st)
                else:
                    self.embag_fwd(dim, gs, bt, xp, ld, st, sub=small)
            if self.ind_slots_dev is not None:
                call("wd_indicator_fwd", ptr(self.slots_dev), S, ptr(self.ind_slots_dev), self.ind_slots_dev.numel(),
                     ptr(bt.ids), ptr(bt.bag_offs), B, xp, ld, st)
            if self.dense_cols_dev is not None:
                call("wd_dense_fwd", ptr(bt.dense), bt.dense.stride(0), ptr(self.dense_cols_dev),
                     len(plan.dense_cols), B, xp, ld, st)
        if spec.has_wide and fuse_wide:
            call("wd_wide_sum", ptr(self._wv_ragged), ptr(self.bias), ptr(slots_dev), S, B, ptr(self.wide_logit), st)
        elif spec.has_wide:
            call("wd_wide_fwd", ptr(self.wide), self.rec_stride or 4, ptr(self.bias), ptr(slots_dev), S, ptr(bt.ids),
                 ptr(bt.bag_offs), B, ptr(self.wide_logit), st)
        if small:
            call("wd_small_tables_fwd", ptr(self.emb) if spec.has_deep else None, ptr(self.wide) if spec.has_wide else None,
                 ptr(self.slots_dev), S, ptr(self.small_idx_dev), len(self.small_idx), self.small_rows, self.small_dim,
                 ptr(bt.ids), ptr(bt.bag_offs), B, self._x_ptr(self.towers[0]) if spec.has_deep else None,
                 self.towers[0]["layout"].ld if spec.has_deep else 0, ptr(self.wide_logit) if spec.has_wide else None,
                 self.rec_stride, st)

    def _small_on(self, bt):
        """This batch's small-table columns go through csrc/small_tables.hip: multi-hot batches with the reference's default
        optimizers, on separate tables or on row records (one-id-per-bag batches keep the one-launch bucketing, whose bags cannot
        be long)."""
        return bool(self.small_idx) and not bt.one_hot and self.default_opts

    def embag_fwd(self, dim, gs, bt, xp, ld, st, sub=False):
        """Embedding-bag gather of one dim group into x (the kernel bench.py reports the HBM roofline of).  sub: `gs` is the
        group without its small-table columns."""
        plan = self.plan
        sl = self.group_lists_big[dim] if sub else plan.emb_groups[dim]
        contiguous = sl == list(range(sl[0], sl[0] + len(sl)))
        if self.rec is not None:
            call("wd_embag_fwd_strided", ptr(self.rec), self.rec_stride, ptr(self.rslots_dev), plan.S, ptr(gs), gs.numel(),
                 dim, ptr(bt.ids), ptr(bt.bag_offs), bt.B, xp, ld, st)
        elif contiguous and dim in (4, 8, 16, 32, 64, 128) and len(sl) <= 128:
            call("wd_embag_fwd_range", ptr(self.emb), ptr(self.slots_dev), plan.S, sl[0], len(sl), dim, ptr(bt.ids),
                 None if bt.one_hot else ptr(bt.bag_offs), bt.B, xp, ld, st)
        else:
            call("wd_embag_fwd", ptr(self.emb), ptr(self.slots_dev), plan.S, ptr(gs), gs.numel(), dim, ptr(bt.ids),
                 ptr(bt.bag_offs), bt.B, xp, ld, st)

    def forward(self, bt: DeviceBatch, need_loss=True):
        """Fills self.logit / self.prob (and self.dlogit / self.loss + the logits-layer backward when labels are given)."""
        self._check_batch(bt)
        spec, st = self.spec, _stream()
        B = bt.B
        train = bt.labels is not None and need_loss
        fuse_in = self._chain_input_ok(bt)
        if fuse_in and self.prefetch:
            if not self._prefetched:            # (pipeline.StepGraph gathers a step ahead and patches: nothing to do here)
                # an eager forward (evaluate / predict / an eager step between two chained graph replays) gathers into the
                # activation buffer a primed graph may be counting on: whatever was primed is gone
                self._primed = None
                self._prefetch_input(bt, st, self._apar)
        elif fuse_in:
            self._sparse_exchange(bt, st)       # rows that have to travel first (sharded engine); nothing on one GPU
        else:
            self._sparse_forward(bt, st)
        if spec.has_deep:
            # one launch: fold the BN affines of every layer into its consumer's weights; clear loss (+ G when the
            # per-layer finalize path accumulates into it).  One-launch tower on one GPU: the folded / packed weights are
            # already there -- the previous train step launched this fold behind its dense update (_fold_at_end)
            if not (self._fold_at_end() and self._folded):
                self._fold(train, st)
                self._folded = self._fold_at_end()
            tw0 = self.towers[0]
            nt = len(self.towers)
            for ti, tw in enumerate(self.towers):
                tl = tw["layout"]
                if ti > 0:  # towers share the input layer (AUTO_REUSE, python/lib/dnn.py:83-90)
                    w0 = tl.seg_width[0]
                    tw["act"][:B, tl.seg_start[0]: tl.seg_start[0] + w0].copy_(
                        tw0["act"][:B, tw0["layout"].seg_start[0]: tw0["layout"].seg_start[0] + w0])
                for xs in tl.x_copies:   # first_dense: every second pair of consumers reads its own copy of x
                    w0 = tl.seg_width[0]
                    tw["act"][:B, tl.seg_start[xs]: tl.seg_start[xs] + w0].copy_(
                        tw["act"][:B, tl.seg_start[0]: tl.seg_start[0] + w0])
                if train and tl.mode in ("first_dense", "list"):
                    tw["dact"][:B].zero_()   # windows are accumulated into; the logits window does not cover them all
                if not self.chain:
                    self._tower_hidden_forward(tw, B, st, train)
            if self.chain:
                self._tower_chain(tw0, bt, B, st, train, fuse_in)
            elif nt == 1:
                self._tower_head(tw0, bt, B, st, train, fused=True)
            else:
                # multi-DNN (python/lib/dnn.py:260-274): logits are summed over towers BEFORE the head, so each
                # tower's logits layer runs forward-only here and its backward after the joint head
                for tw in self.towers:
                    self._tower_head(tw, bt, B, st, False, fused=False)
                torch.add(self.towers[0]["logit"], self.towers[1]["logit"], out=self.dnn_logit)
                for tw in self.towers[2:]:
                    self.dnn_logit.add_(tw["logit"])
                self._plain_head(self.dnn_logit, bt, B, st, train)
        else:
            self.loss.zero_()
            self._plain_head(None, bt, B, st, train)
        return self.logit[:B]

    def _plain_head(self, dnn_logit, bt, B, st, train):
        if train:
            call("wd_bce_sum_fwd_bwd", ptr(dnn_logit), ptr(self.wide_logit), ptr(bt.labels), ptr(bt.weights), B,
                 ptr(self.logit), ptr(self.prob), ptr(self.dlogit), None, st)
            self._loss_sum(bt, B, st)
        else:
            # logits / probabilities only: reuse the head kernel with a zero label vector
            zeros = self.dlogit
            zeros[:B].zero_()
            call("wd_bce_sum_fwd_bwd", ptr(dnn_logit), ptr(self.wide_logit), ptr(zeros), None, B, ptr(self.logit),
                 ptr(self.prob), None, None, st)

    def _loss_sum(self, bt, B, st):
        """Batch-SUM loss from the stored logits in a fixed summation order (the head kernels' own sum is a float atomic)."""
        call("wd_bce_loss_sum", ptr(self.logit), ptr(bt.labels), ptr(bt.weights), B, ptr(self.loss), st)

    def _drop_layer(self, tw, l):
        return self.towers.index(tw) * 64 + l          # layer id that enters the keep function

    def _tower_hidden_forward(self, tw, B, st, train=False):
        if self.half:
            return self._tower_hidden_forward_h(tw, B, st)
        tl, metas, L = tw["layout"], tw["metas"], tw["L"]
        act = tw["act"]
        for l in range(L):
            m = metas[l]
            K, N = m["K"], m["N"]
            a_ptr = act.data_ptr() + 4 * tl.in_start[l]
            c_ptr = act.data_ptr() + 4 * tl.seg_start[l + 1]
            call("wd_gemm_nn_bias_act", a_ptr, tl.ld, ptr(tw["Wf"][l]), N, ptr(tw["bf"][l]), capi.WD_FOLD_PARTS,
                 self.act_id, c_ptr, tl.ld, B, N, K, st)
            if train and self.dropout:     # tf.layers.dropout(net, rate, training=True): TRAIN mode only, before BN
                call("wd_dropout_fwd", c_ptr, tl.ld, B, N, self.dropout, ptr(self.drop_seed), self._drop_layer(tw, l), st)
            for cs in tl.copies_of(l + 1):  # connection list: later windows repeat this layer's output (dnn.py:218-221)
                act[:B, tl.seg_start[cs]: tl.seg_start[cs] + N].copy_(act[:B, tl.seg_start[l + 1]: tl.seg_start[l + 1] + N])

    def _tower_hidden_forward_h(self, tw, B, st):
        """fp16-input tower: x (fp32, from the gather) -> half + transposed half; every layer writes both copies."""
        tl, metas, L = tw["layout"], tw["metas"], tw["L"]
        ah, aT, Bp = tw["act_h"], tw["actT_h"], tw["Bp"]
        s0, w0 = tl.seg_start[0], tl.seg_width[0]
        call("wd_cast_transpose_h", tw["act"].data_ptr() + 4 * s0, tl.ld, B, w0, None, 0, 0, ah.data_ptr() + 2 * s0, tl.ld,
             aT.data_ptr() + 2 * s0 * Bp, Bp, st)
        for l in range(L):
            m = metas[l]
            K, N = m["K"], m["N"]
            seg = tl.seg_start[l + 1]
            call("wd_hgemm_nn", ah.data_ptr() + 2 * tl.in_start[l], tl.ld, ptr(tw["WfT_h"][l]), tw["WfT_h"][l].shape[1],
                 ptr(tw["bf"][l]), capi.WD_FOLD_PARTS, self.act_id, ah.data_ptr() + 2 * seg, tl.ld,
                 aT.data_ptr() + 2 * seg * Bp, Bp, B, N, K, st)

    def _head_out(self, tw):
        """Where the logits layer's input gradient goes: simple mode -> dz of the last hidden layer (act' fused),
        otherwise the logits window of dact (plain store; the window covers every segment later accumulated into)."""
        tl, L = tw["layout"], tw["L"]
        if tl.mode == "simple" and L > 0 and not self.dropout:
            return tw["dz"][0].data_ptr(), tw["metas"][L]["K"], self.act_id
        return tw["dact"].data_ptr() + 4 * tl.in_start[L], tl.ld, 0

    def _tower_head(self, tw, bt, B, st, train, fused):
        """Logits layer.  fused=True: + joint head (wide logit, CE loss) + the layer's backward, one launch."""
        tl, metas, L = tw["layout"], tw["metas"], tw["L"]
        m = metas[L]
        a_ptr = tw["act"].data_ptr() + 4 * tl.in_start[L]
        head = "wd_logits_head"
        if self.half:
            a_ptr, head = tw["act_h"].data_ptr() + 2 * tl.in_start[L], "wd_logits_head_h"
        if fused:
            out_ptr, ld_out, act_id = self._head_out(tw) if (train and not self.half) else (None, 0, 0)
            call(head, a_ptr, tl.ld, m["K"], ptr(tw["Wf"][L]), ptr(tw["bf"][L]), capi.WD_FOLD_PARTS,
                 ptr(self.wide_logit), ptr(bt.labels) if train else None, ptr(bt.weights) if train else None, B,
                 ptr(tw["logit"]), ptr(self.logit), ptr(self.prob), ptr(self.dlogit) if train else None,
                 None, out_ptr, ld_out, act_id, ptr(tw["Gpart"][L]) if train else None, st)
            if train:
                self._loss_sum(bt, B, st)
        else:
            call(head, a_ptr, tl.ld, m["K"], ptr(tw["Wf"][L]), ptr(tw["bf"][L]), capi.WD_FOLD_PARTS,
                 None, None, None, B, ptr(tw["logit"]), None, None, None, None, None, 0, 0, None, st)

    # ------------------------------------------------------------------------------------------
    # backward + optimizers
    # ------------------------------------------------------------------------------------------
    def _tower_backward(self, tw, B, st, need_dx, head_done):
        """Hidden layers L-1..0 (and the logits layer when the fused head has not already done it)."""
        if self.half:
            return self._tower_backward_h(tw, B, st, need_dx)
        tl, metas, L = tw["layout"], tw["metas"], tw["L"]
        act, dact = tw["act"], tw["dact"]
        if self.chain:
            act = tw["acts"][self._apar if self.prefetch else 0]
            # forward() already ran the input-gradient chain; what is left are the batch reductions G_l = bn_{l-1}^T dz_l
            # (split-K products) and the sums of the per-tile partials of the bias / BN gradients and of the loss
            # (column-sum jobs), all in grouped launches of at most WD_TN_GROUP_MAX jobs
            nblk = int(call("wd_tower_chain_blocks", B, self.chain_rt))
            spec_jobs, prod_jobs = [], []
            for l in range(L):    # largest product first: its workgroups start while the small ones fill the gaps
                m = metas[l]
                prod_jobs.append(dict(A=act.data_ptr() + 4 * tl.in_start[l], lda=tl.ld, B=tw["dzl"][l].data_ptr(), ldb=m["N"],
                                      C=tw["Gpart"][l].data_ptr(), M=m["K"], N=m["N"], K=B, nsplit=tw["nsplit"][l]))
            # the column-sum jobs go FIRST: a handful of workgroups, each a chain of dependent round trips over the row tiles --
            # at the end of the grid they would start last and finish after the products
            for l in range(L):
                m = metas[l]
                names = ("db", "dg", "dbeta") if "gamma_off" in m else ("db",)
                for nm in names:
                    spec_jobs.append(dict(A=tw[nm + "_part"][l].data_ptr(), lda=m["N"], B=None, C=tw[nm + "_sum"][l].data_ptr(),
                                          N=m["N"], K=nblk))
            nl = (metas[L]["K"] + 1) * metas[L]["N"]
            spec_jobs.append(dict(A=tw["Gpart"][L].data_ptr(), lda=nl, B=None, C=tw["Glog_sum"].data_ptr(), N=nl, K=nblk))
            spec_jobs.append(dict(A=self.loss_part.data_ptr(), lda=1, B=None, C=self.loss.data_ptr(), N=1, K=nblk))
            spec_jobs += prod_jobs
            for i0 in range(0, len(spec_jobs), capi.WD_TN_GROUP_MAX):
                chunk = spec_jobs[i0: i0 + capi.WD_TN_GROUP_MAX]
                jobs = (capi.WdTnJob * len(chunk))()
                for j, d in zip(jobs, chunk):
                    j.A, j.lda, j.B, j.Cpart, j.N, j.K = d["A"], d["lda"], d["B"], d["C"], d["N"], d["K"]
                    if d["B"] is not None:
                        j.ldb, j.M, j.nsplit, j.append_ones = d["ldb"], d["M"], d["nsplit"], 0
                call("wd_gemm_tn_splitk_group", jobs, len(chunk), st)
            return
        simple = tl.mode == "simple" and not self.dropout   # dropout: act' is not fused into the GEMM epilogues
        acc = 0 if tl.mode == "simple" else 1               # simple: every segment has ONE consumer -> plain stores
        if not head_done:
            # multi-tower: dlogit is shared; run this tower's logits-layer backward (dlogit given as `labels`-free input)
            m = metas[L]
            out_ptr, ld_out, act_id = self._head_out(tw)
            a_ptr = act.data_ptr() + 4 * tl.in_start[L]
            call("wd_gemm_tn_splitk", a_ptr, tl.ld, ptr(self.dlogit), 1, ptr(tw["Gpart"][L]), m["K"], 1, B, 1, 1, st)
            if simple and L > 0:
                call("wd_gemm_nt_actbwd", ptr(self.dlogit), 1, ptr(tw["Wf"][L]), 1, out_ptr, ld_out, B, m["K"], 1,
                     a_ptr, tl.ld, act_id, st)
            else:
                call("wd_gemm_nt", ptr(self.dlogit), 1, ptr(tw["Wf"][L]), 1, out_ptr, ld_out, B, m["K"], 1, 0, st)
        # weight-gradient GEMMs (and the per-layer finalize of non-simple towers) go to side stream 1 and run beside the
        # input-gradient GEMM of the same layer; `tn_done` keeps a dz buffer from being rewritten while a TN still reads it
        ov = self.overlap_tn
        main = torch.cuda.current_stream()
        s2 = self._side(1) if ov else None
        st_w = s2.cuda_stream if ov else st
        tn_done = None
        if not self.all_simple:
            if ov:
                s2.wait_stream(main)
            self._finalize_layer(tw, L, st_w, 1 if not head_done else tw["nsplit"][L])
        cur = 0
        for l in range(L - 1, -1, -1):
            m = metas[l]
            K, N = m["K"], m["N"]
            if simple:
                dz_ptr, lddz = tw["dz"][cur].data_ptr(), N
            else:
                seg = tl.seg_start[l + 1]
                dz_ptr, lddz = tw["dz"][0].data_ptr(), N
                for cs in tl.copies_of(l + 1):          # connection list: the gradient of a segment = sum over its copies
                    dact[:B, seg: seg + N].add_(dact[:B, tl.seg_start[cs]: tl.seg_start[cs] + N])
                if tn_done is not None:
                    main.wait_event(tn_done)            # the previous layer's TN still reads dz[0]
                if self.dropout:
                    call("wd_act_bwd_dropout", dact.data_ptr() + 4 * seg, tl.ld, act.data_ptr() + 4 * seg, tl.ld, self.act_id,
                         dz_ptr, lddz, B, N, self.dropout, ptr(self.drop_seed), self._drop_layer(tw, l), st)
                else:
                    call("wd_act_bwd", dact.data_ptr() + 4 * seg, tl.ld, act.data_ptr() + 4 * seg, tl.ld, self.act_id,
                         dz_ptr, lddz, B, N, st)
            a_ptr = act.data_ptr() + 4 * tl.in_start[l]
            ns = tw["nsplit"][l]
            if ov:
                s2.wait_stream(main)                    # dz of this layer is complete on the main stream
            call("wd_gemm_tn_splitk", a_ptr, tl.ld, dz_ptr, lddz, ptr(tw["Gpart"][l]), K, N, B, ns, 1, st_w)
            if not self.all_simple:
                self._finalize_layer(tw, l, st_w, ns)
            prev_done = tn_done
            if ov:
                tn_done = torch.cuda.Event()
                tn_done.record(s2)
            if simple:
                if l > 0:     # dz of layer l-1 = (dz_l Wf_l^T) * act'(a_l-1's output), written to the other buffer
                    if prev_done is not None:
                        main.wait_event(prev_done)      # ... which the TN of layer l+1 was reading
                    call("wd_gemm_nt_actbwd", dz_ptr, lddz, ptr(tw["Wf"][l]), N, tw["dz"][cur ^ 1].data_ptr(), K, B, K, N,
                         a_ptr, tl.ld, self.act_id, st)
                    cur ^= 1
                elif need_dx:
                    call("wd_gemm_nt", dz_ptr, lddz, ptr(tw["Wf"][l]), N, dact.data_ptr() + 4 * tl.in_start[l], tl.ld,
                         B, K, N, 0, st)
            elif l > 0 or need_dx:
                call("wd_gemm_nt", dz_ptr, lddz, ptr(tw["Wf"][l]), N, dact.data_ptr() + 4 * tl.in_start[l], tl.ld, B,
                     K, N, acc, st)
        if ov:
            main.wait_stream(s2)

    def _tower_backward_h(self, tw, B, st, need_dx):
        """Backward of the fp16-input tower (single tower; the fused head already produced dlogit and the logits
        layer's kernel gradient).  Per hidden layer, two launches: the gradient of its OUTPUT segment pulled from all
        consumer layers with the activation derivative fused (-> dz, half + transposed half), then the split-K
        weight gradient."""
        tl, metas, L = tw["layout"], tw["metas"], tw["L"]
        ah, aT, Bp, pZ, zoff = tw["act_h"], tw["actT_h"], tw["Bp"], tw["pZ"], tw["zoff"]
        Z, ZT, wcat = tw["Z_h"], tw["ZT_h"], tw["wcat"]
        call("wd_cast_transpose_h", ptr(self.dlogit), 1, B, 1, None, 0, 0, ptr(Z), pZ, ptr(ZT), Bp, st)   # block L
        ov = self.overlap_tn       # weight gradients on side stream 1, beside the next layer's segment-gradient GEMM
        main = torch.cuda.current_stream()
        s2 = self._side(1) if ov else None
        st_w = s2.cuda_stream if ov else st
        if not self.all_simple:
            if ov:
                s2.wait_stream(main)
            self._finalize_layer(tw, L, st_w, tw["nsplit"][L])
        for l in range(L - 1, -1, -1):
            m, pj = metas[l], tw["pulls"][l + 1]
            K, N = m["K"], m["N"]
            seg = tl.seg_start[l + 1]
            call("wd_hgemm_nt", Z.data_ptr() + 2 * pj["zbeg"], pZ, wcat.data_ptr() + 2 * pj["base"], pj["pitch"], B, N,
                 pj["kred"], None, 0, 0, Z.data_ptr() + 2 * zoff[l], pZ, ZT.data_ptr() + 2 * zoff[l] * Bp, Bp,
                 ah.data_ptr() + 2 * seg, tl.ld, self.act_id, st)
            ns = tw["nsplit"][l]
            if ov:
                s2.wait_stream(main)
            call("wd_hgemm_tn_splitk", aT.data_ptr() + 2 * tl.in_start[l] * Bp, Bp, ZT.data_ptr() + 2 * zoff[l] * Bp, Bp,
                 ptr(tw["Gpart"][l]), K, N, B, ns, st_w)
            if not self.all_simple:
                self._finalize_layer(tw, l, st_w, ns)
        if need_dx:     # gradient of the pooled input x (fp32: the sparse optimizers consume it)
            pj = tw["pulls"][0]
            call("wd_hgemm_nt", Z.data_ptr() + 2 * pj["zbeg"], pZ, wcat.data_ptr() + 2 * pj["base"], pj["pitch"], B,
                 pj["width"], pj["kred"], tw["dact"].data_ptr() + 4 * tl.seg_start[0], tl.ld, 0, None, 0, None, 0, None,
                 0, 0, st)
        if ov:
            main.wait_stream(s2)

    def _finalize_layer(self, tw, l, st, ns):
        m = tw["metas"][l]
        call("wd_mlp_finalize", ptr(tw["Gpart"][l]), ns, ptr(self.P), m["w_off"], m["b_off"], ptr(tw["s"][l]),
             ptr(tw["t"][l]), ptr(tw["gidx"][l]), ptr(tw["bidx"][l]), self.inv, ptr(self.G), m["K"], m["N"], st)

    def sort_occurrences(self, bt, st):
        plan = self.plan
        self._check_batch(bt)
        call("wd_build_sort_keys", ptr(self.slots_dev), plan.S, ptr(bt.ids), ptr(bt.bag_offs), bt.B * plan.S, bt.nnz,
             ptr(self.keys), ptr(self.vals), st)
        call("wd_sort_pairs", ptr(self.keys), ptr(self.vals), ptr(self.keys_sorted), ptr(self.vals_sorted), bt.nnz,
             plan.key_bits, ptr(self.sort_ws), self.sort_ws_bytes, st)

    def _reduce_dense_grads(self):
        """Hook for data-parallel ranks (dist.py: all_reduce(SUM) of the flat gradient buffer)."""

    def _sparse_bucketize(self, bt: DeviceBatch, st, pset=0, prev=None, between=None):
        """Phase 1 of the sparse backward (ids only): occurrences -> row-range buckets (scratch set `pset`).  Flat row update:
        + the sort of every bucket; prev = scratch set of the batch stepped BEFORE this one (pipelined graph): its patch list
        (rows this batch reads too) is filled as well."""
        self._primed = None      # the scratch set is rewritten: a primed graph's token is void (StepGraph.prime() sets it again behind this call)
        plan = self.plan
        self._check_batch(bt)
        bs = self._bucket_sets[pset]
        if self._bucket_onehot_ok(bt):
            # one id per bag: every slot owns B occurrences -> one launch, no count matrix (csrc/onehot_path.hip)
            cols = bt.ids_cols is not None and bt.ids_cols_valid
            flat = self.flat_update and self.prefetch
            call("wd_bucket_onehot", ptr(self.slots_dev), plan.S, ptr(bt.ids_cols if cols else bt.ids), 1 if cols else 0, bt.B,
                 ptr(bs["start"]), ptr(bs["pairs"]), self.n_buckets, self.max_slot_buckets,
                 None if flat else ptr(bs["ticket"]), ptr(bs["long_list"]) if flat else None, st)
            bs["unsorted"], bs["sorted"], bs["ragged"] = True, flat, False
            if flat:
                bp = self._bucket_sets[prev] if prev is not None else None
                call("wd_bucket_sort", ptr(bs["start"]), ptr(bs["pairs"]), self.n_buckets, ptr(bs["long_list"]),
                     (bs["long_list"].numel() - 2) // 2, ptr(bs["big_list"]), bt.B, plan.S, ptr(bp["start"]) if bp else None, ptr(bp["pairs"]) if bp else None, ptr(bp["patch"]) if bp else None, st)
            return
        bs["unsorted"] = bs["sorted"] = bs["ragged"] = False
        call("wd_sparse_bucketize", ptr(self.slots_small_dev if self._small_on(bt) else self.slots_dev), plan.S, ptr(bt.ids), ptr(bt.bag_offs), bt.B, bt.nnz,
             ptr(bs["cnt"]), ptr(bs["start"]), ptr(bs["rank"]), ptr(bs["pairs"]), self.n_buckets, st)
        if between is not None:
            # (train_step(beside_tower=): the next batch's featurizer goes BETWEEN the bucketing and the sort -- the sort's second
            # launch, for buckets of more than 256 pairs, carries 16 KB of LDS and does not start before the window tower has left
            # its CUs; behind it the featurizer would run after the tower and delay the row update by its own 60-100 us)
            between()
            between = None
        if self._flat_ragged_ok(bt):
            # ragged bags on row records: every bucket sorted here, with the bucketing (beside the input layer / the tower), and the
            # update itself a flat launch over the sorted pairs (wd_row_update_ragged) -- the C2 step's split of the work
            call("wd_bucket_sort_ragged", ptr(bs["start"]), ptr(bs["pairs"]), self.n_buckets, ptr(bs["long_list"]),
                 (bs["long_list"].numel() - 2) // 2, ptr(bs["big_list"]), bt.B, plan.S, self.max_nnz, st)
            bs["sorted"] = bs["ragged"] = True

    def _flat_ragged_ok(self, bt):
        """Multi-hot batches on row records, every column that is not a small table embedded with the record's width: bucketing +
        sort ahead (the sort one wavefront per bucket in registers: no LDS, it runs beside the window tower), flat update
        (WD_FLAT_RAGGED=0: the sort stays inside wd_sparse_apply_rec's update workgroups)."""
        skip = self.small_idx if self._small_on(bt) else ()
        return (self.rec is not None and type(self) is WideDeepEngine and self.default_opts and self.plan.S <= 128
                and self.emb.shape[1] in (4, 8, 16) and self.flat_ragged
                and all(sl.deep == "embedding" and int(sl.dim) == self.emb.shape[1]
                        for i, sl in enumerate(self.plan.slots) if i not in skip))

    def _bucket_onehot_ok(self, bt):
        return (bt.one_hot and self.rec is not None and type(self) is WideDeepEngine and bt.nnz == bt.B * self.plan.S
                and os.environ.get("WD_BUCKET_ONEHOT", "1") != "0")

    def _has_sparse_update(self):
        return (bool(self.group_slots) if self.spec.has_deep else False) or self.spec.has_wide

    def _small_backward(self, bt: DeviceBatch, st):
        """The crossed columns' tables (csrc/small_tables.hip): counted in LDS, no sort -- the bucketing skipped them, the big
        tables' update skips them; needs dx / dlogit only."""
        if not self._small_on(bt):
            return
        plan, spec = self.plan, self.spec
        has_emb = bool(self.group_slots) if spec.has_deep else False
        dx_ptr, ld = None, 0
        if has_emb:
            tw0 = self.towers[0]
            dx_ptr, ld = tw0["dact"].data_ptr() + 4 * tw0["layout"].seg_start[0], tw0["layout"].ld
        lr, l1, l2 = (spec.lin_opt[1], spec.lin_opt[2], spec.lin_opt[3]) if spec.has_wide else (0.0, 0.0, 0.0)
        if self.small_ws is None:
            n = int(call("wd_small_tables_ws_floats", len(self.small_idx), self.small_rows, self.small_dim, self.max_batch))
            self.small_ws = torch.zeros(max(n, 1), dtype=torch.float32, device=self.device)
        call("wd_small_tables_bwd", ptr(self.emb) if has_emb else None, ptr(self.emb_acc) if has_emb else None,
             ptr(self.wide) if spec.has_wide else None, ptr(self.slots_dev), plan.S, ptr(self.small_idx_dev),
             len(self.small_idx), self.small_rows, self.small_dim, ptr(bt.ids), ptr(bt.bag_offs), bt.B, dx_ptr, ld,
             ptr(self.dlogit), float(spec.dnn_opt[1]) if spec.has_deep else 0.0, float(lr), float(l1), float(l2),
             ptr(self.small_ws), self.small_ws.numel(), self.rec_stride, st)

    def _sparse_backward(self, bt: DeviceBatch, st, bucketized=False, pset=0, patch=None, small=True):
        """Scatter-add of the row gradients + fused Adagrad (embedding rows) / FTRL (wide rows, bias).  small=False: the caller
        launches the small tables' update itself (_small_backward, on another stream)."""
        plan, spec = self.plan, self.spec
        has_emb = bool(self.group_slots) if spec.has_deep else False
        if not (has_emb or spec.has_wide):
            return
        if not bucketized:
            self._sparse_bucketize(bt, st, pset)
        if small:
            self._small_backward(bt, st)      # (nothing unless this batch's small tables go through csrc/small_tables.hip)
        bsx = self._bucket_sets[pset]
        dx_ptr, ld = None, 0
        if has_emb:
            tw0 = self.towers[0]
            tl0 = tw0["layout"]
            dx_ptr, ld = tw0["dact"].data_ptr() + 4 * tl0.seg_start[0], tl0.ld
        if self.rec is not None and bsx["sorted"] and bsx.get("ragged"):
            call("wd_row_update_ragged", ptr(self.rec), self.rec_stride, self.emb.shape[1], ptr(self.emb_acc), ptr(self.bias),
                 ptr(self.slots_dev), plan.S, bt.B, ptr(bt.bag_offs), dx_ptr, ld, ptr(self.dlogit), 1, float(spec.dnn_opt[1]),
                 float(spec.lin_opt[1]), float(spec.lin_opt[2]), float(spec.lin_opt[3]), ptr(bsx["pairs"]), self.max_nnz,
                 bsx["start"].data_ptr() + 4 * self.n_buckets, ptr(bsx["long_list"]), (bsx["long_list"].numel() - 2) // 2, st)
            return
        if self.rec is not None and bsx["sorted"]:
            nx, pa = None, None
            if patch is not None:
                bn, pn = self._bucket_sets[patch[0]], patch[1]
                tw0 = self.towers[0]
                nx = capi.WdApplyNext()
                nx.pairs = ptr(bn["pairs"])
                nx.x, nx.ldx = tw0["acts"][pn].data_ptr() + 4 * tw0["layout"].seg_start[0], tw0["layout"].ld
                nx.wide_vals = ptr(self.wv[pn])
                pa = ptr(bsx["patch"])
            call("wd_row_update", ptr(self.rec), self.rec_stride, self.emb.shape[1], ptr(self.emb_acc), ptr(self.bias),
                 ptr(self.slots_dev), plan.S, bt.B, dx_ptr, ld, ptr(self.dlogit), float(spec.dnn_opt[1]),
                 float(spec.lin_opt[1]), float(spec.lin_opt[2]), float(spec.lin_opt[3]), ptr(bsx["pairs"]),
                 ptr(bsx["long_list"]), (bsx["long_list"].numel() - 2) // 2, pa, ctypes.byref(nx) if nx is not None else None, st)
            return
        if self.rec is not None:
            nx = capi.WdApplyNext()
            nx.unsorted_buckets = 1 if bsx["unsorted"] else 0
            if patch is not None:
                # (pset_next, p_next): the next batch's buckets and the activation buffer / weight list its input layer was
                # prefetched into BEFORE this update -- rows rewritten here are stored there again
                bn, pn = self._bucket_sets[patch[0]], patch[1]
                tw0 = self.towers[0]
                nx.bucket_start, nx.pairs = ptr(bn["start"]), ptr(bn["pairs"])
                nx.x, nx.ldx = tw0["acts"][pn].data_ptr() + 4 * tw0["layout"].seg_start[0], tw0["layout"].ld
                nx.wide_vals = ptr(self.wv[pn])
            call("wd_sparse_apply_rec", ptr(self.rec), self.rec_stride, self.emb.shape[1], ptr(self.emb_acc), ptr(self.bias),
                 ptr(self.slots_dev), plan.S, ptr(bt.bag_offs), bt.B, dx_ptr, ld, ptr(self.dlogit), 1, float(spec.dnn_opt[1]),
                 float(spec.lin_opt[1]), float(spec.lin_opt[2]), float(spec.lin_opt[3]), ptr(bsx["start"]), ptr(bsx["pairs"]),
                 self.n_buckets, ctypes.byref(nx), st)
            return
        if self.default_opts:
            lr, l1, l2 = (spec.lin_opt[1], spec.lin_opt[2], spec.lin_opt[3]) if spec.has_wide else (0.0, 0.0, 0.0)
            call("wd_sparse_apply", ptr(self.emb) if has_emb else None, ptr(self.emb_acc) if has_emb else None,
                 ptr(self.wide), ptr(self.bias), ptr(self.slots_dev), plan.S, ptr(bt.bag_offs), bt.B, dx_ptr, ld,
                 ptr(self.dlogit), 1, float(spec.dnn_opt[1]) if spec.has_deep else 0.0, float(lr), float(l1), float(l2),
                 ptr(bsx["start"]), ptr(bsx["pairs"]), self.n_buckets, st)
            return
        od = ctypes.byref(self.opt_c["dnn"]) if has_emb else None
        ol = ctypes.byref(self.opt_c["linear"]) if spec.has_wide else None
        call("wd_sparse_apply_opt", ptr(self.emb) if has_emb else None, ptr(self.emb_a) if has_emb else None,
             ptr(self.emb_acc) if has_emb else None, ptr(self.wide), ptr(self.bias), ptr(self.slots_dev), plan.S,
             ptr(bt.bag_offs), bt.B, dx_ptr, ld, ptr(self.dlogit), 1, od, ol, ptr(bsx["start"]), ptr(bsx["pairs"]),
             self.n_buckets, ptr(self.touched), st)
        if self.pow:
            # Adam moves every row of a sparsely updated variable: the rows without a gradient this step
            call("wd_adam_untouched", ptr(self.emb) if (has_emb and "dnn" in self.pow) else None, ptr(self.emb_a),
                 ptr(self.emb_acc), ptr(self.wide) if "linear" in self.pow else None, ptr(self.slots_dev), plan.S,
                 max(int(s.num_buckets) for s in plan.slots), max(plan.total_rows, 1), ptr(self.touched), od, ol, st)

    def _sparse_backward_unfused(self, bt: DeviceBatch, st):
        """Reference path through the separate ABI entry points (device radix sort + one kernel per update);
        kept for cross-checking the fused kernel in tests."""
        plan, spec = self.plan, self.spec
        if not self.default_opts:
            raise NotImplementedError("the sort-based reference path implements Adagrad (dnn) + Ftrl (linear) only")
        if self.rec is not None:
            raise NotImplementedError("the sort-based reference path works on separate tables (row_records=False)")
        B, S = bt.B, plan.S
        has_emb = bool(self.group_slots) if spec.has_deep else False
        if bt.nnz > 0 and (has_emb or spec.has_wide):
            self.sort_occurrences(bt, st)
            if has_emb:
                tw0 = self.towers[0]
                tl0 = tw0["layout"]
                dx_ptr = tw0["dact"].data_ptr() + 4 * tl0.seg_start[0]
                for dim in self.group_slots:
                    call("wd_embag_bwd_adagrad", ptr(self.emb), ptr(self.emb_acc), ptr(self.slots_dev), S, dim,
                         ptr(self.keys_sorted), ptr(self.vals_sorted), bt.nnz, ptr(bt.bag_offs), dx_ptr, tl0.ld,
                         float(spec.dnn_opt[1]), st)
            if spec.has_wide:
                _, lr, l1, l2, _ = spec.lin_opt
                call("wd_wide_bwd_ftrl", ptr(self.wide), ptr(self.slots_dev), S, ptr(self.keys_sorted),
                     ptr(self.vals_sorted), bt.nnz, ptr(self.dlogit), float(lr), float(l1), float(l2), st)
        if spec.has_wide:
            _, lr, l1, l2, _ = spec.lin_opt
            call("wd_bias_ftrl", ptr(self.bias), ptr(self.dlogit), B, float(lr), float(l1), float(l2), st)

    def _overlap_ok(self):
        return type(self)._sparse_backward is WideDeepEngine._sparse_backward   # subclasses with their own exchange opt out

    def _dense_backward(self, bt: DeviceBatch, st, after_products=None):
        """Weight-gradient products, finalize + dense optimizer (and, one-launch tower on one GPU, the fold for the next step).
        after_products: called between the two (pipeline.StepGraph joins the row update there)."""
        spec, B = self.spec, bt.B
        has_emb = bool(self.group_slots) if spec.has_deep else False
        if True:
            head_done = len(self.towers) == 1
            for tw in self.towers:
                self._tower_backward(tw, B, st, need_dx=has_emb, head_done=head_done)
            if after_products is not None:
                after_products()
            fused_opt = False
            if self.all_simple or self.chain:
                if not head_done:
                    raise NotImplementedError("multi-tower + all-layer finalize")  # guarded in __init__
                # single GPU + Adagrad: the dense update rides in the finalize launch (nothing reduces G in between)
                fused_opt = (self.default_opts and type(self)._reduce_dense_grads is WideDeepEngine._reduce_dense_grads
                             and not self.crelu and os.environ.get("WD_FUSE_ADAGRAD", "1") != "0")
                if self.chain and fused_opt:
                    # gradients from the partials + Adagrad + the packed kernels of the next step, one launch
                    self._chain_tail(capi.WD_TAIL_GRAD | capi.WD_TAIL_UPDATE | capi.WD_TAIL_PACK, st)
                    self._folded = True
                elif self.chain:
                    self._chain_tail(capi.WD_TAIL_GRAD, st)
                elif fused_opt:
                    call("wd_mlp_finalize_adagrad_all", ptr(self.layers_dev), self.n_layers, self.max_layer_k, ptr(self.P),
                         ptr(self.Pacc), self.inv, ptr(self.G), float(spec.dnn_opt[1]), st)
                else:
                    call("wd_mlp_finalize_all", ptr(self.layers_dev), self.n_layers, self.max_layer_k, ptr(self.P),
                         self.inv, ptr(self.G), st)
            for tw in self.towers:        # first_dense: gradient of x = sum over its copies
                tl = tw["layout"]
                if has_emb and tl.x_copies:
                    w0 = tl.seg_width[0]
                    dx = tw["dact"][:B, tl.seg_start[0]: tl.seg_start[0] + w0]
                    for xs in tl.x_copies:
                        dx.add_(tw["dact"][:B, tl.seg_start[xs]: tl.seg_start[xs] + w0])
            tw0 = self.towers[0]
            tl0 = tw0["layout"]
            if has_emb and len(self.towers) > 1:
                w0 = tl0.seg_width[0]
                dx0 = tw0["dact"][:B, tl0.seg_start[0]: tl0.seg_start[0] + w0]
                for tw in self.towers[1:]:
                    tl = tw["layout"]
                    dx0.add_(tw["dact"][:B, tl.seg_start[0]: tl.seg_start[0] + w0])
            self._reduce_dense_grads()
            if self.crelu:
                # tied halves: dL/dW = G'[:, :N] - G'[:, N:]; both halves then take +g / -g through the (sign-symmetric)
                # optimizer and stay exact mirrors of each other, slots included
                for tw in self.towers:
                    for l in range(tw["L"]):
                        m = tw["metas"][l]
                        call("wd_crelu_tie", ptr(self.G), m["w_off"], m["b_off"], m["K"], m["N_tf"], st)
            if fused_opt:
                pass
            elif self.default_opts:
                call("wd_adagrad_dense", ptr(self.P), ptr(self.Pacc), ptr(self.G), self.P.numel(), float(spec.dnn_opt[1]),
                     st)
            else:
                call("wd_opt_dense", ptr(self.P), ptr(self.Pa), ptr(self.Pacc), ptr(self.G), self.P.numel(),
                     ctypes.byref(self.opt_c["dnn_dense"]), st)
            if self._fold_at_end() and not (self.chain and fused_opt):   # the NEXT step's packed kernels
                self._fold(False, st)
                self._folded = True

    def backward_and_update(self, bt: DeviceBatch, bucketized=False, pset=0, lookahead=None, before_join=None):
        spec, st = self.spec, _stream()
        B = bt.B
        has_emb = bool(self.group_slots) if spec.has_deep else False
        # One-launch tower: dx / dlogit exist when forward() returns, so the sparse update depends on nothing of the dense
        # branch.  It goes to the side stream (graph branch) but is ENQUEUED AFTER the dense branch: started first its
        # ~3200 small workgroups fill every CU's LDS and the weight-gradient GEMM (critical path) waits -- 0.239 ms --
        # started behind it the two overlap: 0.2117 -> 0.2028 ms.  WD_SPARSE_SIDE=0: strictly after the dense branch.
        sparse_side = bucketized and getattr(self, "chain", False) and os.environ.get("WD_SPARSE_SIDE", "1") == "1"
        if sparse_side:
            ev_fwd = torch.cuda.Event()
            ev_fwd.record(torch.cuda.current_stream())
        if spec.has_deep:
            self._dense_backward(bt, st)
        if sparse_side:
            side = self._side(0)
            side.wait_event(ev_fwd)
            self._sparse_backward(bt, side.cuda_stream, bucketized=True, pset=pset, small=False)
            # the small tables' update behind the dense tail on this stream, which otherwise idles until the big tables' update
            # (configs[3]: 200 us) is done: the two touch different rows
            self._small_backward(bt, st)
            if lookahead is not None:
                # ... and the bucketing of the next batch (ids only; its scratch set was last read by the update before this
                # one).  Beside the next tower instead -- where plain stream order puts it -- its histogram / scatter workgroups
                # (32 KB of LDS) find no CU the tower's workgroup has left room on, and the row update waits for them:
                # profiles/r5_c4_nocross_step_timeline_before.txt, scatter 8 -> 212 us, update from 218
                self._sparse_bucketize(lookahead[0], st, lookahead[1])
            if before_join is not None:
                # what the caller wants launched on this stream while the row update still runs on the side stream -- the
                # featurizer of the NEXT batch (pipeline.StepGraph: configs[3]'s update is 270 us, this stream idles for the last
                # 100 of them, and a featurizer launched behind the join sat on the critical path: profiles/r6_c4_tokens_timeline_before.txt)
                before_join()
                before_join = None
            torch.cuda.current_stream().wait_stream(side)
        elif lookahead is not None:
            raise ValueError("train_step(lookahead=): needs the update on the side stream (lookahead_ok)")
        elif bucketized:
            torch.cuda.current_stream().wait_stream(self._side(0))
            self._sparse_backward(bt, st, bucketized=True, pset=pset)
        else:
            self._sparse_backward(bt, st)
        if before_join is not None:
            before_join()
        if self.dropout:
            call("wd_counter_tick", ptr(self.drop_seed), st)   # next step draws a new mask
        for scope, pw in self.pow.items():     # Adam: beta1^t, beta2^t -> t + 1 (AdamOptimizer._finish)
            o = spec.dnn_opt if scope == "dnn" else spec.lin_opt
            call("wd_adam_tick", ptr(pw), float(o[2]), float(o[3]), st)

    def lookahead_ok(self, bt):
        """train_step(pset=, lookahead=): the ragged bucketing of the NEXT batch inside this step (one-launch tower, the update on
        the side stream)."""
        # (not with small tables: their update already runs there, and the bucketing on top of it costs the row update more than
        # it saves -- configs[3] with its crosses 0.524 -> 0.541 ms/step, without them 0.505 -> 0.495)
        # (round 6: not on row records with the flat ragged update either -- the bucketing + sort of a batch then run on the side
        # stream behind the previous update, beside this batch's input layer and tower: configs[3] without its crosses 0.4954 ->
        # 0.4585 ms/step, profiles/r6_c4_flat_ragged_ab.txt)
        return bool(type(self) is WideDeepEngine and getattr(self, "chain", False) and self.overlap_bucket and self._has_sparse_update()
                    and not self._bucket_onehot_ok(bt) and not self._small_on(bt) and not self._flat_ragged_ok(bt)
                    and os.environ.get("WD_SPARSE_SIDE", "1") == "1")

    def train_step(self, bt: DeviceBatch, pset=None, lookahead=None, before_join=None, beside_tower=None):
        """One step of python/lib/joint.py:224-262: forward, batch-SUM loss, both optimizers.
        pset: this batch's occurrences are already bucketed, in scratch set `pset` (by the previous step's `lookahead`);
        lookahead = (next batch, scratch set): bucket that batch's occurrences behind this step's dense tail, while the row update
        runs on the side stream -- pipeline.StepGraph, for batches `lookahead_ok`.  before_join / beside_tower: launches that do not
        depend on the step (the next batch's featurizer) -- issued on this stream before the step joins its row update, or on the side
        stream behind this batch's bucketing (where there is none: before the join)."""
        if bt.labels is None:
            raise ValueError("train_step needs labels")
        bucketized = False
        if pset is not None:
            bucketized = True
        elif self.overlap_bucket and self._overlap_ok() and self._has_sparse_update():
            main, side = torch.cuda.current_stream(), self._side(0)
            side.wait_stream(main)                       # the ids were produced on the main stream
            if beside_tower is not None and not self._bucket_onehot_ok(bt):
                # launches that depend on nothing of this step, for the side stream behind the bucketing: they meet the input layer
                # and the tower (MFMA-bound, the memory system idle) instead of the row update -- the next batch's featurizer
                with torch.cuda.stream(side):
                    self._sparse_bucketize(bt, side.cuda_stream, between=beside_tower)
                beside_tower = None
            else:
                self._sparse_bucketize(bt, side.cuda_stream)
            bucketized = True
        if beside_tower is not None:
            before_join = beside_tower if before_join is None else (lambda a=before_join, b=beside_tower: (b(), a()))
        self.forward(bt, need_loss=True)
        self.backward_and_update(bt, bucketized=bucketized, pset=pset or 0, lookahead=lookahead, before_join=before_join)
        # the reference bumps global_step once per minimize() plus the explicit assign_add (quirk C.4)
        self.global_step += 3 if self.spec.model_type == "wide_deep" else 2
        return self.loss

    # ------------------------------------------------------------------------------------------
    # HIP-graph capture of the whole step (fixed batch geometry)
    # ------------------------------------------------------------------------------------------
    def capture_train_step(self, bt: DeviceBatch, warmup=2):
        """Capture forward+backward+updates on `bt`'s buffers into a hipGraph; returns a replay callable.
        The caller refreshes the contents of bt's tensors in place between replays."""
        if self.spec.lr_decay:
            import warnings
            warnings.warn("capture_train_step: the model decays its learning rates (train.yaml lr_decay); a captured step keeps the "
                          "rates of the capture")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.train_step(bt)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = hipgraph.new_graph()
        with torch.cuda.graph(graph, stream=side):
            self.train_step(bt)
        self._graph = graph

        def replay():
            graph.replay()
            self.global_step += 3 if self.spec.model_type == "wide_deep" else 2
            return self.loss

        return replay

    # ------------------------------------------------------------------------------------------
    # state exchange in the reference's checkpoint naming
    # ------------------------------------------------------------------------------------------
    def _emb_view(self, buf, i):
        """[num_buckets, dim] view of slot i's rows in an embedding-shaped buffer (the table or one of its optimizer slots)."""
        s, off = self.plan.slots[i], self.plan.emb_off[i]
        if self.rec is not None and buf is self.emb:       # row-record layout: the table is a strided view of the records
            r0 = self.plan.row_base[i]
            return self.rec[r0: r0 + s.num_buckets, : s.dim]
        return buf[off: off + s.num_buckets * s.dim].view(s.num_buckets, s.dim)

    def _slot_bufs(self, scope):
        """[(buffer, checkpoint-name suffix)] of a scope's variable + its optimizer slots."""
        opt = self.spec.dnn_opt if scope == "dnn" else self.spec.lin_opt
        return opt_slot_names(opt)

    def state_shapes(self):
        """{TF variable name: shape} of what export_state() returns -- from the plan alone, nothing is copied off the device
        (estimator._check_tf_state: a restore must not stage a second host copy of every table just to learn the names)."""
        plan, spec = self.plan, self.spec
        out = {}
        if spec.has_deep:
            sufs = [""] + [x for x in self._slot_bufs("dnn") if x is not None]
            for i, s in enumerate(plan.slots):
                if plan.emb_off[i] >= 0:
                    for suf in sufs:
                        out["dnn/input_from_feature_columns/input_layer/%s/embedding_weights%s" % (s.deep_name, suf)] = (
                            int(s.num_buckets), int(s.dim))
            for ti, tw in enumerate(self.towers):
                for l, m in enumerate(tw["metas"]):
                    scope = "dnn/dnn_%d/" % (ti + 1) + ("hiddenlayer_%d/" % l if l < tw["L"] else "logits/")
                    for suf in sufs:
                        out[scope + "kernel" + suf] = (len(plan.tf_rows_of_layer(ti, l)), int(m["N_tf"]))
                        out[scope + "bias" + suf] = (int(m["N_tf"]),)
                        if "gamma_off" in m:
                            out[scope + "batch_normalization/gamma" + suf] = (int(m["N"]),)
                            out[scope + "batch_normalization/beta" + suf] = (int(m["N"]),)
                    if "gamma_off" in m:
                        out[scope + "batch_normalization/moving_mean"] = (int(m["N"]),)
                        out[scope + "batch_normalization/moving_variance"] = (int(m["N"]),)
        if spec.has_wide:
            sufs = [""] + [x for x in self._slot_bufs("linear") if x is not None]
            for s in plan.slots:
                if s.wide:
                    for suf in sufs:
                        out["linear/linear_model/%s/weights%s" % (s.name, suf)] = (int(s.num_buckets), 1)
            for suf in sufs:
                out["linear/linear_model/bias_weights" + suf] = (1,)
        for scope, names in self.pow_names.items():
            out[names[0]], out[names[1]] = (), ()
        out["global_step"] = ()
        return out

    def export_state(self, tables=True):
        """TF-named variables (+ optimizer slots).  tables=False: dense-tower parameters, bias and counters only (the
        embedding / wide tables of a 100M-row model are tens of GB; tests/helpers.compact_oracle samples their rows)."""
        plan, spec = self.plan, self.spec
        out = {}
        if spec.has_deep:
            sa, sb, sc = self._slot_bufs("dnn")
            for i, s in enumerate(plan.slots):
                if tables and plan.emb_off[i] >= 0:
                    nm = "dnn/input_from_feature_columns/input_layer/%s/embedding_weights" % s.deep_name
                    for buf, suf in ((self.emb, ""), (self.emb_a, sa), (self.emb_acc, sb), (self.emb_c, sc)):
                        if suf is not None:
                            out[nm + suf] = self._emb_view(buf, i).cpu().clone()
            for ti, tw in enumerate(self.towers):
                p = "dnn/dnn_%d/" % (ti + 1)
                for l, m in enumerate(tw["metas"]):
                    K, N, Ntf = m["K"], m["N"], m["N_tf"]
                    scope = p + ("hiddenlayer_%d/" % l if l < tw["L"] else "logits/")
                    rows = torch.from_numpy(plan.tf_rows_of_layer(ti, l)).to(self.device)
                    for buf, suf in ((self.P, ""), (self.Pa, sa), (self.Pacc, sb), (self.Pc, sc)):
                        if suf is None:
                            continue
                        out[scope + "kernel" + suf] = buf[m["w_off"]: m["w_off"] + K * N].view(K, N)[rows][:, :Ntf].cpu().clone()
                        out[scope + "bias" + suf] = buf[m["b_off"]: m["b_off"] + Ntf].cpu().clone()
                        if "gamma_off" in m:
                            out[scope + "batch_normalization/gamma" + suf] = buf[m["gamma_off"]: m["gamma_off"] + N].cpu().clone()
                            out[scope + "batch_normalization/beta" + suf] = buf[m["beta_off"]: m["beta_off"] + N].cpu().clone()
                    if "gamma_off" in m:
                        out[scope + "batch_normalization/moving_mean"] = torch.zeros(N)
                        out[scope + "batch_normalization/moving_variance"] = torch.ones(N)
        if spec.has_wide:
            sa, sb, sc = self._slot_bufs("linear")
            b = self.bias.cpu()
            for i, s in enumerate(plan.slots):
                if tables and s.wide:
                    nm = "linear/linear_model/%s/weights" % s.name
                    r0 = plan.row_base[i]
                    blk = self.wide[r0: r0 + s.num_buckets].cpu()
                    for col, suf in ((0, ""), (1, sa), (2, sb), (3, sc)):
                        if suf is not None:
                            out[nm + suf] = blk[:, col:col + 1].clone()
            for col, suf in ((0, ""), (1, sa), (2, sb), (3, sc)):
                if suf is not None:
                    out["linear/linear_model/bias_weights" + suf] = b[col:col + 1].clone()
        for scope, names in self.pow_names.items():
            pw = self.pow[scope].cpu()
            out[names[0]], out[names[1]] = pw[0].clone(), pw[1].clone()
        out["global_step"] = torch.tensor(self.global_step, dtype=torch.int64)
        return out

    def import_state(self, state):
        plan, spec, dev = self.plan, self.spec, self.device
        self._folded = False
        if spec.has_deep:
            sa, sb, sc = self._slot_bufs("dnn")
            for i, s in enumerate(plan.slots):
                if plan.emb_off[i] >= 0:
                    nm = "dnn/input_from_feature_columns/input_layer/%s/embedding_weights" % s.deep_name
                    for buf, suf in ((self.emb, ""), (self.emb_a, sa), (self.emb_acc, sb), (self.emb_c, sc)):
                        if suf is not None and nm + suf in state:
                            self._emb_view(buf, i).copy_(state[nm + suf].to(dev).reshape(s.num_buckets, s.dim))
            for ti, tw in enumerate(self.towers):
                p = "dnn/dnn_%d/" % (ti + 1)
                for l, m in enumerate(tw["metas"]):
                    K, N, Ntf = m["K"], m["N"], m["N_tf"]
                    scope = p + ("hiddenlayer_%d/" % l if l < tw["L"] else "logits/")
                    rows = torch.from_numpy(plan.tf_rows_of_layer(ti, l)).to(dev)
                    odd_a, odd_b = OPT_SLOT_ODD[spec.dnn_opt[0]]
                    for buf, suf, odd in ((self.P, "", True), (self.Pa, sa, odd_a), (self.Pacc, sb, odd_b), (self.Pc, sc, True)):
                        if suf is None or scope + "kernel" + suf not in state:
                            continue
                        kv, bv = state[scope + "kernel" + suf].to(dev), state[scope + "bias" + suf].to(dev)
                        if Ntf != N:      # crelu: the mirrored half (odd quantities negated, accumulators copied)
                            sg = -1.0 if odd else 1.0
                            kv, bv = torch.cat([kv, sg * kv], dim=1), torch.cat([bv, sg * bv])
                        buf[m["w_off"]: m["w_off"] + K * N].view(K, N)[rows] = kv
                        buf[m["b_off"]: m["b_off"] + N] = bv
                        if "gamma_off" in m:
                            buf[m["gamma_off"]: m["gamma_off"] + N] = state[scope + "batch_normalization/gamma" + suf].to(dev)
                            buf[m["beta_off"]: m["beta_off"] + N] = state[scope + "batch_normalization/beta" + suf].to(dev)
        if spec.has_wide:
            sa, sb, sc = self._slot_bufs("linear")
            for i, s in enumerate(plan.slots):
                if s.wide:
                    nm = "linear/linear_model/%s/weights" % s.name
                    r0 = plan.row_base[i]
                    for col, suf in ((0, ""), (1, sa), (2, sb), (3, sc)):
                        if suf is not None and nm + suf in state:
                            self.wide[r0: r0 + s.num_buckets, col:col + 1] = state[nm + suf].to(dev)
            nm = "linear/linear_model/bias_weights"
            for col, suf in ((0, ""), (1, sa), (2, sb), (3, sc)):
                if suf is not None and nm + suf in state:
                    self.bias[col:col + 1] = state[nm + suf].to(dev)
        for scope, names in self.pow_names.items():
            if names[0] in state:
                self.pow[scope][0] = float(state[names[0]])
                self.pow[scope][1] = float(state[names[1]])
        self._primed = None
        if "global_step" in state:
            self.global_step = int(state["global_step"])
        if spec.has_deep and self.chain:
            # the one-launch tower reads MFMA-packed COPIES of the kernels, and a captured step (pipeline.StepGraph, dist.py)
            # bakes in "the copies are current": rewrite them now instead of leaving it to the next eager forward()
            self._chain_tail(capi.WD_TAIL_PACK, _stream())
            self._folded = self._fold_at_end()
