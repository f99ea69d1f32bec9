"""RawBatch (parsed TSV rows) -> DeviceBatch (example-major bag CSR of ids in HBM).

Runs the feature-column transforms the reference declares in `_build_model_columns`
(python/lib/build_estimator.py:83-158) for one batch:

  hash_bucket  tokens -> wd_fingerprint64 (one launch for ALL string features) -> wd_emit_hash_slot (% buckets)
  vocab        token -> index in vocabulary_list, OOV (-1) dropped           -> wd_emit_int_slot
  identity     id if 0 <= id < n else 0 (default_value=0); -1 dropped        -> wd_emit_int_slot
  bucketized   #boundaries <= normalizer(x)   (wide column, quirk C.5)       -> wd_emit_int_slot
  crossed      per key: Fingerprint64 of the raw tokens / int ids of identity & bucketized keys -> wd_cross_hash
  numeric      raw float; the normalizer runs in wd_dense_fwd

`cross_padding`: 'tf_dense' (default) reproduces the reference, where crossed columns see the
padded_batch tensors of their string keys including the '' padding (every example contributes Lmax values per string
key, Lmax = longest list of that feature IN THE BATCH); 'ragged' crosses only real tokens.

The small per-batch index arithmetic (bag lengths, vocabulary lookups, bucketize of the <= 3 continuous columns) is
host-side numpy, like the reference's graph-build-time Python; the hashing and all id emission run on the GPU.
"""
import ctypes

import numpy as np
import torch

from . import capi
from .capi import call, ptr
from .engine import DeviceBatch


def _normalize(x, normalizer):
    if not normalizer:
        return x
    kind, p0, p1 = normalizer
    x = x.astype(np.float32)
    if kind == "min_max":
        return (x - np.float32(p0)) / (np.float32(p1) - np.float32(p0))
    if kind == "standard":
        return (x - np.float32(p0)) / np.float32(p1)
    return np.log(x)


def _bucketize(x, boundaries):
    """tf bucketized_column: number of boundaries <= x (float32 compare)."""
    b = np.asarray(boundaries, dtype=np.float32)
    return np.searchsorted(b, x.astype(np.float32), side="right").astype(np.int32)


_NP2T = {np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.float32): torch.float32,
         np.dtype(np.uint8): torch.uint8}


ONES = object()    # count vector of a single-valued column (every example has exactly one value)


class _Stage(object):
    """Host arrays of one batch packed into ONE buffer and moved with ONE host-to-device copy (a batch of the repo-default
    conf needs ~230 small arrays: CSR offsets, values and gather indices per slot and per cross key)."""

    def __init__(self):
        self.items, self.size, self.seen, self.refs = [], 0, {}, []

    def add(self, arr, dtype):
        key = (id(arr), np.dtype(dtype))
        if key in self.seen:                      # the same host array staged once (callers keep it alive until upload)
            return self.seen[key]
        self.seen[key] = len(self.items)
        self.refs.append(arr)                     # pins id(arr) for the life of the stage
        a = np.ascontiguousarray(arr, dtype=dtype)
        off = (self.size + 15) // 16 * 16
        self.items.append((off, a))
        self.size = off + a.nbytes
        return len(self.items) - 1

    def upload(self, dev):
        host = np.zeros(max(self.size, 16), dtype=np.uint8)
        for off, a in self.items:
            host[off: off + a.nbytes] = a.reshape(-1).view(np.uint8)
        self.dbuf = torch.from_numpy(host).to(dev, non_blocking=True)
        self.base = self.dbuf.data_ptr()

    def ptr(self, h):
        """device address of array h (what the C ABI takes)"""
        return ctypes.c_void_p(self.base + self.items[h][0])

    def tensor(self, h):
        off, a = self.items[h]
        if not a.nbytes:
            return torch.zeros(0, dtype=_NP2T[a.dtype], device=self.dbuf.device)
        return self.dbuf[off: off + a.nbytes].view(_NP2T[a.dtype])


class Featurizer(object):
    def __init__(self, engine, cross_padding="tf_dense"):
        if cross_padding not in ("tf_dense", "ragged"):
            raise ValueError("cross_padding must be 'tf_dense' or 'ragged'")
        # sharded engines hash in the GLOBAL id space (engine.hash_plan); only the engine splits an id into (owner, local row)
        self.engine, self.plan, self.cross_padding = engine, getattr(engine, "hash_plan", engine.plan), cross_padding
        self.dev = engine.device
        slots = self.plan.slots
        # string features whose fingerprints are needed (hash slots + string cross keys)
        feats = []
        for s in slots:
            if s.kind == "hash" and s.feature not in feats:
                feats.append(s.feature)
            if s.kind == "cross":
                for k in s.cross_keys:
                    if k.kind == "string" and k.feature not in feats:
                        feats.append(k.feature)
        self.fp_features = feats
        self.vocab_maps = {i: {v.encode(): j for j, v in enumerate(s.vocab)} for i, s in enumerate(slots) if s.kind == "vocab"}
        # packed vocabularies for the C lookup of csrc/tsv_ingest.c
        self.vocab_packed = {}
        for i, s in enumerate(slots):
            if s.kind == "vocab":
                bs = [v.encode() for v in s.vocab]
                vo = np.zeros(len(bs) + 1, dtype=np.int32)
                np.cumsum([len(b) for b in bs], out=vo[1:])
                self.vocab_packed[i] = (np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy(), vo, len(bs))

    def _vocab_lookup(self, slot, pc):
        """index in vocabulary_list per token of the feature (-1 = out of vocabulary)"""
        from .dataset import ingest_lib
        L = ingest_lib()
        if L is None:
            vm = self.vocab_maps[slot]
            return np.fromiter((vm.get(t, -1) for t in pc.tokens()), dtype=np.int32, count=pc.n)
        vb, vo, nv = self.vocab_packed[slot]
        out = np.zeros(max(pc.n, 1), dtype=np.int32)
        L.wd_vocab_lookup(ctypes.c_void_p(pc.bytes.ctypes.data), ctypes.c_void_p(pc.tok_offs.ctypes.data),
                          ctypes.c_int64(pc.base), ctypes.c_int64(pc.base + pc.n), ctypes.c_void_p(vb.ctypes.data),
                          ctypes.c_void_p(vo.ctypes.data), ctypes.c_int32(nv), ctypes.c_void_p(out.ctypes.data))
        return out[: pc.n]

    def _dev(self, a, dtype):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.dev, dtype=dtype, non_blocking=True)

    def to_device(self, raw):
        plan, eng = self.plan, self.engine
        B, S = raw.B, plan.S
        st = torch.cuda.current_stream().cuda_stream
        # ---- 1. fingerprints of every token of the batch, one launch.  The ingest already packed all string features
        #         into one byte buffer + offsets (feature-major) with one trailing '' token (padding of crossed columns)
        T = len(raw.tok_offs) - 2
        empty_index = T
        tok_base = {f: c.base for f, c in raw.cat.items()}
        d_bytes, d_offs = self._dev(raw.tok_bytes, torch.uint8), self._dev(raw.tok_offs, torch.int32)
        fp = torch.empty(T + 1, dtype=torch.int64, device=self.dev)
        call("wd_fingerprint64", ptr(d_bytes), ptr(d_offs), T + 1, ptr(fp), st)

        # ---- 2. bag lengths per (example, slot) on the host --------------------------------------------------
        lens_bs = np.zeros((B, S), dtype=np.int64)
        emit = []   # (slot, kind, payload)
        lens_cache, str_keys = {}, {}

        def lens_of(feature):
            if feature not in lens_cache:
                lens_cache[feature] = np.diff(raw.cat[feature].ex_offs).astype(np.int64)
            return lens_cache[feature]
        for i, s in enumerate(plan.slots):
            if s.kind == "hash":
                pc = raw.cat[s.feature]
                fo = pc.ex_offs
                lens_bs[:, i] = lens_of(s.feature)
                emit.append((i, "hash", (pc.base, fo, pc.n)))
            elif s.kind == "vocab":
                pc = raw.cat[s.feature]
                fo = pc.ex_offs
                idx = self._vocab_lookup(i, pc)
                keep = idx >= 0
                ex_of = np.repeat(np.arange(B), lens_of(s.feature))
                cnt = np.bincount(ex_of[keep], minlength=B).astype(np.int64)
                lens_bs[:, i] = cnt
                emit.append((i, "int", (idx[keep], cnt)))
            elif s.kind == "identity":
                v = raw.ints[s.feature]
                keep = v != -1                                   # -1 is the int ignore_value of the sparse conversion
                vals = np.where((v >= 0) & (v < s.num_buckets), v, 0).astype(np.int32)
                lens_bs[:, i] = keep
                emit.append((i, "int", (vals[keep], ONES if keep.all() else keep.astype(np.int64))))
            elif s.kind == "bucket":
                x = _normalize(raw.floats[s.feature], s.normalizer)
                lens_bs[:, i] = 1
                emit.append((i, "int", (_bucketize(x, s.boundaries), ONES)))
            elif s.kind == "cross":
                cnt = ONES
                keys = []
                for k in s.cross_keys:
                    if k.kind == "string":
                        if k.feature not in str_keys:            # the same feature is a key of several crossed columns
                            pc = raw.cat[k.feature]
                            fo = pc.ex_offs
                            real = lens_of(k.feature)
                            if self.cross_padding == "tf_dense":
                                lmax = int(real.max()) if B else 0
                                kc = ONES if lmax == 1 else np.full(B, lmax, dtype=np.int64)
                                # gather indices into fp: real tokens then the '' fingerprint up to lmax
                                col = np.arange(lmax)[None, :]
                                gi = np.where(col < real[:, None], tok_base[k.feature] + fo[:-1, None] + col, empty_index)
                                str_keys[k.feature] = (gi.reshape(-1), kc)
                            else:
                                str_keys[k.feature] = (tok_base[k.feature] + np.arange(pc.n), real)
                        gi, kc = str_keys[k.feature]
                        keys.append(("fp", gi, kc))
                    elif k.kind == "identity":
                        v = raw.ints[k.feature]
                        keep = v != -1
                        vals = np.where((v >= 0) & (v < k.num_buckets), v, 0).astype(np.int64)
                        kc = ONES if keep.all() else keep.astype(np.int64)
                        keys.append(("int", vals[keep], kc))
                    else:   # bucketized raw value (un-normalised numeric column inside crosses, quirk C.5)
                        vals = _bucketize(raw.floats[k.feature], k.boundaries).astype(np.int64)
                        kc = ONES
                        keys.append(("int", vals, kc))
                    if kc is not ONES:
                        cnt = kc if cnt is ONES else cnt * kc
                lens_bs[:, i] = 1 if cnt is ONES else cnt
                emit.append((i, "cross", keys))
            else:
                raise ValueError("unknown slot kind %s" % s.kind)
        nnz = int(lens_bs.sum())
        if nnz > eng.max_nnz or B > eng.max_batch:
            raise ValueError("batch (B=%d, nnz=%d) exceeds engine capacity (max_batch=%d, max_nnz=%d)"
                             % (B, nnz, eng.max_batch, eng.max_nnz))
        bag_offs = np.zeros(B * S + 1, dtype=np.int32)
        np.cumsum(lens_bs.reshape(-1), out=bag_offs[1:])

        # ---- 3. everything the emission kernels read: one packed host buffer, one copy ----------------------------
        stg = _Stage()
        h_bag = stg.add(bag_offs, np.int32)
        h_iota = stg.add(np.arange(B + 1, dtype=np.int32), np.int32)     # CSR offsets of every single-valued column

        def csr(cnt):
            if cnt is ONES:
                return h_iota
            o = np.zeros(B + 1, dtype=np.int32)
            np.cumsum(cnt, out=o[1:])
            return stg.add(o, np.int32)
        todo = []
        for i, kind, payload in emit:
            if kind == "hash":
                base, fo, ntok = payload
                todo.append((i, kind, (base, stg.add(fo, np.int32))))
            elif kind == "int":
                vals, cnt = payload
                todo.append((i, kind, (stg.add(vals if len(vals) else np.zeros(1, np.int32), np.int32), csr(cnt))))
            else:
                if len(payload) > capi.WD_MAX_CROSS_KEYS:
                    raise ValueError("crossed column `%s` has more than %d keys" % (plan.slots[i].name, capi.WD_MAX_CROSS_KEYS))
                todo.append((i, kind, [(vk, stg.add(v if len(v) else np.zeros(1, np.int64), np.int64), csr(cnt))
                                       for vk, v, cnt in payload]))
        nd = len(plan.dense_cols)
        h_dense = stg.add(np.stack([raw.floats[d.feature] for d in plan.dense_cols], axis=1), np.float32) if nd else None
        h_lab = stg.add(raw.labels, np.float32) if raw.labels is not None else None
        use_w = raw.weights is not None and self.engine.spec.use_weight_column
        h_wts = stg.add(raw.weights, np.float32) if use_w else None
        stg.upload(self.dev)
        d_bag = stg.tensor(h_bag)
        ids = torch.zeros(max(nnz, 1), dtype=torch.int32, device=self.dev)

        # ---- 4. id emission on the device --------------------------------------------------------------------
        keep_alive = [d_bytes, d_offs, fp, stg.dbuf]
        p_bag, p_ids = ptr(d_bag), ptr(ids)
        gathered = {}
        for i, kind, payload in todo:
            s = plan.slots[i]
            if kind == "hash":
                base, h_fo = payload
                call("wd_emit_hash_slot", fp.data_ptr() + 8 * base, stg.ptr(h_fo), B, s.num_buckets, p_bag, S, i, p_ids, st)
            elif kind == "int":
                h_v, h_fo = payload
                call("wd_emit_int_slot", stg.ptr(h_v), stg.ptr(h_fo), B, p_bag, S, i, p_ids, st)
            else:
                ck = capi.WdCrossKeys()
                ck.nkeys = len(payload)
                for k, (vk, h_v, h_fo) in enumerate(payload):
                    if vk == "fp":                                 # gather of fingerprints incl. '' padding
                        if h_v not in gathered:
                            gathered[h_v] = fp[stg.tensor(h_v)]
                        ck.vals[k] = gathered[h_v].data_ptr()
                    else:
                        ck.vals[k] = stg.ptr(h_v).value
                    ck.offs[k] = stg.ptr(h_fo).value
                call("wd_cross_hash", ctypes.byref(ck), B, s.hash_key, s.num_buckets, p_bag, S, i, p_ids, st)

        dense = stg.tensor(h_dense).view(B, nd) if nd else None
        labels = stg.tensor(h_lab) if h_lab is not None else None
        weights = stg.tensor(h_wts) if h_wts is not None else None
        bt = DeviceBatch(B, ids, d_bag, dense, labels, weights, nnz=nnz, one_hot=bool((lens_bs == 1).all()))
        bt._keep = keep_alive + list(gathered.values())   # the emission kernels are asynchronous
        return bt
