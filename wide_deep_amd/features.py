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


class _Stage(object):
    """Host arrays of one batch packed into ONE buffer and moved with ONE host-to-device copy (a batch of the repo-default
    conf needs ~230 small arrays: CSR offsets, values and gather indices per slot and per cross key)."""

    def __init__(self):
        self.items, self.size = [], 0

    def add(self, arr, dtype):
        a = np.ascontiguousarray(arr, dtype=dtype)
        off = (self.size + 15) // 16 * 16
        self.items.append((off, a))
        self.size = off + a.nbytes
        return len(self.items) - 1

    def upload(self, dev):
        host = np.zeros(max(self.size, 16), dtype=np.uint8)
        for off, a in self.items:
            host[off: off + a.nbytes] = a.reshape(-1).view(np.uint8)
        dbuf = torch.from_numpy(host).to(dev, non_blocking=True)
        self.dbuf = dbuf
        return [dbuf[off: off + a.nbytes].view(_NP2T[a.dtype]) if a.nbytes else
                torch.zeros(0, dtype=_NP2T[a.dtype], device=dev) for off, a in self.items]


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
        for i, s in enumerate(plan.slots):
            if s.kind == "hash":
                pc = raw.cat[s.feature]
                fo = pc.ex_offs
                lens_bs[:, i] = np.diff(fo)
                emit.append((i, "hash", (pc.base, fo, pc.n)))
            elif s.kind == "vocab":
                pc = raw.cat[s.feature]
                fo = pc.ex_offs
                idx = self._vocab_lookup(i, pc)
                keep = idx >= 0
                ex_of = np.repeat(np.arange(B), np.diff(fo))
                cnt = np.bincount(ex_of[keep], minlength=B).astype(np.int64)
                lens_bs[:, i] = cnt
                emit.append((i, "int", (idx[keep], cnt)))
            elif s.kind == "identity":
                v = raw.ints[s.feature]
                keep = v != -1                                   # -1 is the int ignore_value of the sparse conversion
                vals = np.where((v >= 0) & (v < s.num_buckets), v, 0).astype(np.int32)
                lens_bs[:, i] = keep
                emit.append((i, "int", (vals[keep], keep.astype(np.int64))))
            elif s.kind == "bucket":
                x = _normalize(raw.floats[s.feature], s.normalizer)
                lens_bs[:, i] = 1
                emit.append((i, "int", (_bucketize(x, s.boundaries), np.ones(B, np.int64))))
            elif s.kind == "cross":
                cnt = np.ones(B, dtype=np.int64)
                keys = []
                for k in s.cross_keys:
                    if k.kind == "string":
                        pc = raw.cat[k.feature]
                        fo = pc.ex_offs
                        real = np.diff(fo).astype(np.int64)
                        if self.cross_padding == "tf_dense":
                            lmax = int(real.max()) if B else 0
                            kc = np.full(B, lmax, dtype=np.int64)
                            # gather indices into fp: real tokens then the '' fingerprint up to lmax
                            col = np.arange(lmax)[None, :]
                            gi = np.where(col < real[:, None], tok_base[k.feature] + fo[:-1, None] + col, empty_index)
                            keys.append(("fp", gi.reshape(-1), kc))
                        else:
                            gi = tok_base[k.feature] + np.arange(pc.n)
                            keys.append(("fp", gi, real))
                            kc = real
                    elif k.kind == "identity":
                        v = raw.ints[k.feature]
                        keep = v != -1
                        vals = np.where((v >= 0) & (v < k.num_buckets), v, 0).astype(np.int64)
                        kc = keep.astype(np.int64)
                        keys.append(("int", vals[keep], kc))
                    else:   # bucketized raw value (un-normalised numeric column inside crosses, quirk C.5)
                        vals = _bucketize(raw.floats[k.feature], k.boundaries).astype(np.int64)
                        kc = np.ones(B, dtype=np.int64)
                        keys.append(("int", vals, kc))
                    cnt = cnt * kc
                lens_bs[:, i] = cnt
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
        def csr(cnt):
            o = np.zeros(B + 1, dtype=np.int32)
            np.cumsum(cnt, out=o[1:])
            return o

        stg = _Stage()
        h_bag = stg.add(bag_offs, np.int32)
        todo = []
        for i, kind, payload in emit:
            if kind == "hash":
                base, fo, ntok = payload
                todo.append((i, kind, (base, stg.add(fo, np.int32))))
            elif kind == "int":
                vals, cnt = payload
                todo.append((i, kind, (stg.add(vals if len(vals) else np.zeros(1, np.int32), np.int32), stg.add(csr(cnt), np.int32))))
            else:
                if len(payload) > capi.WD_MAX_CROSS_KEYS:
                    raise ValueError("crossed column `%s` has more than %d keys" % (plan.slots[i].name, capi.WD_MAX_CROSS_KEYS))
                todo.append((i, kind, [(vk, stg.add(v if len(v) else np.zeros(1, np.int64), np.int64), stg.add(csr(cnt), np.int32))
                                       for vk, v, cnt in payload]))
        nd = len(plan.dense_cols)
        h_dense = stg.add(np.stack([raw.floats[d.feature] for d in plan.dense_cols], axis=1), np.float32) if nd else None
        h_lab = stg.add(raw.labels, np.float32) if raw.labels is not None else None
        use_w = raw.weights is not None and self.engine.spec.use_weight_column
        h_wts = stg.add(raw.weights, np.float32) if use_w else None
        dv = stg.upload(self.dev)
        d_bag = dv[h_bag]
        ids = torch.zeros(max(nnz, 1), dtype=torch.int32, device=self.dev)

        # ---- 4. id emission on the device --------------------------------------------------------------------
        keep_alive = [d_bytes, d_offs, fp, stg.dbuf]
        for i, kind, payload in todo:
            s = plan.slots[i]
            if kind == "hash":
                base, h_fo = payload
                call("wd_emit_hash_slot", fp.data_ptr() + 8 * base, ptr(dv[h_fo]), B, s.num_buckets, ptr(d_bag), S, i, ptr(ids), st)
            elif kind == "int":
                h_v, h_fo = payload
                call("wd_emit_int_slot", ptr(dv[h_v]), ptr(dv[h_fo]), B, ptr(d_bag), S, i, ptr(ids), st)
            else:
                ck = capi.WdCrossKeys()
                ck.nkeys = len(payload)
                for k, (vk, h_v, h_fo) in enumerate(payload):
                    d_v = fp[dv[h_v]] if vk == "fp" else dv[h_v]           # "fp": gather of fingerprints incl. '' padding
                    keep_alive.append(d_v)
                    ck.vals[k], ck.offs[k] = d_v.data_ptr(), dv[h_fo].data_ptr()
                call("wd_cross_hash", ctypes.byref(ck), B, s.hash_key, s.num_buckets, ptr(d_bag), S, i, ptr(ids), st)

        dense = dv[h_dense].view(B, nd) if nd else None
        labels = dv[h_lab] if h_lab is not None else None
        weights = dv[h_wts] if h_wts is not None else None
        bt = DeviceBatch(B, ids, d_bag, dense, labels, weights, nnz=nnz, one_hot=bool((lens_bs == 1).all()))
        bt._keep = keep_alive   # the emission kernels are asynchronous
        return bt
