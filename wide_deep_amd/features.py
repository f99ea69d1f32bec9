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

Two implementations of the same transforms:

  mode="device" (default): the host uploads the parsed batch as it is (token bytes + offsets, integer / float features as
      [feature][example] matrices: ONE staged copy) and the whole bag CSR is built on the GPU from a device-side slot table:
      wd_fingerprint64 -> wd_feat_vocab_lookup -> wd_feat_lens -> wd_feat_offsets (scan) -> wd_feat_emit, on the featurizer's
      own stream (it overlaps with the train step of the previous batch; the only host wait is for the id count).
  mode="host": per-slot numpy for the index arithmetic (bag lengths, vocabulary lookups, bucketize) and one emit launch per
      column -- round 1's path, kept as the cross-check of the device path (tests/test_gpu_c1.py) and for the CPU test of
      the host logic (tests/test_featurizer_host_cpu.py).
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


class FixedStage(object):
    """The staged form of a parsed batch with FIXED capacities and therefore fixed device addresses: what a featurizer + train step
    captured into ONE hipGraph (estimator.WideAndDeepClassifier.train, one graph per batch size) reads its batch from.  `fill(raw)`
    packs a RawBatch into a pinned host copy of the layout (rotating: a copy may still be in flight) and issues ONE host-to-device
    copy on the current stream; what varies from batch to batch besides the contents -- the token count -- travels in the buffer
    (`hdr[0]`), the kernels read it on the device (wd_fingerprint64_dyn, wd_feat_batch_t.ntok_dev)."""

    def __init__(self, fz, B, tok_cap, bytes_cap, labels=True, weights=False, n_host=3):
        plan = fz.plan
        F, NI, NF, nd = max(len(fz.str_feats), 1), max(len(fz.int_feats), 1), max(len(fz.float_rows), 1), len(plan.dense_cols)
        self.fz, self.B, self.tok_cap, self.bytes_cap = fz, int(B), int(tok_cap), int(bytes_cap)
        lay = [("bytes", np.uint8, (self.bytes_cap,)), ("toffs", np.int32, (self.tok_cap + 2,)), ("ex", np.int32, (F, B + 1)),
               ("base", np.int32, (F,)), ("tokn", np.int32, (F,)), ("lmax", np.int32, (F,)), ("hdr", np.int32, (4,)),
               ("ints", np.int64, (NI, max(B, 1))), ("floats", np.float32, (NF, max(B, 1)))]
        if nd:
            lay.append(("dense", np.float32, (B, nd)))
        if labels:
            lay.append(("lab", np.float32, (B,)))
        if weights:
            lay.append(("wts", np.float32, (B,)))
        self.items, off = {}, 0
        for name, dt, shape in lay:
            off = (off + 15) // 16 * 16
            nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
            self.items[name] = (off, np.dtype(dt), shape, nbytes)
            off += nbytes
        self.size = (off + 15) // 16 * 16
        self.dbuf = torch.zeros(self.size, dtype=torch.uint8, device=fz.dev)
        self.base = self.dbuf.data_ptr()
        self._host = [torch.zeros(self.size, dtype=torch.uint8).pin_memory() for _ in range(n_host)]
        self._views = [{k: h.numpy()[o: o + nb].view(dt).reshape(shape) for k, (o, dt, shape, nb) in self.items.items()} for h in self._host]
        self._busy = [None] * n_host
        self._k = 0
        self.h = {k: (k if k in self.items else None) for k in ("bytes", "toffs", "ex", "base", "tokn", "lmax", "hdr", "ints", "floats",
                                                                 "dense", "lab", "wts")}

    def fits(self, raw):
        return (raw.B == self.B and len(raw.tok_offs) - 2 <= self.tok_cap and len(raw.tok_bytes) <= self.bytes_cap
                and (raw.labels is not None) == ("lab" in self.items)
                and (raw.weights is not None and self.fz.engine.spec.use_weight_column) == ("wts" in self.items))

    def ptr(self, h):
        return ctypes.c_void_p(self.base + self.items[h][0])

    def tensor(self, h):
        o, dt, shape, nb = self.items[h]
        return self.dbuf[o: o + nb].view(_NP2T[dt])

    def fill(self, raw):
        """pack `raw` (the caller checked `fits`) and copy it to the device on the current stream"""
        fz, plan = self.fz, self.fz.plan
        k = self._k
        self._k = (k + 1) % len(self._host)
        if self._busy[k] is not None:
            self._busy[k].synchronize()              # the copy that last read this pinned buffer
        v = self._views[k]
        nb, T = len(raw.tok_bytes), len(raw.tok_offs) - 2
        v["bytes"][:nb] = raw.tok_bytes
        v["toffs"][: T + 2] = raw.tok_offs
        if getattr(raw, "packed", None) is not None:
            self._fill_packed(raw, v, T)
            self.dbuf.copy_(self._host[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._busy[k] = ev
            return
        for j, f in enumerate(fz.str_feats):
            pc = raw.cat[f]
            v["ex"][j] = pc.ex_offs
            v["base"][j], v["tokn"][j] = pc.base, pc.n
            v["lmax"][j] = int(np.diff(pc.ex_offs).max()) if self.B else 0
        v["hdr"][0] = T
        for j, f in enumerate(fz.int_feats):
            v["ints"][j] = raw.ints[f]
        for j, (f, log) in enumerate(fz.float_rows):
            x = np.asarray(raw.floats[f], dtype=np.float32)
            v["floats"][j] = np.log(x) if log else x
        if "dense" in v:
            for j, d in enumerate(plan.dense_cols):
                v["dense"][:, j] = raw.floats[d.feature]
        if "lab" in v:
            v["lab"][:] = raw.labels
        if "wts" in v:
            v["wts"][:] = raw.weights
        self.dbuf.copy_(self._host[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._busy[k] = ev


    def _fill_packed(self, raw, v, T):
        """`fill` from the C parser's feature-major arrays (dataset.RawBatch.packed): a dozen array copies per batch where the
        per-feature loop makes ~6 small numpy calls per feature -- 0.54 ms of host time per step on the shipped conf (33 string + 6
        numeric features) at ANY batch size, which bound `python train.py`'s loop at batch 64-512 (profiles/r6_c1_train_loop.md)."""
        fz, plan = self.fz, self.fz.plan
        pk = raw.packed
        key = (pk["str"][0], pk["int"][0], pk["flt"][0])
        m = getattr(self, "_pmap", None)
        if m is None or m[0] != key:
            sn, inn, fn = {f: j for j, f in enumerate(key[0])}, {f: j for j, f in enumerate(key[1])}, {f: j for j, f in enumerate(key[2])}
            m = self._pmap = (key,
                              np.asarray([sn[f] for f in fz.str_feats], dtype=np.int64),
                              np.asarray([inn[f] for f in fz.int_feats], dtype=np.int64),
                              np.asarray([fn[f] for f, _ in fz.float_rows], dtype=np.int64),
                              np.asarray([j for j, (_, log) in enumerate(fz.float_rows) if log], dtype=np.int64),
                              np.asarray([fn[d.feature] for d in plan.dense_cols], dtype=np.int64))
        _, si, ii, fi, logrows, di = m
        _, ex, tok_base, ntok = pk["str"]
        F = len(si)
        if F:
            e = v["ex"][:F]
            np.take(ex, si, axis=0, out=e)
            v["base"][:F] = tok_base[si]
            v["tokn"][:F] = ntok[si]
            v["lmax"][:F] = (e[:, 1:] - e[:, :-1]).max(axis=1) if self.B else 0
        v["hdr"][0] = T
        if len(ii):
            v["ints"][: len(ii)] = pk["int"][1][ii]
        if len(fi):
            x = pk["flt"][1][fi]
            if len(logrows):
                x[logrows] = np.log(x[logrows])
            v["floats"][: len(fi)] = x
        if "dense" in v:
            v["dense"][:] = pk["flt"][1][di].T
        if "lab" in v:
            v["lab"][:] = raw.labels
        if "wts" in v:
            v["wts"][:] = raw.weights


class ParsedDeviceBatch(object):
    """A parsed batch resident in HBM (the staged buffer of Featurizer._stage: token bytes + offsets, per-feature example
    ranges, integer / float feature matrices, labels) with the buffers the featurizer's launches fill: fingerprints, vocabulary
    indices, bag lengths, the bag CSR `bag`, the ids `ids` (capacity `cap`), device flags [overflow, some bag is not one id].
    `.batch` is the DeviceBatch over them (Featurizer.resident / run / finalize)."""
    pass


class Featurizer(object):
    def __init__(self, engine, cross_padding="tf_dense", mode=None):
        if cross_padding not in ("tf_dense", "ragged"):
            raise ValueError("cross_padding must be 'tf_dense' or 'ragged'")
        import os
        # (an engine that is not on a GPU only exists as a stub in the CPU tests of the host logic, which replace the entry points)
        mode = mode or os.environ.get("WD_FEATURIZER", "device" if torch.device(engine.device).type == "cuda" else "host")
        if mode not in ("device", "host"):
            raise ValueError("featurizer mode must be 'device' or 'host'")
        self.mode = mode
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
        if mode == "device":
            self._build_device_tables()

    # ---- device mode: slot table, boundary table, vocabularies -- built once ------------------------------------------------
    def _build_device_tables(self):
        slots = self.plan.slots
        self.str_feats, self.int_feats, self.float_rows, bounds = [], [], [], []

        def sidx(f):
            if f not in self.str_feats:
                self.str_feats.append(f)
            return self.str_feats.index(f)

        def iidx(f):
            if f not in self.int_feats:
                self.int_feats.append(f)
            return self.int_feats.index(f)

        def fidx(f, log):
            if (f, log) not in self.float_rows:
                self.float_rows.append((f, log))       # log: the host applies np.log (bit-exact with the oracle's numpy)
            return self.float_rows.index((f, log))

        def bnd(b):
            off = len(bounds)
            bounds.extend(float(x) for x in b)
            return len(b), off
        arr = (capi.WdFeatSlot * max(len(slots), 1))()
        vocab_feats = {}
        for i, s in enumerate(slots):
            d = arr[i]
            d.kind, d.num_buckets = capi.WD_FEAT_KINDS[s.kind], int(s.num_buckets)
            if s.kind in ("hash", "vocab"):
                d.src = sidx(s.feature)
                if s.kind == "vocab":
                    if vocab_feats.setdefault(s.feature, i) != i:
                        raise ValueError("feature `%s` feeds two vocabulary columns" % s.feature)
            elif s.kind == "identity":
                d.src = iidx(s.feature)
            elif s.kind == "bucket":
                kind, p0, p1 = s.normalizer if s.normalizer else (None, 0.0, 1.0)
                d.src = fidx(s.feature, kind == "log")
                d.norm_kind = {None: 0, "min_max": 1, "standard": 2, "log": 0}[kind]
                d.p0, d.p1 = float(p0), float(p1)
                d.nbound, d.bound_off = bnd(s.boundaries)
            else:
                if len(s.cross_keys) > capi.WD_MAX_CROSS_KEYS:
                    raise ValueError("crossed column `%s` has more than %d keys" % (s.name, capi.WD_MAX_CROSS_KEYS))
                d.nkeys, d.hash_key = len(s.cross_keys), int(s.hash_key)
                for k, ck in enumerate(s.cross_keys):
                    kk = d.keys[k]
                    kk.kind = capi.WD_FEAT_KEY_KINDS[ck.kind]
                    if ck.kind == "string":
                        kk.src = sidx(ck.feature)
                    elif ck.kind == "identity":
                        kk.src, kk.num_buckets = iidx(ck.feature), int(ck.num_buckets)
                    else:       # bucketized RAW value (quirk C.5)
                        kk.src = fidx(ck.feature, False)
                        kk.nbound, kk.bound_off = bnd(ck.boundaries)
        dev = self.dev
        self.feat_slots_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
        self.bounds_dev = torch.tensor(bounds if bounds else [0.0], dtype=torch.float32, device=dev)
        self.vocab_dev = {}
        for i, (vb, vo, nv) in self.vocab_packed.items():
            self.vocab_dev[i] = (torch.from_numpy(vb).to(dev), torch.from_numpy(vo).to(dev), nv)
        # vocabulary per STRING FEATURE for the one-launch lookup (wd_feat_vocab_lookup_all)
        varr = (capi.WdFeatVocab * max(len(self.str_feats), 1))()
        for i, (vb, vo, nv) in self.vocab_dev.items():
            j = self.str_feats.index(slots[i].feature)
            varr[j].bytes, varr[j].offs, varr[j].nvocab = vb.data_ptr(), vo.data_ptr(), int(nv)
        self.vocab_table_dev = torch.from_numpy(np.frombuffer(bytes(varr), dtype=np.uint8).copy()).to(dev)
        self._fs = torch.cuda.Stream(device=dev)
        # the tables above were uploaded on the caller's stream: one edge, once, so that the featurizer's own stream -- which is
        # deliberately NOT ordered behind the training stream per batch -- never reads them before they have arrived
        self._fs.wait_stream(torch.cuda.current_stream(dev))

    def _stage(self, raw):
        """1. the parsed batch, as it is, in one staged buffer + ONE host-to-device copy (on the current stream)"""
        plan = self.plan
        B = raw.B
        F = len(self.str_feats)
        stg = _Stage()
        h = {"bytes": stg.add(raw.tok_bytes, np.uint8), "toffs": stg.add(raw.tok_offs, np.int32)}
        ex = np.zeros((max(F, 1), B + 1), dtype=np.int32)
        base = np.zeros(max(F, 1), dtype=np.int32)
        tokn = np.zeros(max(F, 1), dtype=np.int32)
        lmax = np.zeros(max(F, 1), dtype=np.int32)
        for j, f in enumerate(self.str_feats):
            pc = raw.cat[f]
            ex[j], base[j], tokn[j] = pc.ex_offs, pc.base, pc.n
            lmax[j] = int(np.diff(pc.ex_offs).max()) if B else 0
        h["ex"], h["base"], h["lmax"] = stg.add(ex, np.int32), stg.add(base, np.int32), stg.add(lmax, np.int32)
        h["tokn"], h["hdr"] = stg.add(tokn, np.int32), None
        ints = (np.stack([np.asarray(raw.ints[f], dtype=np.int64) for f in self.int_feats]) if self.int_feats
                else np.zeros((1, max(B, 1)), np.int64))
        rows = []
        for f, log in self.float_rows:
            x = np.asarray(raw.floats[f], dtype=np.float32)
            rows.append(np.log(x) if log else x)
        floats = np.stack(rows) if rows else np.zeros((1, max(B, 1)), np.float32)
        h["ints"], h["floats"] = stg.add(ints, np.int64), stg.add(floats, np.float32)
        nd = len(plan.dense_cols)
        h["dense"] = stg.add(np.stack([raw.floats[d.feature] for d in plan.dense_cols], axis=1), np.float32) if nd else None
        h["lab"] = stg.add(raw.labels, np.float32) if raw.labels is not None else None
        use_w = raw.weights is not None and self.engine.spec.use_weight_column
        h["wts"] = stg.add(raw.weights, np.float32) if use_w else None
        stg.upload(self.dev)
        return stg, h

    def resident(self, raw, ids_capacity=None):
        """The parsed batch staged into HBM + every buffer the featurizer's launches fill, allocated once: what `run` works on
        without allocating or waiting (a ParsedDeviceBatch can be featurized inside a captured step, like a synth.TokenBatch is
        hashed).  ids_capacity: entries of the id array (default: the engine's max_nnz); `pdb.batch.nnz` is that CAPACITY --
        the engine only uses it as a sizing hint -- until `finalize` reads the real count."""
        plan, eng = self.plan, self.engine
        if self.mode != "device":
            raise ValueError("Featurizer.resident needs mode='device'")
        B, S = raw.B, plan.S
        if B > eng.max_batch:
            raise ValueError("batch (B=%d) exceeds engine capacity (max_batch=%d)" % (B, eng.max_batch))
        cap = int(ids_capacity) if ids_capacity is not None else int(eng.max_nnz)
        if cap > eng.max_nnz:
            raise ValueError("ids_capacity %d exceeds engine capacity (max_nnz=%d)" % (cap, eng.max_nnz))
        pdb = ParsedDeviceBatch()
        pdb.B, pdb.T, pdb.raw = B, len(raw.tok_offs) - 2, raw
        pdb.stg, pdb.h = self._stage(raw)
        self._alloc_head(pdb)
        pdb.cap = cap
        pdb.ids = torch.zeros(max(cap, 1), dtype=torch.int32, device=self.dev)
        self._make_batch(pdb, nnz=cap, one_hot=False)
        return pdb

    def resident_fixed(self, B, tok_cap, bytes_cap, ids_capacity=None, labels=True, weights=False, nnz_hint=None):
        """A ParsedDeviceBatch over a FixedStage: fixed addresses, refilled batch after batch (`pdb.stg.fill(raw)`), featurized by
        `run` with the token count read on the device -- what a captured featurizer + train step works on.  nnz_hint: the engine's
        sizing hint for the id count (default: the capacity)."""
        eng = self.engine
        if self.mode != "device":
            raise ValueError("Featurizer.resident_fixed needs mode='device'")
        if B > eng.max_batch:
            raise ValueError("batch (B=%d) exceeds engine capacity (max_batch=%d)" % (B, eng.max_batch))
        cap = min(int(ids_capacity) if ids_capacity is not None else int(eng.max_nnz), int(eng.max_nnz))
        pdb = ParsedDeviceBatch()
        pdb.B, pdb.T, pdb.raw, pdb.dynamic = int(B), int(tok_cap), None, True
        pdb.stg = FixedStage(self, B, tok_cap, bytes_cap, labels=labels, weights=weights)
        pdb.h = pdb.stg.h
        self._alloc_head(pdb)
        pdb.q.ntok_dev = pdb.stg.ptr("hdr").value
        pdb.cap = cap
        pdb.ids = torch.zeros(max(cap, 1), dtype=torch.int32, device=self.dev)
        self._make_batch(pdb, nnz=min(int(nnz_hint), cap) if nnz_hint else cap, one_hot=False)
        return pdb

    def _alloc_head(self, pdb):
        stg, h, B, S, T = pdb.stg, pdb.h, pdb.B, self.plan.S, pdb.T
        i32 = dict(dtype=torch.int32, device=self.dev)
        pdb.fp = torch.empty(T + 1, dtype=torch.int64, device=self.dev)
        pdb.tok_val = torch.empty(max(T, 1), **i32) if self.vocab_dev else None
        n = B * S
        pdb.lens = torch.empty(max(n, 1), **i32)
        pdb.bag = torch.empty(n + 1, **i32)
        pdb.flags = torch.zeros(2, **i32)          # [0] overflow of the id array, [1] some bag does not hold exactly one id
        q = pdb.q = capi.WdFeatBatch()
        q.fp, q.tok_val = pdb.fp.data_ptr(), (pdb.tok_val.data_ptr() if pdb.tok_val is not None else None)
        q.ex_offs, q.tok_base = stg.ptr(h["ex"]).value, stg.ptr(h["base"]).value
        q.lmax = stg.ptr(h["lmax"]).value if self.cross_padding == "tf_dense" else None
        q.ints, q.floats, q.bounds = stg.ptr(h["ints"]).value, stg.ptr(h["floats"]).value, self.bounds_dev.data_ptr()
        q.batch, q.S, q.empty_index = B, S, T
        need = int(call("wd_feat_offsets_workspace_bytes", max(n, 1)))
        pdb.scan_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)   # per-block sums of wd_feat_lens (per batch: two may be in flight)

    def _make_batch(self, pdb, nnz, one_hot):
        stg, h, B, nd = pdb.stg, pdb.h, pdb.B, len(self.plan.dense_cols)
        dense = stg.tensor(h["dense"]).view(B, nd) if nd else None
        labels = stg.tensor(h["lab"]) if h.get("lab") is not None else None
        weights = stg.tensor(h["wts"]) if h.get("wts") is not None else None
        pdb.batch = DeviceBatch(B, pdb.ids, pdb.bag, dense, labels, weights, nnz=nnz, one_hot=one_hot)
        pdb.batch._keep = [pdb.fp, pdb.tok_val, pdb.lens, stg.dbuf, pdb.flags, pdb.scan_ws]
        return pdb.batch

    def _run_head(self, pdb, st):
        """2. fingerprints of every token (+ the trailing ''), vocabulary indices; 3a. lengths -> bag CSR"""
        plan, stg, h = self.plan, pdb.stg, pdb.h
        n = pdb.B * plan.S
        dyn = stg.ptr(h["hdr"]) if getattr(pdb, "dynamic", False) else None      # fixed-capacity stage: the token count is on the device
        if dyn is not None:
            call("wd_fingerprint64_dyn", stg.ptr(h["bytes"]), stg.ptr(h["toffs"]), pdb.T + 1, dyn, ptr(pdb.fp), st)
        else:
            call("wd_fingerprint64", stg.ptr(h["bytes"]), stg.ptr(h["toffs"]), pdb.T + 1, ptr(pdb.fp), st)
        if self.vocab_dev:      # every vocabulary_list column in one launch
            call("wd_feat_vocab_lookup_all", stg.ptr(h["bytes"]), stg.ptr(h["toffs"]), pdb.T, dyn, stg.ptr(h["base"]),
                 stg.ptr(h["tokn"]), len(self.str_feats), ptr(self.vocab_table_dev), ptr(pdb.tok_val), st)
        call("wd_feat_lens", ptr(self.feat_slots_dev), ctypes.byref(pdb.q), ptr(pdb.lens), ptr(pdb.scan_ws), st)
        call("wd_feat_offsets", ptr(pdb.lens), ptr(pdb.scan_ws), n, ptr(pdb.bag), pdb.cap, ptr(pdb.flags), st)

    def _run_emit(self, pdb, st):
        """3b. the ids, one lane per id (csrc/hash.hip k_feat_emit_par); grid sized by the batch: no id count needed here"""
        call("wd_feat_emit", ptr(self.feat_slots_dev), ctypes.byref(pdb.q), ptr(pdb.bag), ptr(pdb.ids), pdb.cap, st)

    def run(self, pdb):
        """Every launch of the featurizer for a resident batch, on the CURRENT stream: no allocation, no host wait (capturable).
        Returns pdb.batch (ids / bag offsets rewritten in place)."""
        st = torch.cuda.current_stream().cuda_stream
        self._run_head(pdb, st)
        self._run_emit(pdb, st)
        return pdb.batch

    def finalize(self, pdb):
        """One host read (id count, one-id-per-bag, overflow) behind a `run`: makes pdb.batch.nnz / .one_hot exact.  A captured
        step that replays `run` on the same resident batch keeps them valid (same tokens, same ids)."""
        stats = torch.cat([pdb.bag[-1:], pdb.flags]).cpu()
        nnz, over, multi = int(stats[0]), int(stats[1]), int(stats[2])
        if over or nnz > pdb.cap:
            raise ValueError("batch (B=%d, nnz=%d) exceeds the id capacity %d" % (pdb.B, nnz, pdb.cap))
        pdb.batch.nnz, pdb.batch.one_hot = nnz, (not multi) if pdb.B * self.plan.S else True
        return pdb.batch

    def _to_device_dev(self, raw):
        plan, eng = self.plan, self.engine
        B, S = raw.B, plan.S
        if B > eng.max_batch:
            raise ValueError("batch (B=%d) exceeds engine capacity (max_batch=%d)" % (B, eng.max_batch))
        cur = torch.cuda.current_stream()
        fs = self._fs       # NOT ordered behind `cur`: nothing here reads what the training stream writes, so the kernels below
        pdb = ParsedDeviceBatch()
        pdb.B, pdb.T, pdb.raw = B, len(raw.tok_offs) - 2, raw
        with torch.cuda.stream(fs):     # (and the host's wait for the id count) do not wait for the previous train step
            st = fs.cuda_stream
            pdb.stg, pdb.h = self._stage(raw)
            self._alloc_head(pdb)
            pdb.cap = int(eng.max_nnz)
            self._run_head(pdb, st)
            # the one host wait: this entry point sizes the id array exactly and tells the engine whether the batch is
            # one-id-per-bag (resident() / run() do neither and never wait)
            stats = torch.cat([pdb.bag[-1:], pdb.flags[1:]]).cpu()
            nnz, one_hot = int(stats[0]), (not int(stats[1])) if B * S else True
            if nnz > eng.max_nnz:
                raise ValueError("batch (B=%d, nnz=%d) exceeds engine capacity (max_batch=%d, max_nnz=%d)"
                                 % (B, nnz, eng.max_batch, eng.max_nnz))
            pdb.cap = nnz
            pdb.ids = torch.zeros(max(nnz, 1), dtype=torch.int32, device=self.dev)
            self._run_emit(pdb, st)
        cur.wait_stream(fs)
        for t in (pdb.ids, pdb.bag, pdb.stg.dbuf):            # allocated on the featurizer's stream, consumed on the caller's
            t.record_stream(cur)
        return self._make_batch(pdb, nnz=nnz, one_hot=one_hot)

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
        if self.mode == "device":
            return self._to_device_dev(raw)
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
