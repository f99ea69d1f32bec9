"""Synthetic Criteo-shaped batches (SURVEY section 8(d) recipes) and host->HBM batch packing.

Host-side plumbing only: builds the example-major bag CSR the kernels consume, either directly as
ids or as raw string tokens that the hash kernels turn into ids on the device.
"""
import numpy as np
import torch

from .capi import call, ptr
from .engine import DeviceBatch


def bag_lengths(rng, B, S, mean_len, max_len=32):
    if mean_len <= 1:
        return np.ones((B, S), dtype=np.int64)
    # 1 + Poisson(mean-1), clipped (BASELINE config 4: avg 5 ids/slot)
    return np.clip(1 + rng.poisson(mean_len - 1.0, size=(B, S)), 1, max_len).astype(np.int64)


def zipf_ranks(rng, n, vocab, s=1.05):
    """Zipf(s) over `vocab` ranks via inverse-CDF on a truncated support."""
    w = 1.0 / np.power(np.arange(1, vocab + 1, dtype=np.float64), s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.searchsorted(cdf, rng.random(n), side="left").astype(np.int64)


def make_raw_batch(plan, B, seed=20260925, mean_len=1, dist="uniform", pos_rate=0.03, n_raw=None):
    """Host batch with RAW per-feature values (before hashing): dict with
       lens [B,S] int64, raw [nnz] int64 (raw categorical value per occurrence, example-major),
       dense [B,nd] f32, labels [B] f32."""
    rng = np.random.default_rng(seed)
    S = plan.S
    lens = bag_lengths(rng, B, S, mean_len)
    nnz = int(lens.sum())
    slot_of = np.repeat(np.tile(np.arange(S), B), lens.reshape(-1))
    nb = np.asarray([s.num_buckets for s in plan.slots], dtype=np.int64)
    hi = nb[slot_of] if n_raw is None else np.full(nnz, n_raw, dtype=np.int64)
    if dist == "uniform":
        raw = (rng.random(nnz) * hi).astype(np.int64)
    elif dist == "zipf":
        vocab = int(hi.max())
        ranks = zipf_ranks(rng, nnz, vocab)
        perm_rng = np.random.default_rng(12345)
        perm = perm_rng.permutation(vocab)
        raw = perm[ranks] % hi
    else:
        raise ValueError(dist)
    nd = len(plan.dense_cols)
    dense = rng.standard_normal((B, nd)).astype(np.float32) if nd else None
    labels = (rng.random(B) < pos_rate).astype(np.float32)
    return {"B": B, "lens": lens, "raw": raw, "dense": dense, "labels": labels}


def offsets_from_lens(lens):
    offs = np.zeros(lens.size + 1, dtype=np.int32)
    np.cumsum(lens.reshape(-1), out=offs[1:])
    return offs


def to_device_ids(plan, hb, device="cuda", weights=None):
    """Treat raw values as already-bucketed ids (raw % num_buckets)."""
    S = plan.S
    lens = hb["lens"]
    slot_of = np.repeat(np.tile(np.arange(S), hb["B"]), lens.reshape(-1))
    nb = np.asarray([s.num_buckets for s in plan.slots], dtype=np.int64)
    ids = (hb["raw"] % nb[slot_of]).astype(np.int32)
    offs = offsets_from_lens(lens)
    t = lambda a, dt: torch.as_tensor(a, dtype=dt).to(device) if a is not None else None
    return DeviceBatch(hb["B"], t(ids, torch.int32), t(offs, torch.int32), t(hb["dense"], torch.float32),
                       t(hb["labels"], torch.float32), t(weights, torch.float32), nnz=len(ids),
                       one_hot=bool((lens == 1).all()))


def pack_decimal_tokens(raw):
    """raw int64 [n] -> (uint8 bytes, int32 offs[n+1]) of their decimal strings."""
    strs = np.char.mod("%d", raw)
    lens = np.char.str_len(strs).astype(np.int32)
    offs = np.zeros(len(raw) + 1, dtype=np.int32)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer("".join(strs.tolist()).encode(), dtype=np.uint8).copy()
    return data, offs


class TokenBatch:
    """Raw string tokens resident in HBM + the buffers the hash kernel fills."""

    def __init__(self, plan, hb, device="cuda", weights=None):
        self.B = hb["B"]
        data, tok_offs = pack_decimal_tokens(hb["raw"])
        lens = hb["lens"]
        self.ntok = len(hb["raw"])
        self.one_per_bag = bool((lens == 1).all())
        self.bytes = torch.from_numpy(data).to(device)
        self.tok_offs = torch.from_numpy(tok_offs).to(device)
        offs = offsets_from_lens(lens)
        self.bag_offs = torch.from_numpy(offs).to(device)
        t = lambda a: torch.as_tensor(a, dtype=torch.float32).to(device) if a is not None else None
        self.ids = torch.zeros(max(self.ntok, 1), dtype=torch.int32, device=device)
        self.batch = DeviceBatch(self.B, self.ids, self.bag_offs, t(hb["dense"]), t(hb["labels"]), t(weights),
                                 nnz=self.ntok, one_hot=self.one_per_bag)
        if self.one_per_bag:        # slot-major copy of the ids for the one-launch bucketing (wd_hash_bucket_cols)
            self.batch.ids_cols = torch.zeros(max(self.ntok, 1), dtype=torch.int32, device=device)


class FeaturizedBatch:
    """A synthetic batch whose ids the Featurizer has already produced (features.Featurizer.to_device on a make_parsed_batch
    batch: hash slots AND crossed columns) -- stands where a TokenBatch stands (`.batch`), with nothing left to hash."""

    def __init__(self, batch, host):
        self.batch, self.B, self.host = batch, batch.B, host


class ParsedTokenBatch:
    """A synthetic PARSED batch resident in HBM (features.Featurizer.resident on a make_parsed_batch batch: token bytes, per-feature
    example ranges, floats, labels) -- stands where a TokenBatch stands: `hash_tokens` runs the whole device featurizer on it
    (fingerprints -> bag lengths -> bag CSR -> hash-bucket ids AND crossed columns), in the step, with no host wait.
    Constructed with one eager featurizer run + `finalize`, so that `.batch.nnz` / `.one_hot` are exact for the captures."""

    def __init__(self, featurizer, raw, host, ids_capacity=None):
        self.fz, self.host, self.B = featurizer, host, raw.B
        self.pdb = featurizer.resident(raw, ids_capacity)
        featurizer.run(self.pdb)
        self.batch = featurizer.finalize(self.pdb)


def hash_tokens(engine, tb, phase="all"):
    """tokens -> ids on the device (wd_hash_bucket): a4 of SURVEY section 8.  phase: "all", or -- a pipelined capture that wants the two
    halves of a parsed batch's featurizer on two streams -- "head" (fingerprints, bag lengths, bag CSR: memory-light, fits beside the
    tower) then "emit" (the ids: 64-bit integer VALU, slow beside the tower); a plain TokenBatch is hashed whole by "head"."""
    if isinstance(tb, FeaturizedBatch):
        return tb.batch
    if isinstance(tb, ParsedTokenBatch):     # a4 + a5: hash buckets and crossed columns of a parsed batch (features.Featurizer.run)
        if phase == "all":
            return tb.fz.run(tb.pdb)
        st = torch.cuda.current_stream().cuda_stream
        if phase == "head":
            tb.fz._run_head(tb.pdb, st)
        else:
            tb.fz._run_emit(tb.pdb, st)
        return tb.batch
    if phase == "emit":
        return tb.batch
    plan = getattr(engine, "hash_plan", engine.plan)            # sharded engines hash in the global id space
    slots_dev = getattr(engine, "hash_slots_dev", engine.slots_dev)
    st = torch.cuda.current_stream().cuda_stream
    if tb.batch.ids_cols is not None and tb.ntok == tb.B * plan.S:
        call("wd_hash_bucket_cols", ptr(tb.bytes), ptr(tb.tok_offs), tb.ntok, ptr(slots_dev), plan.S, ptr(tb.ids),
             ptr(tb.batch.ids_cols), st)
        tb.batch.ids_cols_valid = True
        return tb.batch
    tb.batch.ids_cols_valid = False      # this entry point writes the example-major ids only: a slot-major copy is stale now
    call("wd_hash_bucket", ptr(tb.bytes), ptr(tb.tok_offs), tb.ntok, None if tb.one_per_bag else ptr(tb.bag_offs),
         tb.B * plan.S, ptr(slots_dev), plan.S, ptr(tb.ids), st)
    return tb.batch


def make_parsed_batch(plan, B, seed=20260925, mean_len=1, dist="uniform", pos_rate=0.03, weights=None):
    """A synthetic batch in the form dataset.input_fn hands to the Featurizer (dataset.RawBatch: packed string tokens per
    feature, float features, labels) -- for models whose columns need more than one hash per token: crossed columns over
    multi-valued string features (BASELINE configs[3]; python/lib/build_estimator.py:138-155).  The string features are the
    hash slots' (one decimal token per occurrence, lengths 1 + Poisson(mean_len - 1)); crosses read the same tokens.
    Returns (RawBatch, hb) with hb = {"lens" [B, n_features], "raw" per feature, "dense", "labels"} for the oracle."""
    from .dataset import PackedTokens, RawBatch
    rng = np.random.default_rng(seed)
    feats = []
    for s in plan.slots:
        if s.kind == "hash" and s.feature not in feats:
            feats.append(s.feature)
    nb = {s.feature: int(s.num_buckets) for s in plan.slots if s.kind == "hash"}
    F = len(feats)
    lens = bag_lengths(rng, B, F, mean_len)                        # [B, F]
    toks, per_feat, base = [], [], 0
    raw_of = {}
    perm = None
    for j, f in enumerate(feats):
        n = int(lens[:, j].sum())
        if dist == "uniform":
            raw = (rng.random(n) * nb[f]).astype(np.int64)
        elif dist == "zipf":
            ranks = zipf_ranks(rng, n, nb[f])
            if perm is None or len(perm) != nb[f]:
                perm = np.random.default_rng(12345).permutation(nb[f])
            raw = perm[ranks]
        else:
            raise ValueError(dist)
        raw_of[f] = raw
        ex = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(lens[:, j], out=ex[1:])
        toks.append(raw)
        per_feat.append((f, base, n, ex))
        base += n
    allraw = np.concatenate(toks) if toks else np.zeros(0, np.int64)
    data, toffs = pack_decimal_tokens(allraw)
    # the Featurizer's buffers end in one empty token (the '' a missing value parses to) and a NUL byte
    tok_offs = np.concatenate([toffs, toffs[-1:]]).astype(np.int32)
    tok_bytes = np.concatenate([data, np.zeros(1, np.uint8)])
    cat = {f: PackedTokens(tok_bytes, tok_offs, b0, n, ex) for f, b0, n, ex in per_feat}
    nd = len(plan.dense_cols)
    dense = rng.standard_normal((B, nd)).astype(np.float32) if nd else None
    floats = {d.feature: np.ascontiguousarray(dense[:, i]) for i, d in enumerate(plan.dense_cols)}
    labels = (rng.random(B) < pos_rate).astype(np.float32)
    w = None
    if weights is not None:
        w = np.where(labels > 0, np.float32(weights[0]), np.float32(weights[1])).astype(np.float32)
    raw = RawBatch(B, cat, {}, floats, labels, w, tok_bytes, tok_offs)
    nnz = 0          # ids the Featurizer will emit (ragged crosses: the product of the key features' counts per example)
    col = {f: j for j, f in enumerate(feats)}
    for s in plan.slots:
        if s.kind == "hash":
            nnz += int(lens[:, col[s.feature]].sum())
        elif s.kind == "cross":
            nnz += int(np.prod(np.stack([lens[:, col[k.feature]] for k in s.cross_keys], axis=1), axis=1).sum())
    return raw, {"B": B, "features": feats, "lens": lens, "raw": raw_of, "dense": dense, "labels": labels, "weights": w, "nnz": nnz}
