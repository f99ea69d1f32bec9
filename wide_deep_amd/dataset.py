"""TSV input pipeline with the reference's `input_fn` contract (python/lib/dataset.py:23-195, 293-310).

    input_fn(csv_data_file, img_data_file, mode, batch_size) -> iterator of RawBatch

What the reference does with tf.data is restated for whole batches:
  * TextLine -> decode_csv(field_delim='\\t', use_quote_delim=False, na_value='-') with per-field defaults:
    used string categories '' , identity categories int 0, continuous 0.0, label int 0 (dataset.py:86-105);
    an empty field or '-' takes the default;
  * multivalue mode (train.yaml `multivalue: 1`): string features are split on ',' with empty pieces skipped
    (tf.string_split default), everything else stays one value per example (dataset.py:145-152);
  * label = (clk == 1); optional weight column = pos / neg loss weight when BOTH are set (dataset.py:70-72,159-163);
  * train mode shuffles with buffer `num_examples` and seed 123 (dataset.py:181-182).  TF's shuffle order cannot be
    reproduced outside TF; this is a seeded buffer shuffle with the same buffer semantics;
  * batches keep the ragged token lists as CSR (the padded_batch of the reference is reconstructed where it matters:
    RawBatch.lmax(feature) gives the padded width that crossed columns see);
  * `is_distribution`: every worker reads lines i % num_workers == worker_index (dataset.py:74-79,173-174); when
    torch.distributed is initialised instead, rank / world_size are used the same way.

Ingest: files are read as raw bytes and a batch of lines is parsed by the C library
csrc/tsv_ingest.c (two passes, no Python object per field or token); every string feature's tokens land directly in ONE
packed byte buffer + offsets -- the layout the GPU hash kernel reads.  `WD_PY_INGEST=1` (or a missing library) selects the
pure-Python parser of the same semantics, which tests keep as a cross-check.

The image dataset (`img_data_file`, _ImageDataSet) is out of scope: a non-empty value raises.
"""
import ctypes
import functools
import os

import numpy as np

from .read_conf import Config

_HERE = os.path.dirname(os.path.abspath(__file__))
_INGEST_PATH = os.path.join(_HERE, "_lib", "libwd_ingest.so")
_ingest = None


def ingest_lib():
    """libwd_ingest.so or None (then the Python parser runs)."""
    global _ingest
    if _ingest is None:
        if os.environ.get("WD_PY_INGEST") or not os.path.exists(_INGEST_PATH):
            _ingest = False
        else:
            L = ctypes.CDLL(_INGEST_PATH)
            L.wd_tsv_scan.restype = ctypes.c_int64
            L.wd_tsv_count.restype = ctypes.c_int
            L.wd_tsv_fill.restype = ctypes.c_int
            L.wd_vocab_lookup.restype = None
            _ingest = L
    return _ingest or None


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


class PackedTokens(object):
    """Tokens of ONE string feature for a batch: a window of the batch's shared packed buffers.
    bytes / tok_offs are shared by all features (tok_offs absolute into bytes); this feature owns tokens
    [base, base + n); ex_offs[B+1] indexes them per example (relative to base).
    Iterating yields (list_of_token_bytes, ex_offs) for code that wants Python objects (tests)."""

    def __init__(self, tok_bytes, tok_offs, base, n, ex_offs):
        self.bytes, self.tok_offs, self.base, self.n, self.ex_offs = tok_bytes, tok_offs, int(base), int(n), ex_offs

    def tokens(self):
        o, b = self.tok_offs, self.bytes.tobytes()
        return [b[o[t]: o[t + 1]] for t in range(self.base, self.base + self.n)]

    def __iter__(self):
        yield self.tokens()
        yield self.ex_offs

    def __getitem__(self, i):
        return (self.tokens(), self.ex_offs)[i]


class RawBatch(object):
    """One batch of parsed rows (host memory).
    cat[f]    = PackedTokens                                   string categorical features (ragged)
    ints[f]   = int32[B]                                       identity categorical features
    floats[f] = float32[B]                                     continuous features
    labels    = float32[B] or None (pred mode); weights = float32[B] or None
    tok_bytes / tok_offs: the packed token buffer of ALL string features (feature-major), ready for wd_fingerprint64
    """

    def __init__(self, B, cat, ints, floats, labels, weights, tok_bytes=None, tok_offs=None):
        self.B, self.cat, self.ints, self.floats, self.labels, self.weights = B, cat, ints, floats, labels, weights
        self.tok_bytes, self.tok_offs = tok_bytes, tok_offs
        # the C parser's own arrays, feature-major (features.FixedStage.fill packs a batch from them in a dozen array copies
        # instead of a handful per feature): {"str": (names, ex_offs [n][B + 1], tok_base [n], ntok [n]), "int": (names, [n][B]),
        # "flt": (names, [n][B])}, or None (the pure-Python parser)
        self.packed = None

    def lmax(self, feature):
        """Width of the padded [B, Lmax] tensor padded_batch would build for a string feature."""
        offs = self.cat[feature].ex_offs
        return int(np.max(np.diff(offs))) if self.B else 0


def list_files(path):
    """Directory -> sorted file list without dotfiles; a file -> [file]  (python/lib/utils/util.py:36-45)."""
    if os.path.isdir(path):
        return sorted(os.path.join(path, f) for f in os.listdir(path) if not f.startswith("."))
    return [path]


class CsvDataset(object):
    def __init__(self, data_file, conf=None):
        if not os.path.exists(data_file):
            raise AssertionError("data file: %s not found. Please check input data path" % data_file)
        self._files = list_files(data_file)
        self._conf = conf or Config()
        train = self._conf.train
        dist = self._conf.distribution
        self._shuffle_buffer = int(train["num_examples"])
        self._pos_w, self._neg_w = train["pos_sample_loss_weight"], train["neg_sample_loss_weight"]
        self._use_weight = self._pos_w is not None and self._neg_w is not None
        self._multivalue = bool(train["multivalue"])
        self.num_parallel_calls = int(train.get("num_parallel_calls") or 0)      # conf/train.yaml:55 (unset = default)
        self._num_workers, self._worker_index = 1, 0
        if dist.get("is_distribution"):
            cluster = dist["cluster"]
            self._num_workers = 1 + len(cluster["worker"])
            self._worker_index = dist["task_index"] if dist["job_name"] == "worker" else self._num_workers - 1
        else:
            try:
                import torch.distributed as td
                if td.is_available() and td.is_initialized():
                    self._num_workers, self._worker_index = td.get_world_size(), td.get_rank()
            except ImportError:
                pass
        schema = self._conf.read_schema()
        feature_conf = self._conf.read_feature_conf()
        self._feature_conf = feature_conf
        # field layout: position 0 = label, then every schema field in order (used or not)
        names = [schema[k] for k in sorted(schema)]
        if names[0] != "clk":
            raise ValueError("schema.yaml: column 1 must be the label `clk`")
        self._fields = names[1:]
        self._str_feats, self._int_feats, self._flt_feats = [], [], []   # (name, column index among non-label fields)
        for i, f in enumerate(self._fields):
            c = feature_conf.get(f)
            if c is None:
                continue
            if c["type"] == "category":
                (self._int_feats if c["transform"] == "identity" else self._str_feats).append((f, i))
            else:
                self._flt_feats.append((f, i))
        self._str_names = tuple(f for f, _ in self._str_feats)
        self._int_names = tuple(f for f, _ in self._int_feats)
        self._flt_names = tuple(f for f, _ in self._flt_feats)

    # ---- raw bytes + line index ----------------------------------------------------------------------
    def _load(self):
        """All files as one byte buffer (newline-terminated) and this worker's line (start, end) offsets."""
        chunks = []
        for path in self._files:
            with open(path, "rb") as fh:
                b = fh.read()
            if b and not b.endswith(b"\n"):
                b += b"\n"
            chunks.append(b)
        buf = np.frombuffer(b"".join(chunks), dtype=np.uint8)
        nl = np.flatnonzero(buf == 10).astype(np.int64)
        starts = np.concatenate([np.zeros(1, np.int64), nl[:-1] + 1]) if len(nl) else np.zeros(0, np.int64)
        ends = nl + 1
        if self._num_workers > 1:
            # every rank gets the same number of lines (the last n % workers lines are dropped): the ranks of the sharded
            # engine meet in collectives every step, so they must see the same number of equally sized batches
            per = len(starts) // self._num_workers
            keep = np.flatnonzero(np.arange(len(starts)) % self._num_workers == self._worker_index)[:per]
            starts, ends = starts[keep], ends[keep]
        return buf, starts, ends

    # ---- parsing: C library ----------------------------------------------------------------------------
    def _batch_c(self, L, buf, starts, ends, is_pred):
        B = len(starts)
        shift = 0 if is_pred else 1
        nfields = len(self._fields) + shift
        sc = np.asarray([i + shift for _, i in self._str_feats], dtype=np.int32)
        ic = np.asarray([i + shift for _, i in self._int_feats], dtype=np.int32)
        fc = np.asarray([i + shift for _, i in self._flt_feats], dtype=np.int32)
        ns, ni, nf = len(sc), len(ic), len(fc)
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        ends = np.ascontiguousarray(ends, dtype=np.int64)
        ntok, nbytes = np.zeros(max(ns, 1), np.int64), np.zeros(max(ns, 1), np.int64)
        err = ctypes.c_int64(-1)
        rc = L.wd_tsv_count(_p(buf), _p(starts), _p(ends), ctypes.c_int64(B), ctypes.c_int32(nfields), _p(sc),
                            ctypes.c_int32(ns), ctypes.c_int32(1 if self._multivalue else 0), _p(ntok), _p(nbytes),
                            ctypes.byref(err))
        if rc != 0:
            raise ValueError("Expect %d fields but have a different count in record %d of the batch" % (nfields, err.value))
        tok_base = np.zeros(max(ns, 1), np.int64)
        byte_base = np.zeros(max(ns, 1), np.int64)
        if ns:
            tok_base[1:] = np.cumsum(ntok)[:-1]
            byte_base[1:] = np.cumsum(nbytes)[:-1]
        T, NB = int(ntok[:ns].sum()), int(nbytes[:ns].sum())
        tok_bytes = np.zeros(NB + 1, np.uint8)
        tok_offs = np.zeros(T + 2, np.int32)          # + one trailing '' token used as padding by crossed columns
        ex_offs = np.zeros((max(ns, 1), B + 1), np.int32)
        ints = np.zeros((max(ni, 1), B), np.int32)
        flts = np.zeros((max(nf, 1), B), np.float32)
        labels = np.zeros(B, np.float32)
        rc = L.wd_tsv_fill(_p(buf), _p(starts), _p(ends), ctypes.c_int64(B), ctypes.c_int32(nfields), _p(sc),
                           ctypes.c_int32(ns), ctypes.c_int32(1 if self._multivalue else 0), _p(tok_base), _p(byte_base),
                           _p(tok_bytes), _p(tok_offs), _p(ex_offs), _p(ic), ctypes.c_int32(ni), _p(ints), _p(fc),
                           ctypes.c_int32(nf), _p(flts), ctypes.c_int32(-1 if is_pred else 0), _p(labels),
                           ctypes.byref(err))
        if rc != 0:
            kind = {-1: "field count", -2: "integer field", -3: "float field"}.get(rc, "parse")
            raise ValueError("TSV %s error in record %d of the batch" % (kind, err.value))
        tok_offs[T] = NB
        tok_offs[T + 1] = NB
        cat = {f: PackedTokens(tok_bytes, tok_offs, tok_base[j], ntok[j], ex_offs[j]) for j, (f, _) in enumerate(self._str_feats)}
        raw = self._finish(B, cat, {f: ints[j] for j, (f, _) in enumerate(self._int_feats)},
                           {f: flts[j] for j, (f, _) in enumerate(self._flt_feats)}, None if is_pred else labels,
                           tok_bytes, tok_offs)
        raw.packed = {"str": (self._str_names, ex_offs, tok_base, ntok), "int": (self._int_names, ints), "flt": (self._flt_names, flts)}
        return raw

    # ---- parsing: pure Python (cross-check path) -------------------------------------------------------
    def _batch_py(self, buf, starts, ends, is_pred):
        raw = buf.tobytes()
        lines = [raw[s:e].rstrip(b"\r\n") for s, e in zip(starts.tolist(), ends.tolist())]
        B = len(lines)
        nf = len(self._fields)
        shift = 0 if is_pred else 1
        rows = []
        for ln in lines:
            parts = ln.split(b"\t")
            if len(parts) != nf + shift:
                raise ValueError("Expect %d fields but have %d in record" % (nf + shift, len(parts)))
            rows.append(parts)
        all_toks, per_feat = [], []
        for f, i in self._str_feats:
            toks, offs = [], np.zeros(B + 1, dtype=np.int32)
            for b, parts in enumerate(rows):
                v = parts[i + shift]
                if v != b"-" and v != b"":
                    if self._multivalue:
                        toks.extend(p for p in v.split(b",") if p)
                    else:
                        toks.append(v)
                offs[b + 1] = len(toks)
            per_feat.append((f, len(all_toks), len(toks), offs))
            all_toks.extend(toks)
        lens = np.fromiter((len(t) for t in all_toks), dtype=np.int64, count=len(all_toks))
        tok_offs = np.zeros(len(all_toks) + 2, np.int32)
        np.cumsum(lens, out=tok_offs[1:len(all_toks) + 1])
        tok_offs[len(all_toks) + 1] = tok_offs[len(all_toks)]
        tok_bytes = np.frombuffer(b"".join(all_toks) + b"\0", dtype=np.uint8).copy()
        cat = {f: PackedTokens(tok_bytes, tok_offs, base, n, offs) for f, base, n, offs in per_feat}
        ints, floats = {}, {}
        for f, i in self._int_feats:
            a = np.zeros(B, dtype=np.int32)
            for b, parts in enumerate(rows):
                v = parts[i + shift]
                if v != b"-" and v != b"":
                    a[b] = int(v)
            ints[f] = a
        for f, i in self._flt_feats:
            a = np.zeros(B, dtype=np.float32)
            for b, parts in enumerate(rows):
                v = parts[i + shift]
                if v != b"-" and v != b"":
                    a[b] = np.float32(float(v))
            floats[f] = a
        labels = None
        if not is_pred:
            labels = np.zeros(B, dtype=np.float32)
            for b, parts in enumerate(rows):
                v = parts[0]
                labels[b] = 1.0 if (v not in (b"-", b"") and int(v) == 1) else 0.0
        return self._finish(B, cat, ints, floats, labels, tok_bytes, tok_offs)

    def _finish(self, B, cat, ints, floats, labels, tok_bytes, tok_offs):
        weights = None
        if labels is not None and self._use_weight:
            weights = np.where(labels > 0, np.float32(self._pos_w or 1), np.float32(self._neg_w or 1)).astype(np.float32)
        return RawBatch(B, cat, ints, floats, labels, weights, tok_bytes, tok_offs)

    def batch_jobs(self, mode, batch_size):
        """one thunk per batch, in order; calling it parses that batch (thread-safe: a thunk only reads the file buffer)"""
        assert mode in ("train", "eval", "pred"), "mode must in `train`, `eval`, or `pred`, found %s" % mode
        is_pred = mode == "pred"
        buf, starts, ends = self._load()
        order = np.arange(len(starts))
        if mode == "train":
            order = _buffer_shuffle(len(starts), self._shuffle_buffer, seed=123)
        L = ingest_lib()
        for b0 in range(0, len(order), batch_size):
            idx = order[b0: b0 + batch_size]
            if L is not None:
                yield functools.partial(self._batch_c, L, buf, starts[idx], ends[idx], is_pred)
            else:
                yield functools.partial(self._batch_py, buf, starts[idx], ends[idx], is_pred)

    def input_fn(self, mode, batch_size):
        for job in self.batch_jobs(mode, batch_size):
            yield job()


def _buffer_shuffle(n, buffer_size, seed):
    """tf.data shuffle semantics on line indices: keep a buffer of `buffer_size` elements, emit a uniformly random one,
    refill with the next input; a dataset that fits the buffer is a plain uniform permutation."""
    rng = np.random.RandomState(seed)
    if n <= buffer_size:
        return rng.permutation(n)
    out = np.empty(n, dtype=np.int64)
    buf = list(range(buffer_size))
    k = 0
    for x in range(buffer_size, n):
        j = rng.randint(buffer_size)
        out[k] = buf[j]
        buf[j] = x
        k += 1
    tail = np.asarray(buf, dtype=np.int64)
    rng.shuffle(tail)
    out[k:] = tail
    return out


def prefetched(jobs, depth=2, workers=1):
    """The reference ends its tf.data pipeline with `.prefetch(...)` (python/lib/dataset.py:185, 281) and offers
    `num_parallel_calls` for the parser (`:177`, conf/train.yaml:55): batches are parsed ahead of the consumer.  Here: `jobs`
    yields one thunk per batch; up to `depth` of them are in flight on `workers` threads (the C ingest runs without the GIL),
    results come back IN ORDER, so the TSV parse overlaps the featurizer and the train step of earlier batches.  Order and
    content are those of calling the thunks one after the other."""
    from concurrent.futures import ThreadPoolExecutor
    from collections import deque
    pool, pending = ThreadPoolExecutor(max(int(workers), 1), thread_name_prefix="wd_ingest"), deque()
    try:
        for job in jobs:
            pending.append(pool.submit(job))
            if len(pending) >= depth:
                yield pending.popleft().result()      # a parser error surfaces here, at the batch it belongs to
        while pending:
            yield pending.popleft().result()
    finally:                                          # consumer stopped early (train(steps=...)) or an error: drop the rest
        for f in pending:
            f.cancel()
        pool.shutdown(wait=True)


def input_fn(csv_data_file, img_data_file, mode, batch_size, conf=None, prefetch=None):
    """Reference signature (python/lib/dataset.py:293-310).  Returns an iterator of RawBatch.
    prefetch: batches parsed ahead of the consumer (env WD_PREFETCH; 0 = parse in the consumer's thread) on
    `num_parallel_calls` threads (conf/train.yaml:55, env WD_INGEST_THREADS; unset = up to 4: the C parser runs without the
    GIL and 4 threads parse 1.5 M rows/s instead of 0.7 M -- with the featurizer on the device the consumer keeps up: the C1
    loop at batch 8192 goes from 0.78 M to 2.18 M examples/s, profiles/r2zz_c1_*.json)."""
    if prefetch is None:
        prefetch = int(os.environ.get("WD_PREFETCH", "2"))
    if img_data_file:
        raise NotImplementedError("image input (cnn tower) is out of scope of this engine (SURVEY section 2, row 14)")
    ds = CsvDataset(csv_data_file, conf)
    if not prefetch:
        return ds.input_fn(mode, batch_size)
    workers = int(os.environ.get("WD_INGEST_THREADS", "0")) or ds.num_parallel_calls or max(1, min(4, os.cpu_count() or 1))
    return prefetched(ds.batch_jobs(mode, batch_size), depth=max(prefetch, 2 * workers), workers=workers)
