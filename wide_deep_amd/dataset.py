"""TSV input pipeline with the reference's `input_fn` contract (python/lib/dataset.py:23-195, 293-310).

    input_fn(csv_data_file, img_data_file, mode, batch_size) -> iterator of RawBatch

What the reference does with tf.data is restated on the host (plumbing before the hot path; a C++/GPU ingest is the
"next" row f1 of SURVEY section 8):
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

The image dataset (`img_data_file`, _ImageDataSet) is out of scope: a non-empty value raises.
"""
import os

import numpy as np

from .read_conf import Config


class RawBatch(object):
    """One batch of parsed rows (host memory).
    cat[f]    = (tokens: list of bytes, offs: int32[B+1])      string categorical features (ragged)
    ints[f]   = int32[B]                                       identity categorical features
    floats[f] = float32[B]                                     continuous features
    labels    = float32[B] or None (pred mode); weights = float32[B] or None
    """

    def __init__(self, B, cat, ints, floats, labels, weights):
        self.B, self.cat, self.ints, self.floats, self.labels, self.weights = B, cat, ints, floats, labels, weights

    def lmax(self, feature):
        """Width of the padded [B, Lmax] tensor padded_batch would build for a string feature."""
        offs = self.cat[feature][1]
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

    # ---- parsing ------------------------------------------------------------------------------------
    def _rows(self, is_pred):
        k = 0
        for path in self._files:
            with open(path, "rb") as fh:
                for line in fh:
                    if k % self._num_workers == self._worker_index:
                        yield line.rstrip(b"\r\n")
                    k += 1

    def _batch(self, lines, is_pred):
        B = len(lines)
        nf = len(self._fields)
        shift = 0 if is_pred else 1
        rows = []
        for ln in lines:
            parts = ln.split(b"\t")
            if len(parts) != nf + shift:
                raise ValueError("Expect %d fields but have %d in record" % (nf + shift, len(parts)))
            rows.append(parts)
        cat, ints, floats = {}, {}, {}
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
            cat[f] = (toks, offs)
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
        labels = weights = None
        if not is_pred:
            lab = np.zeros(B, dtype=np.float32)
            for b, parts in enumerate(rows):
                v = parts[0]
                lab[b] = 1.0 if (v not in (b"-", b"") and int(v) == 1) else 0.0
            labels = lab
            if self._use_weight:
                weights = np.where(lab > 0, np.float32(self._pos_w or 1), np.float32(self._neg_w or 1)).astype(np.float32)
        return RawBatch(B, cat, ints, floats, labels, weights)

    def input_fn(self, mode, batch_size):
        assert mode in ("train", "eval", "pred"), "mode must in `train`, `eval`, or `pred`, found %s" % mode
        is_pred = mode == "pred"
        it = self._rows(is_pred)
        if mode == "train":
            it = _buffer_shuffle(it, self._shuffle_buffer, seed=123)
        buf = []
        for ln in it:
            buf.append(ln)
            if len(buf) == batch_size:
                yield self._batch(buf, is_pred)
                buf = []
        if buf:
            yield self._batch(buf, is_pred)


def _buffer_shuffle(it, buffer_size, seed):
    """tf.data shuffle semantics: keep a buffer of `buffer_size` elements, emit a uniformly random one, refill."""
    rng = np.random.RandomState(seed)
    buf = []
    for x in it:
        if len(buf) < buffer_size:
            buf.append(x)
            continue
        j = rng.randint(len(buf))
        out, buf[j] = buf[j], x
        yield out
    while buf:
        j = rng.randint(len(buf))
        buf[j], buf[-1] = buf[-1], buf[j]
        yield buf.pop()


def input_fn(csv_data_file, img_data_file, mode, batch_size, conf=None):
    """Reference signature (python/lib/dataset.py:293-310).  Returns an iterator of RawBatch."""
    if img_data_file:
        raise NotImplementedError("image input (cnn tower) is out of scope of this engine (SURVEY section 2, row 14)")
    return CsvDataset(csv_data_file, conf).input_fn(mode, batch_size)
