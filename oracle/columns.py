"""oracle/columns.py -- TEST INFRASTRUCTURE ONLY (the checker, never the product).

CPU restatement of the part of the path that sits BEFORE the model: TSV row -> feature tensors -> per-column
sparse ids, i.e. what the reference builds with tf.data + tf.feature_column:

  * row parsing            python/lib/dataset.py:86-164   (decode_csv defaults, '-' = NA, ',' multi-value split,
                                                           label = clk == 1, weight column)
  * column wiring          python/lib/build_estimator.py:49-169  (hash / vocab / identity / numeric / bucketized /
                                                           crossed columns, embedding_dim rule, is_deep)
  * TF column semantics    SURVEY Appendix A.2-A.6 and quirks C.5, C.10, C.16

It shares NO code with wide_deep_amd/ (own parser, own wiring), uses plain numpy plus the C hash functions of
wd_oracle.c, and emits the neutral structures `oracle.OracleWideDeep` consumes.
"""
import math

import numpy as np
import yaml

from . import oracle as O


def embedding_dim(n):
    # python/lib/build_estimator.py:57-59 (natural log)
    return int(2 ** math.ceil(math.log(n ** 0.25)))


class Columns(object):
    """Wiring of one conf directory (feature.yaml, cross_feature.yaml, schema.yaml, train.yaml, model.yaml)."""

    def __init__(self, conf_dir, model_type=None, global_embedding_dim=None):
        rd = lambda n: yaml.safe_load(open("%s/%s.yaml" % (conf_dir, n)))
        self.schema = rd("schema")
        self.features = rd("feature")
        self.crosses = rd("cross_feature")
        self.train = rd("train")["train"]
        self.model = rd("model")
        self.model_type = model_type or self.train["model_type"]
        self.fields = [str(self.schema[k]).lower() for k in sorted(self.schema)][1:]   # without the label column
        gdim = global_embedding_dim or self.model.get("embedding_dim")
        self.wide_cols, self.deep_cols, self.recipes = [], [], {}
        for f, c in self.features.items():
            t, p = c["transform"], c["parameter"]
            if c["type"] == "category":
                if t == "hash_bucket":
                    self.recipes[f] = ("hash", f, int(p))
                    self.wide_cols.append({"name": f, "key": f, "num_buckets": int(p)})
                    self.deep_cols.append({"name": f + "_embedding", "kind": "embedding", "key": f, "num_buckets": int(p),
                                           "dim": int(c.get("embedding_dim") or gdim or embedding_dim(p))})
                elif t == "vocab":
                    vocab = [str(v) for v in p]
                    self.recipes[f] = ("vocab", f, vocab)
                    self.wide_cols.append({"name": f, "key": f, "num_buckets": len(vocab)})
                    self.deep_cols.append({"name": f + "_indicator", "kind": "indicator", "key": f, "num_buckets": len(vocab), "dim": 0})
                else:
                    self.recipes[f] = ("identity", f, int(p))
                    self.wide_cols.append({"name": f, "key": f, "num_buckets": int(p)})
                    self.deep_cols.append({"name": f + "_indicator", "kind": "indicator", "key": f, "num_buckets": int(p), "dim": 0})
            else:
                norm, bounds = p["normalization"], p["boundaries"]
                self.recipes[f] = ("numeric", f, t, norm)
                self.deep_cols.append({"name": f, "kind": "numeric", "key": f, "num_buckets": 0, "dim": 1})
                if bounds:
                    key = f + "_bucketized"
                    self.recipes[key] = ("bucket_norm", f, t, norm, [float(b) for b in bounds])
                    self.wide_cols.append({"name": key, "key": key, "num_buckets": len(bounds) + 1})
        for name, c in self.crosses.items():
            parts = [x.strip() for x in name.split("&")]
            size = 1000 * c["hash_bucket_size"] or 10000
            deep = c["is_deep"] if c.get("is_deep") is not None else 1
            keys, names = [], []
            for x in parts:
                fc = self.features[x]
                if fc["type"] == "continuous":
                    keys.append(("bucket_raw", x, [float(b) for b in fc["parameter"]["boundaries"]]))
                    names.append(x + "_bucketized")
                elif fc["transform"] == "identity":
                    keys.append(("identity", x, int(fc["parameter"])))
                    names.append(x)
                else:
                    keys.append(("string", x))
                    names.append(x)
            cname = "_X_".join(sorted(names))
            self.recipes[cname] = ("cross", keys, int(size))
            self.wide_cols.append({"name": cname, "key": cname, "num_buckets": int(size)})
            if deep:
                self.deep_cols.append({"name": cname + "_embedding", "kind": "embedding", "key": cname, "num_buckets": int(size),
                                       "dim": int(gdim or embedding_dim(size))})
        pw, nw = self.train["pos_sample_loss_weight"], self.train["neg_sample_loss_weight"]
        self.use_weight = pw is not None and nw is not None      # produced only if BOTH are set (dataset.py:70-72)
        self.pos_w, self.neg_w = (pw or 1), (nw or 1)

    # ---- rows -> feature tensors (dataset.py:133-164) ------------------------------------------------
    def parse(self, lines, is_pred=False):
        multivalue = bool(self.train["multivalue"])
        B = len(lines)
        strs = {f: [] for f, c in self.features.items() if c["type"] == "category" and c["transform"] != "identity"}
        ints = {f: np.zeros(B, np.int64) for f, c in self.features.items() if c["type"] == "category" and c["transform"] == "identity"}
        flts = {f: np.zeros(B, np.float32) for f, c in self.features.items() if c["type"] == "continuous"}
        labels = np.zeros(B, np.float32)
        col = {f: i + (0 if is_pred else 1) for i, f in enumerate(self.fields)}
        for b, ln in enumerate(lines):
            if isinstance(ln, str):
                ln = ln.encode()
            parts = ln.rstrip(b"\r\n").split(b"\t")
            assert len(parts) == len(self.fields) + (0 if is_pred else 1), "field count"
            na = lambda v: v == b"-" or v == b""
            if not is_pred:
                labels[b] = 1.0 if (not na(parts[0]) and int(parts[0]) == 1) else 0.0
            for f in strs:
                v = parts[col[f]]
                if na(v):
                    strs[f].append([])
                elif multivalue:
                    strs[f].append([t for t in v.split(b",") if t])     # tf.string_split skips empty pieces
                else:
                    strs[f].append([v])
            for f in ints:
                v = parts[col[f]]
                ints[f][b] = 0 if na(v) else int(v)
            for f in flts:
                v = parts[col[f]]
                flts[f][b] = np.float32(0.0) if na(v) else np.float32(float(v))
        weights = None
        if self.use_weight and not is_pred:
            weights = np.where(labels > 0, np.float32(self.pos_w), np.float32(self.neg_w)).astype(np.float32)
        return {"B": B, "str": strs, "int": ints, "flt": flts, "labels": None if is_pred else labels, "weights": weights}

    # ---- feature tensors -> per-column sparse ids (tf.feature_column transforms) ---------------------
    @staticmethod
    def _csr(lists):
        offs = np.zeros(len(lists) + 1, np.int32)
        offs[1:] = np.cumsum([len(x) for x in lists])
        flat = [y for x in lists for y in x]
        return flat, offs

    @staticmethod
    def _norm(x, t, norm):
        x = x.astype(np.float32)
        if t == "min_max":
            return (x - np.float32(norm[0])) / (np.float32(norm[1]) - np.float32(norm[0]))
        if t == "standard":
            return (x - np.float32(norm[0])) / np.float32(norm[1])
        if t == "log":
            return np.log(x)
        return x

    def transform(self, parsed, cross_padding="tf_dense"):
        B = parsed["B"]
        ids, dense = {}, {}
        for key, r in self.recipes.items():
            kind = r[0]
            if kind == "hash":
                flat, offs = self._csr(parsed["str"][r[1]])          # '' never occurs: empty pieces were skipped
                ids[key] = (O.hash_bucket(flat, r[2]) if flat else np.zeros(0, np.int64), offs)
            elif kind == "vocab":
                index = {v.encode(): i for i, v in enumerate(r[2])}
                rows = [[index[t] for t in row if t in index] for row in parsed["str"][r[1]]]   # OOV = -1 -> pruned
                flat, offs = self._csr(rows)
                ids[key] = (np.asarray(flat, np.int64), offs)
            elif kind == "identity":
                v = parsed["int"][r[1]]
                rows = [[] if x == -1 else [int(x) if 0 <= x < r[2] else 0] for x in v]   # default_value=0
                flat, offs = self._csr(rows)
                ids[key] = (np.asarray(flat, np.int64), offs)
            elif kind == "numeric":
                dense[key] = self._norm(parsed["flt"][r[1]], r[2], r[3])
            elif kind == "bucket_norm":                                 # quirk C.5: normalised value, raw boundaries
                x = self._norm(parsed["flt"][r[1]], r[2], r[3])
                ids[key] = (O.bucketize(x, r[4]), np.arange(B + 1, dtype=np.int32))
            elif kind == "cross":
                cols = []
                for k in r[1]:
                    if k[0] == "string":
                        rows = parsed["str"][k[1]]
                        if cross_padding == "tf_dense":                 # quirk C.16: the padded [B, Lmax] tensor is crossed
                            lmax = max([len(x) for x in rows] + [0])
                            rows = [list(x) + [b""] * (lmax - len(x)) for x in rows]
                        flat, offs = self._csr(rows)
                        data, o = O.pack_tokens(flat)
                        cols.append((O.fingerprint64_batch(data, o) if flat else np.zeros(0, np.uint64), offs))
                    elif k[0] == "identity":
                        v = parsed["int"][k[1]]
                        rows = [[] if x == -1 else [int(x) if 0 <= x < k[2] else 0] for x in v]
                        flat, offs = self._csr(rows)
                        cols.append((np.asarray(flat, np.int64).astype(np.uint64), offs))
                    else:                                               # un-normalised numeric column inside crosses
                        cols.append((O.bucketize(parsed["flt"][k[1]], k[2]).astype(np.uint64), np.arange(B + 1, dtype=np.int32)))
                ids[key] = O.cross_hash(cols, r[2])
        out = {"ids": ids, "dense": dense, "batch_size": B, "weights": parsed["weights"]}
        if parsed["labels"] is not None:
            out["labels"] = parsed["labels"]
        return out

    def towers(self):
        hidden, mode = self.model["dnn_hidden_units"], self.model["dnn_connected_mode"]
        # python/lib/dnn.py:253-258: a 1-D hidden list is one DNN; one mode (a name, or a connection list whose first item
        # has three characters like '0-1') serves every DNN
        if not (hidden and isinstance(hidden[0], (list, tuple))):
            hidden = [hidden]
        if isinstance(mode, str) or (isinstance(mode[0], str) and len(mode[0]) == 3):
            mode = [mode] * len(hidden)
        return [(list(h), m) for h, m in zip(hidden, mode)]

    def optimizers(self):
        """(dnn_opt, lin_opt) tuples for OracleWideDeep from model.yaml (python/lib/utils/model_util.py:62-105;
        TF defaults: Adagrad initial_accumulator_value 0.1, Ftrl initial_accumulator_value 0.1, lr_power -0.5)."""
        import re

        def kw(text, key, default):
            m = re.search(key + r"\s*=\s*([-+0-9.eE]+)", text)
            return float(m.group(1)) if m else default

        def one(text, default_lr):
            lr = kw(text, "learning_rate", default_lr)
            if "Adagrad" in text:
                return ("Adagrad", lr, kw(text, "initial_accumulator_value", 0.1))
            if "Ftrl" in text:
                return ("Ftrl", lr, kw(text, "l1_regularization_strength", 0.0), kw(text, "l2_regularization_strength", 0.0),
                        kw(text, "initial_accumulator_value", 0.1))
            if "RMSProp" in text:
                return ("RMSProp", lr, kw(text, "decay", 0.9), kw(text, "momentum", 0.0), kw(text, "epsilon", 1e-10))
            if "Adam" in text:
                return ("Adam", lr, kw(text, "beta1", 0.9), kw(text, "beta2", 0.999), kw(text, "epsilon", 1e-8))
            if "SGD" in text or "GradientDescent" in text:
                return ("SGD", lr)
            raise ValueError("unsupported optimizer `%s`" % text)

        d, l = str(self.model["dnn_optimizer"]), str(self.model["linear_optimizer"])
        return (one(d, float(self.model.get("dnn_initial_learning_rate") or 0.05)),
                one(l, float(self.model.get("linear_initial_learning_rate") or 0.05)))
