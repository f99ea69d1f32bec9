"""Feature -> column planning (host logic of the hot path).

Mirrors what the reference's ``_build_model_columns`` (python/lib/build_estimator.py:49-169) does
at graph-build time, but produces a flat *plan* for the gfx950 kernels instead of
tf.feature_column objects:

  * every categorical column (hash / vocab / identity / bucketized / crossed) is a *slot* of the
    example-major bag CSR;
  * slots get a contiguous row range in ONE fused row space (wide table, sort keys) and, when they
    carry an embedding, an element offset in ONE flat embedding buffer;
  * the deep input matrix x is laid out embedding columns first (16-byte aligned for float4
    stores), then indicator columns, then numeric columns; ``tf_input_perm`` maps the
    name-sorted TF concat order to these internal columns so that weights can be
    exchanged with the reference's checkpoint naming.
"""
import math
import re
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

BN_EPS = 1e-3  # tf.layers.batch_normalization default epsilon


def embedding_dim(num_buckets):
    """Reference rule (python/lib/build_estimator.py:57-59): 2**ceil(ln(d**0.25)), natural log."""
    return int(np.power(2, np.ceil(np.log(num_buckets ** 0.25))))


@dataclass
class CrossKey:
    feature: str
    kind: str  # 'string' | 'identity' | 'bucket'
    num_buckets: int = 0
    boundaries: Optional[List[float]] = None


@dataclass
class CatSlot:
    name: str                 # TF categorical column name (= linear_model variable scope)
    kind: str                 # 'hash' | 'vocab' | 'identity' | 'bucket' | 'cross'
    num_buckets: int
    feature: Optional[str] = None
    deep: Optional[str] = None   # 'embedding' | 'indicator' | None
    dim: int = 0
    wide: bool = True
    vocab: Optional[List[str]] = None
    boundaries: Optional[List[float]] = None
    normalizer: Optional[tuple] = None      # ('min_max'|'standard'|'log', p0, p1) applied BEFORE bucketize (quirk C.5)
    cross_keys: Optional[List[CrossKey]] = None
    hash_key: int = 0xDECAFCAFFE
    # indicator columns: width of the multi-hot vector in the deep input when it is not num_buckets -- on a row-sharded rank
    # (dist.local_spec) the slot holds ceil(V / world) wide rows while the deep input still has V columns
    deep_width: Optional[int] = None

    @property
    def ind_width(self):
        return int(self.deep_width or self.num_buckets)

    @property
    def deep_name(self):
        if self.deep == "embedding":
            return self.name + "_embedding"
        if self.deep == "indicator":
            return self.name + "_indicator"
        return None


@dataclass
class DenseCol:
    name: str
    feature: str
    kind: int = 0       # 0 identity, 1 min_max, 2 standard, 3 log
    p0: float = 0.0
    p1: float = 1.0


@dataclass
class TowerSpec:
    hidden_units: List[int]
    mode: object = "simple"     # a name, or a connection list: tuple of (i, j) layer pairs (python/lib/dnn.py:195-224)


def parse_connections(mode, n_hidden):
    """`['0-1', '0-3', '1-2']` (python/lib/dnn.py:65-66, 195-205) -> sorted tuple of (i, j): layer j also reads what
    layer i read ... index 0 = the input layer, L = len(hidden_units); "smaller index first".  Accepts a list / tuple of
    'i-j' strings or (i, j) pairs, or ONE string 'i-j,i-j' (the conf reader only lets a string through, read_conf.py:183)."""
    if isinstance(mode, str):
        mode = [p for p in mode.replace(" ", "").split(",") if p]
    pairs = set()
    for e in mode:
        try:
            i, j = (int(v) for v in (e.split("-") if isinstance(e, str) else e))
        except (TypeError, ValueError):
            raise ValueError("Invalid connected_mode entry `%s`: expected 'i-j'" % (e,))
        if not 0 <= i < j <= n_hidden:
            raise ValueError("Invalid connection `%s-%s`: need 0 <= i < j <= len(hidden_units) = %d (index 0 is the input "
                             "layer, smaller index first)" % (i, j, n_hidden))
        pairs.add((i, j))
    return tuple(sorted(pairs))


def is_connection_list(mode):
    """A connection list rather than a mode name: a sequence, or a string made of 'i-j' items."""
    if isinstance(mode, str):
        return bool(re.fullmatch(r"\s*\d+\s*-\s*\d+\s*(,\s*\d+\s*-\s*\d+\s*)*,?\s*", mode))
    return isinstance(mode, (list, tuple))


@dataclass
class ModelSpec:
    model_type: str = "wide_deep"          # 'wide' | 'deep' | 'wide_deep'
    slots: List[CatSlot] = field(default_factory=list)
    dense_cols: List[DenseCol] = field(default_factory=list)
    towers: List[TowerSpec] = field(default_factory=list)
    activation: str = "relu"
    batch_norm: bool = True
    dropout: Optional[float] = None
    # optimizer tuples (build_estimator.opt_tuple): ("SGD", lr) ("Adagrad", lr, init_accum) ("Ftrl", lr, l1, l2, init_accum)
    # ("RMSProp", lr, decay, momentum, epsilon) ("Adam", lr, beta1, beta2, epsilon)
    dnn_opt: tuple = ("Adagrad", 0.05, 0.1)
    lin_opt: tuple = ("Ftrl", 0.1, 0.5, 1.0, 0.1)
    use_weight_column: bool = False
    pos_weight: float = 1.0
    neg_weight: float = 1.0
    # Learning-rate decay, OFF (None) by default -- the reference's exponential_decay runs over a fresh tf.Variable(0) that
    # nothing increments, so its learning rates never move (python/lib/joint.py:145-154).  Opt-in
    # (train.yaml `lr_decay: true`, build_estimator.build_model_spec): {"dnn": (decay_rate, decay_steps), "linear": (...)} for
    # the scopes whose optimizer is given by NAME (those take the model_fn's learning rate; a constructor string keeps its own,
    # python/lib/utils/model_util.py:84-105):  lr_t = lr_0 * decay_rate ** (global_step / decay_steps)  over TF's global step --
    # which advances 3 per train step in wide_deep mode (2 otherwise; quirk C.4): with decay_steps = num_examples / batch_size
    # (python/lib/joint.py:78; a float: joint.py:25 imports division from __future__) a rate has decayed by `decay_rate` after a THIRD
    # of an epoch, not after one as the reference's comment has it.
    lr_decay: Optional[dict] = None

    def decayed_lr(self, scope, global_step, lr0=None):
        """learning rate of `scope` ("dnn" | "linear") at TF global step `global_step` (tf.train.exponential_decay, staircase=False);
        lr0: the scope's INITIAL rate (WideDeepEngine.lr0 -- an engine's own spec carries the current rate, not the initial one)"""
        opt = self.dnn_opt if scope == "dnn" else self.lin_opt
        lr0 = float(opt[1]) if lr0 is None else float(lr0)
        sch = (self.lr_decay or {}).get(scope)
        if not sch:
            return lr0
        rate, steps = sch
        return lr0 * float(rate) ** (float(global_step) / float(steps))

    @property
    def has_deep(self):
        return self.model_type in ("deep", "wide_deep")

    @property
    def has_wide(self):
        return self.model_type in ("wide", "wide_deep")


def _round_up(x, m):
    return (x + m - 1) // m * m


class TowerLayout:
    """Column layout of one tower's activation buffer + per-layer GEMM windows.

    Segment 0 is the deep input x, segment l+1 the output of hidden layer l.  Each layer reads ONE
    contiguous column window of the buffer (free concat):
      simple      seg_l only
      dense       [seg0 | seg1 | ... | seg_l]          (python/lib/dnn.py:155-173)
      resnet      [seg_l | seg_{l-1} | ... | seg0]     (python/lib/dnn.py:175-193, concat not add: quirk C.8)
      last_dense  hidden layers as simple, logits over [seg0 | ... | seg_L]  (python/lib/dnn.py:135-153)
      connection list ((i, j) pairs, mode "list"): net_j = [net_i for every i -> j, ascending | h_{j-1}]  (python/lib/dnn.py:
                  195-224) where net_i is itself such a concat; a window holds COPIES of the segments it repeats (filled
                  when the source layer is done, their gradients added back to the source) and ends in its own segment.
    """

    def __init__(self, deep_dim, hidden, mode):
        self.hidden = list(hidden)
        widths = [deep_dim] + self.hidden
        L = len(self.hidden)
        self.connections = None
        if is_connection_list(mode):
            self.connections = parse_connections(mode, L)
            mode = "list" if self.connections else "simple"    # no edge: every layer reads its predecessor only
        if mode not in ("simple", "dense", "resnet", "last_dense", "first_dense", "list"):
            raise ValueError("Invalid connected_mode: `%s` (simple, first_dense, last_dense, dense, resnet, or a connection "
                             "list like ['0-1', '0-3', '1-2'])" % (mode,))
        self.mode = mode
        self.copies = {}            # extra segment (index > L) -> the segment whose values it repeats
        self.x_copies = []          # ... those of segment 0 (first_dense, connection lists): filled behind the input layer
        xseg_of = {}                # first_dense: layer l >= 1 -> the x segment inside its window
        win = None                  # connection list: segments of each layer's window, in concat order
        if mode == "list":
            if any(w % 4 for w in widths):
                raise ValueError("connection lists: hidden widths and the deep input width must be multiples of 4 "
                                 "(16-byte aligned windows); got %s / %d" % (self.hidden, deep_dim))
            into = {}
            for i, j in self.connections:
                into.setdefault(j, []).append(i)
            flat = [[0]]            # net_l as a list of source segments (with multiplicity)
            for l in range(1, L + 1):
                flat.append([s for i in sorted(into.get(l, [])) for s in flat[i]] + [l])
            starts, win, c = [0] * (L + 1), [[0]], deep_dim
            for l in range(1, L + 1):
                w = []
                for src in flat[l][:-1]:
                    cs = len(widths)
                    widths.append(widths[src])
                    starts.append(c)
                    c += widths[src]
                    self.copies[cs] = src
                    w.append(cs)
                starts[l] = c
                c += widths[l]
                win.append(w + [l])
            self.x_copies = [cs for cs, src in self.copies.items() if src == 0]
            total = c
        elif mode in ("dense", "last_dense"):
            starts, c = [], 0
            for w in widths:
                starts.append(c)
                c += w
            total = c
        elif mode == "resnet":
            # newest segment leftmost: [seg_L | ... | seg1 | seg0]; every layer reads a suffix window
            starts = [0] * (L + 1)
            c = 0
            for l in range(L, -1, -1):
                starts[l] = c
                c += widths[l]
            total = c
        elif mode == "first_dense":
            if any(w % 4 for w in self.hidden) or deep_dim % 4:
                raise ValueError("first_dense: hidden widths and the deep input width must be multiples of 4 "
                                 "(16-byte aligned windows); got %s / %d" % (self.hidden, deep_dim))
            # layer l >= 1 (and the logits layer) reads [h_{l-1} | x] (python/lib/dnn.py:117-133).  One copy of x has two
            # neighbours, so consumers are paired: [h_0 | x | h_1] [h_2 | x' | h_3] ... -- consumer 2g+1 reads
            # [h_2g | x_g], consumer 2g+2 reads [x_g | h_2g+1]; x' .. are copies (filled after the input layer, their
            # gradients added back into x's).  The kernel row order inside a window is internal (tf_rows_of_layer).
            starts = [0] * (L + 1)
            c = 0
            for g in range((L + 1) // 2):
                starts[2 * g + 1] = c                       # h_{2g} = segment 2g+1
                c += widths[2 * g + 1]
                if g == 0:
                    starts[0] = c
                    xs = 0
                else:
                    xs = len(widths)
                    widths.append(deep_dim)
                    starts.append(c)
                    self.x_copies.append(xs)
                    self.copies[xs] = 0
                c += deep_dim
                xseg_of[2 * g + 1] = xs
                if 2 * g + 2 <= L:                          # consumer 2g+2 exists; its h_{2g+1} is segment 2g+2
                    xseg_of[2 * g + 2] = xs
                    starts[2 * g + 2] = c
                    c += widths[2 * g + 2]
            if L == 0:
                starts[0] = 0
                c = deep_dim
            total = c
        else:  # simple: every segment 16-float aligned, no window spans two segments
            starts, c = [], 0
            for w in widths:
                starts.append(c)
                c += _round_up(w, 16)
            total = c
        self.seg_start = starts
        self.seg_width = widths
        self.ld = _round_up(total, 16)
        # per layer (hidden layers 0..L-1, then logits = index L): input window [in_start, in_start+K), list of segments
        self.in_start, self.in_K, self.in_segs = [], [], []
        for l in range(L + 1):
            is_logits = l == L
            if mode == "list":
                segs = win[l]
            elif mode == "simple" or (mode == "last_dense" and not is_logits):
                segs = [l]
            elif mode == "first_dense":
                segs = [0] if l == 0 else [l, xseg_of[l]]   # TF order: [h_{l-1} | x]
            elif mode in ("dense", "last_dense"):
                segs = list(range(0, l + 1))
            else:  # resnet
                segs = list(range(l, -1, -1))
            s0 = min(starts[j] for j in segs)
            e0 = max(starts[j] + widths[j] for j in segs)
            assert e0 - s0 == sum(widths[j] for j in segs), "a layer's input window must be contiguous"
            self.in_start.append(s0)
            self.in_K.append(e0 - s0)
            self.in_segs.append(segs)

    def canon(self, seg):
        """Segment whose values `seg` holds (copies: first_dense's of the deep input, a connection list's of anything)."""
        return self.copies.get(seg, seg)

    def copies_of(self, seg):
        """Copy segments of hidden segment `seg`, in buffer order."""
        return [cs for cs, src in self.copies.items() if src == seg]

    def window_cols(self, l):
        """For layer l: list of (segment, unit) for every column of its input window, -1 segment for pad."""
        s0, K = self.in_start[l], self.in_K[l]
        cols = [(-1, 0)] * K
        for j in self.in_segs[l]:
            for u in range(self.seg_width[j]):
                cols[self.seg_start[j] + u - s0] = (j, u)
        return cols


class FeaturePlan:
    def __init__(self, spec: ModelSpec):
        self.spec = spec
        slots = list(spec.slots)
        emb = sorted([s for s in slots if s.deep == "embedding" and spec.has_deep],
                     key=lambda s: (-(s.dim % 4 == 0), -s.dim, s.name))
        ind = sorted([s for s in slots if s.deep == "indicator" and spec.has_deep], key=lambda s: s.name)
        rest = sorted([s for s in slots if s not in emb and s not in ind], key=lambda s: s.name)
        if not spec.has_wide:
            rest = []  # wide-only columns are not needed by a deep-only model
        self.slots = emb + ind + rest
        self.S = len(self.slots)
        self.slot_index = {s.name: i for i, s in enumerate(self.slots)}
        self.n_emb, self.n_ind = len(emb), len(ind)

        # fused categorical row space + flat embedding buffer
        self.row_base, self.emb_off = [], []
        r, e = 0, 0
        for s in self.slots:
            self.row_base.append(r)
            r += int(s.num_buckets)
            if s.deep == "embedding" and spec.has_deep:
                self.emb_off.append(e)
                e += int(s.num_buckets) * int(s.dim)
                e = _round_up(e, 4)
            else:
                self.emb_off.append(-1)
        self.total_rows = r
        self.emb_elems = e
        if self.total_rows >= (1 << 32):
            raise ValueError("fused categorical row space exceeds 2^32 rows")
        self.key_bits = max(1, int(math.ceil(math.log2(max(self.total_rows, 2)))))

        # deep input columns
        self.out_col = [-1] * self.S
        c = 0
        if spec.has_deep:
            for i, s in enumerate(self.slots):
                if s.deep == "embedding":
                    if s.dim % 4 == 0:
                        c = _round_up(c, 4)
                    self.out_col[i] = c
                    c += s.dim
            for i, s in enumerate(self.slots):
                if s.deep == "indicator":
                    self.out_col[i] = c
                    c += s.ind_width
        self.dense_cols = list(spec.dense_cols) if spec.has_deep else []
        self.dense_out_col = []
        for d in self.dense_cols:
            self.dense_out_col.append(c)
            c += 1
        # internal width (may contain alignment holes, see tf_input_perm).  Wide inputs are padded to a multiple of the
        # GEMM reduction slab (64): the pad columns stay zero, their weight rows stay zero (zero gradient), and the
        # first layer then runs on full slabs only (a ragged last slab costs ~3 us per GEMM at C2, three per step)
        self.deep_dim = _round_up(c, 64) if c > 128 else _round_up(c, 4)   # pad columns stay zero (no TF rows map to them)
        self.emb_groups = {}
        for i, s in enumerate(self.slots):
            if s.deep == "embedding" and spec.has_deep:
                self.emb_groups.setdefault(int(s.dim), []).append(i)
        self.ind_slots = [i for i, s in enumerate(self.slots) if s.deep == "indicator" and spec.has_deep]

        # TF concat order (columns sorted by name) -> internal column index
        tf_cols = []
        if spec.has_deep:
            for i, s in enumerate(self.slots):
                if s.deep == "embedding":
                    tf_cols.append((s.deep_name, self.out_col[i], s.dim))
                elif s.deep == "indicator":
                    tf_cols.append((s.deep_name, self.out_col[i], s.ind_width))
            for j, d in enumerate(self.dense_cols):
                tf_cols.append((d.name, self.dense_out_col[j], 1))
        tf_cols.sort(key=lambda t: t[0])
        self.tf_deep_cols = tf_cols
        perm = []
        for _, c0, w in tf_cols:
            perm.extend(range(c0, c0 + w))
        self.tf_input_perm = np.asarray(perm, dtype=np.int64)   # len = TF deep input dim
        self.tf_deep_dim = len(perm)

        # towers + flat dense parameter layout
        # crelu (tf.nn.crelu = concat(relu(z), relu(-z)), python/lib/utils/model_util.py:52) is built as a relu layer of twice
        # the width whose kernel is tied to [W | -W] and whose bias to [b | -b] (relu(x(-W) - b) = relu(-z) exactly: IEEE
        # negation commutes with every rounding): the tower kernels see an ordinary 2N-wide relu layer, BN and the next
        # layer's kernel rows are 2N wide as in TF, the TF variables `kernel [K, N]` / `bias [N]` are the left halves.
        self.crelu = spec.has_deep and spec.activation == "crelu"
        wmul = 2 if self.crelu else 1
        self.towers = ([TowerLayout(self.deep_dim, [wmul * h for h in t.hidden_units], t.mode) for t in spec.towers]
                       if spec.has_deep else [])
        self.param_segments = []   # (name, offset, shape)
        self.layer_meta = []       # per tower: list of dicts per layer (hidden..., logits)
        off = 0
        for ti, tl in enumerate(self.towers):
            metas = []
            L = len(tl.hidden)
            for l in range(L + 1):
                K = tl.in_K[l]
                N = tl.hidden[l] if l < L else 1
                m = {"K": K, "N": N, "w_off": off}
                m["N_tf"] = N // 2 if (self.crelu and l < L) else N      # columns of the TF variable (crelu: left half)
                off += K * N
                m["b_off"] = off
                off += N
                if l < L and spec.batch_norm:
                    m["gamma_off"] = off
                    off += N
                    m["beta_off"] = off
                    off += N
                off = _round_up(off, 4)
                metas.append(m)
            # gamma / beta index per input column of each layer
            for l in range(L + 1):
                gi, bi = [], []
                for (seg, u) in tl.window_cols(l):
                    seg = tl.canon(seg)
                    if seg >= 1 and spec.batch_norm:
                        gi.append(metas[seg - 1]["gamma_off"] + u)
                        bi.append(metas[seg - 1]["beta_off"] + u)
                    else:
                        gi.append(-1)
                        bi.append(-1)
                metas[l]["gamma_idx"] = np.asarray(gi, dtype=np.int32)
                metas[l]["beta_idx"] = np.asarray(bi, dtype=np.int32)
            self.layer_meta.append(metas)
        self.dense_param_elems = max(off, 4)

    # ---- helpers -----------------------------------------------------------------------------
    def tf_rows_of_layer(self, ti, l):
        """Row indices (into the internal [K, N] kernel of tower ti layer l) in TF row order, i.e. the
        order of the reference's concat, with seg0 expanded through tf_input_perm."""
        tl = self.towers[ti]
        rows = []
        s0 = tl.in_start[l]
        if tl.mode in ("simple",) or (tl.mode == "last_dense" and l < len(tl.hidden)):
            order = [l]
        elif tl.mode in ("first_dense", "list"):
            order = list(tl.in_segs[l])                    # [h_{l-1}, x] / [net_i ... | h_{l-1}]: the reference's concat order
        elif tl.mode in ("dense", "last_dense"):
            order = list(range(0, l + 1))
        else:
            order = list(range(l, -1, -1))
        for j in order:
            base = tl.seg_start[j] - s0
            if tl.canon(j) == 0:
                rows.extend((base + self.tf_input_perm).tolist())
            else:
                rows.extend(range(base, base + tl.seg_width[j]))
        return np.asarray(rows, dtype=np.int64)


# ---------------------------------------------------------------------------------------------
# Synthetic Criteo-shaped specs (BASELINE.json configs 2-5)
# ---------------------------------------------------------------------------------------------
def small_table_slots(slots, has_deep, has_wide, max_floats, mode="cross"):
    """Indices (into `slots`, a FeaturePlan's order) of the categorical columns whose whole table fits in LDS and that go through
    csrc/small_tables.hip: crossed columns (mode 'cross'; 'all': any column) that are embedded with a width <= 16 or wide, not
    indicator columns, admitted against the JOINT maxima the kernels size their LDS by -- rows of the longest table x (widest
    embedding + 2) <= max_floats (a 2000-row wide-only cross and a 1000-row cross of width 4 fit one by one and not together:
    the later one stays on the general path).  mode '0': none."""
    out, jr, jd = [], 0, 0
    if mode == "0":
        return out
    for i, s in enumerate(slots):
        emb = s.deep == "embedding" and has_deep
        d = int(s.dim) if emb else 0
        wide = bool(has_wide and s.wide)
        if ((s.kind == "cross" or mode == "all") and not (s.deep == "indicator" and has_deep) and (d > 0 or wide) and d <= 16
                and max(jr, int(s.num_buckets)) * (max(jd, d) + 2) <= max_floats):
            out.append(i)
            jr, jd = max(jr, int(s.num_buckets)), max(jd, d)
    return out


def criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=16, hidden=(256, 128, 64), mode="simple",
                model_type="wide_deep", batch_norm=True, crosses=(), cross_buckets=200, use_weight_column=False,
                pos_weight=0.99, neg_weight=0.01):
    slots = []
    for i in range(n_sparse):
        slots.append(CatSlot(name="C%02d" % i, kind="hash", num_buckets=int(buckets), feature="C%02d" % i,
                             deep="embedding", dim=int(dim), wide=True))
    for keys in crosses:
        feats = ["C%02d" % k for k in keys]
        slots.append(CatSlot(name="_X_".join(sorted(feats)), kind="cross", num_buckets=int(cross_buckets),
                             deep="embedding", dim=embedding_dim(cross_buckets), wide=True,
                             cross_keys=[CrossKey(f, "string") for f in feats]))
    dense = [DenseCol(name="I%02d" % i, feature="I%02d" % i) for i in range(n_dense)]
    return ModelSpec(model_type=model_type, slots=slots, dense_cols=dense,
                     towers=[TowerSpec(list(hidden), mode)], batch_norm=batch_norm,
                     use_weight_column=use_weight_column, pos_weight=pos_weight, neg_weight=neg_weight)


# ---------------------------------------------------------------------------------------------
# optimizer state layout (tf.train slot names; include/wd_hip.h wd_opt_t: slot a / slot b of a variable)
# ---------------------------------------------------------------------------------------------
OPT_SLOT_NAMES = {"SGD": (None, None), "Adagrad": (None, "/Adagrad"), "Ftrl": ("/Ftrl_1", "/Ftrl"),
                  "RMSProp": ("/RMSProp", "/RMSProp_1"), "Adam": ("/Adam", "/Adam_1")}


def ftrl_lr_power(opt):
    """learning_rate_power of an Ftrl tuple ("Ftrl", lr, l1, l2, init[, lr_power[, l2_shrinkage]]); TF default -0.5."""
    return float(opt[5]) if len(opt) > 5 else -0.5


def ftrl_l2_shrinkage(opt):
    """l2_shrinkage_regularization_strength of an Ftrl tuple (TF default 0: the 7th element exists only when it is not)."""
    return float(opt[6]) if (opt[0] == "Ftrl" and len(opt) > 6) else 0.0


def rmsprop_centered(opt):
    """("RMSProp", lr, decay, momentum, epsilon[, centered])"""
    return opt[0] == "RMSProp" and len(opt) > 5 and bool(opt[5])


def opt_slot_names(opt):
    """Checkpoint suffixes (slot a, slot b, slot c) of an optimizer tuple -- TF numbers the slot variables of one optimizer in
    creation order: RMSPropOptimizer._create_slots makes rms, [mg when centered,] momentum, so centered=True shifts
    momentum from /RMSProp_1 to /RMSProp_2 and the mean gradient (slot c) takes /RMSProp_1."""
    if rmsprop_centered(opt):
        return "/RMSProp", "/RMSProp_2", "/RMSProp_1"
    return OPT_SLOT_NAMES[opt[0]] + (None,)


# slots that are ODD functions of the gradient history (they flip sign with the parameter: a crelu layer's mirrored half
# holds their negation), per optimizer (slot a, slot b); slot c (mean gradient) is odd
OPT_SLOT_ODD = {"SGD": (False, False), "Adagrad": (False, False), "Ftrl": (True, False), "RMSProp": (False, True),
                "Adam": (True, False)}


def opt_slot_init(opt):
    """Initial values (slot a, slot b) of an optimizer tuple; None: the optimizer has no such slot.  (Slot c -- centered
    RMSProp's mean gradient -- starts at 0.)"""
    kind = opt[0]
    if kind == "SGD":
        return None, None
    if kind == "Adagrad":
        return None, float(opt[2])
    if kind == "Ftrl":
        return 0.0, float(opt[4])
    if kind == "RMSProp":
        return 1.0, 0.0           # rms slot starts at ones, momentum at zeros (RMSPropOptimizer._create_slots)
    if kind == "Adam":
        return 0.0, 0.0
    raise ValueError("unsupported optimizer %r" % (opt,))


def opt_params(opt):
    """(p0, p1, p2, p3) of wd_opt_t."""
    kind = opt[0]
    if kind == "Ftrl":
        return float(opt[2]), float(opt[3]), ftrl_lr_power(opt), ftrl_l2_shrinkage(opt)
    if kind in ("RMSProp", "Adam"):
        return float(opt[2]), float(opt[3]), float(opt[4]), 0.0
    return 0.0, 0.0, 0.0, 0.0


def adam_pow_names(dnn_opt, lin_opt, has_deep, has_wide):
    """Checkpoint names of Adam's non-slot beta powers: the optimizer built first by python/lib/joint.py:224-262 (dnn,
    then linear) owns beta1_power / beta2_power, a second Adam instance gets the _1 suffix."""
    out, n = {}, 0
    for scope, opt, on in (("dnn", dnn_opt, has_deep), ("linear", lin_opt, has_wide)):
        if on and opt[0] == "Adam":
            suf = "" if n == 0 else "_%d" % n
            out[scope] = ("beta1_power" + suf, "beta2_power" + suf)
            n += 1
    return out


def bucket_geometry(vocab_sizes, occ_per_slot, nb_max, target=64.0):
    """Row-range buckets of the fused sparse backward (wd_sparse_bucketize): slot s is cut into ceil(V_s / 2^shift_s)
    buckets of 2^shift_s consecutive rows, sized for ~`target` occurrences per bucket when ids are uniform.  A slot
    whose vocabulary is smaller than the bucket count it would get keeps one bucket PER ROW (shift 0), which the
    update kernel handles without sorting -- the tiny vocabularies of the reference's conf/feature.yaml (2-55 rows,
    thousands of occurrences per row at batch 8192) land there.  Returns (shifts, bases, total buckets <= nb_max)."""
    vocab_sizes = [max(int(v), 1) for v in vocab_sizes]
    t = float(target)
    while True:
        shifts, bases, total = [], [], 0
        for v in vocab_sizes:
            want = max(1.0, float(occ_per_slot) / t)
            sh = max(0, math.ceil(math.log2(v / want))) if v > want else 0
            shifts.append(sh)
            bases.append(total)
            total += (v + (1 << sh) - 1) >> sh
        if total <= nb_max:
            return shifts, bases, max(total, 1)
        t *= 1.25
