"""Multi-GPU path: synchronous data parallel ranks + row-sharded tables exchanged with all-to-all.

The reference's only parallelism is TF's asynchronous parameter server with PS-partitioned
variables (python/train.py:202-225, python/lib/joint.py:140-143, python/lib/build_estimator.py:172-198).
The MI355X equivalent: one process per GPU, examples split across ranks, every
embedding / wide table sharded by ROW across the ranks of the node (owner = id % world,
local row = id // world: balances every slot regardless of its size or skew), and

  forward   all_to_all(ids) -> owner gathers rows -> all_to_all(rows) -> requester pools per bag
  backward  all_to_all(per-occurrence row gradients) -> owner: sort + segment-sum + fused Adagrad / FTRL
  dense     one flat-buffer all_reduce(SUM) of the tower gradients (the loss is a batch SUM, App. A.7)

so that N ranks are numerically one GPU running the global batch (up to fp32 summation order).
RCCL over xGMI is reached through torch.distributed (backend "nccl"); with the "gloo" backend
(CPU tests, or several ranks sharing one GPU) tensors are staged through host memory.

The exchange orchestration below is device-agnostic torch plumbing; the compute callbacks
(owner gather / requester pooling / owner update) are the HIP kernels in ShardedWideDeepEngine and
oracle functions in the CPU tests.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import capi, hipgraph
from .capi import call, ptr
from .engine import DeviceBatch, WideDeepEngine, _stream
from .plan import CatSlot, FeaturePlan, ModelSpec, bucket_geometry, ftrl_l2_shrinkage, ftrl_lr_power, small_table_slots


# ---------------------------------------------------------------------------------------------
# exchange plumbing (any device, any backend)
# ---------------------------------------------------------------------------------------------
def _a2a(out, inp, out_splits, in_splits, group=None, async_op=False):
    """all_to_all_single that also works for CUDA tensors on the gloo backend (host staging).
    async_op: returns the work handle (None on the staged path, which completes before it returns)."""
    backend = dist.get_backend(group)
    if inp.is_cuda and backend == "gloo":
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_to_all_single(o, i, out_splits, in_splits, group=group)
        out.copy_(o)
        return None
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)


def _all_reduce_sum(t, group=None, async_op=False):
    backend = dist.get_backend(group)
    if t.is_cuda and backend == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        t.copy_(c)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class _Segments:
    """A train step as [hipGraph, collective, hipGraph, ...]: the kernels between two collectives are captured once and
    replayed with one launch each; the collectives themselves stay ordinary torch.distributed calls on the same stream
    (nothing of RCCL is captured), so the replayed step is the eager step minus ~30 kernel-launch round trips."""

    def __init__(self):
        self.ops = []
        self.pool = torch.cuda.graph_pool_handle()
        self.g = None

    def begin(self):
        self.g = hipgraph.new_graph()
        # thread_local: the process group's watchdog thread polls events while we capture
        self.g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self.mark = capi.NCALLS

    def boundary(self, fn):
        if capi.NCALLS == self.mark:      # two collectives back to back: no launch in between, keep the open segment
            self.ops.append(fn)
            return
        self.g.capture_end()
        self.ops.append(self.g.replay)
        self.ops.append(fn)
        self.begin()

    def end(self):
        self.g.capture_end()
        if capi.NCALLS != self.mark:
            self.ops.append(self.g.replay)
        self.g = None


class ExchangePlan:
    """Who-sends-what for one batch of occurrences.

    ids [n] (slot-local ids, any int dtype), slot_of [n] (slot index per occurrence),
    local_row_base [S] (first local fused row of each slot on every rank).
    owner = id % W, local fused row = local_row_base[slot] + id // W.
    `order` lists the occurrences grouped by owner (stable), `inv` is its inverse permutation."""

    def __init__(self, ids, slot_of, local_row_base, world, group=None):
        self.world, self.group = world, group
        ids = ids.long()
        owner = ids % world
        lrow = local_row_base[slot_of.long()] + ids // world
        self.n = int(ids.numel())
        _, self.order = torch.sort(owner, stable=True)
        self.inv = torch.empty_like(self.order)
        self.inv[self.order] = torch.arange(self.n, device=ids.device)
        self.send_rows = lrow[self.order].to(torch.int32)
        send_counts = torch.bincount(owner, minlength=world)[:world]
        recv_counts = torch.empty_like(send_counts)
        _a2a(recv_counts, send_counts, None, None, group)          # one int per peer
        self.send_counts = [int(v) for v in send_counts.cpu()]
        self.recv_counts = [int(v) for v in recv_counts.cpu()]
        self.n_recv = sum(self.recv_counts)
        self.req_rows = torch.empty(self.n_recv, dtype=torch.int32, device=ids.device)
        _a2a(self.req_rows, self.send_rows, self.recv_counts, self.send_counts, group)

    def to_owner(self, payload):
        """payload [n, ...] in bucketed (order) layout -> [n_recv, ...] aligned with req_rows."""
        out = torch.empty((self.n_recv,) + tuple(payload.shape[1:]), dtype=payload.dtype, device=payload.device)
        k = int(np.prod(payload.shape[1:])) if payload.dim() > 1 else 1
        _a2a(out.view(-1), payload.contiguous().view(-1), [c * k for c in self.recv_counts],
             [c * k for c in self.send_counts], self.group)
        return out

    def to_requester(self, payload):
        """payload [n_recv, ...] aligned with req_rows -> [n, ...] in bucketed (order) layout."""
        out = torch.empty((self.n,) + tuple(payload.shape[1:]), dtype=payload.dtype, device=payload.device)
        k = int(np.prod(payload.shape[1:])) if payload.dim() > 1 else 1
        _a2a(out.view(-1), payload.contiguous().view(-1), [c * k for c in self.send_counts],
             [c * k for c in self.recv_counts], self.group)
        return out


def occurrence_slots(bag_offs, B, S, nnz):
    """slot index and example index of every occurrence of an example-major bag CSR."""
    lens = (bag_offs[1:] - bag_offs[:-1]).long()
    bag = torch.repeat_interleave(torch.arange(B * S, device=bag_offs.device), lens, output_size=nnz)
    return bag % S, bag // S, lens


def shard_rows(v, world):
    return (int(v) + world - 1) // world


def replicated_slots(plan, spec, world):
    """Plan indices of the columns a row-sharded model keeps WHOLE on every rank instead of exchanging their rows: the small
    tables of csrc/small_tables.hip (crossed columns of a few hundred buckets -- python/lib/build_estimator.py:138-155 -- whose
    bags hold the product of their keys' counts: configs[3]'s two crosses are 1.2 M of a batch's 2.3 M occurrences for 2 x 200
    rows).  Forward from the local copy, gradient sums all-reduced, the same update on every rank.  Taken when the embedded ones
    come behind every exchanged embedded column in the plan's order (the owner's "embedding row = local fused row" needs the
    exchanged tables first) and the exchanged columns share one embedding width; otherwise every column is exchanged."""
    import os
    idx = small_table_slots(plan.slots, spec.has_deep, spec.has_wide, capi.SMALL_MAX_FLOATS, os.environ.get("WD_SMALL_TABLES", "cross"))
    if not idx or not spec_default_opts(spec):
        return []
    emb = [i for i, s in enumerate(plan.slots) if s.deep == "embedding" and spec.has_deep]
    xemb = [i for i in emb if i not in idx]
    if xemb and any(i < max(xemb) for i in idx if i in emb):
        return []
    if len({int(plan.slots[i].dim) for i in xemb}) > 1:
        return []
    return idx


def spec_default_opts(spec):
    """the reference's default optimizer pair (conf/model.yaml): Adagrad on the dnn scope, Ftrl (lr_power -0.5) on the linear one"""
    return ((not spec.has_deep or spec.dnn_opt[0] == "Adagrad") and
            (not spec.has_wide or (spec.lin_opt[0] == "Ftrl" and ftrl_lr_power(spec.lin_opt) == -0.5
                                   and ftrl_l2_shrinkage(spec.lin_opt) == 0.0)))


def local_spec(spec: ModelSpec, world, keep=()):
    """Same model, every categorical column holding only this rank's rows (ceil(V / world)); `keep`: names of the columns that
    stay whole (replicated_slots)."""
    slots = []
    for s in spec.slots:
        d = CatSlot(**{k: getattr(s, k) for k in s.__dataclass_fields__})
        if s.name not in keep:
            d.num_buckets = shard_rows(s.num_buckets, world)
        if s.deep == "indicator":
            d.deep_width = int(s.num_buckets)     # the multi-hot vector of the deep input keeps all V columns
        slots.append(d)
    return ModelSpec(model_type=spec.model_type, slots=slots, dense_cols=list(spec.dense_cols),
                     towers=list(spec.towers), activation=spec.activation, batch_norm=spec.batch_norm,
                     dropout=spec.dropout, dnn_opt=spec.dnn_opt, lin_opt=spec.lin_opt,
                     use_weight_column=spec.use_weight_column, pos_weight=spec.pos_weight, neg_weight=spec.neg_weight,
                     lr_decay=spec.lr_decay)


# ---------------------------------------------------------------------------------------------
# HIP engine
# ---------------------------------------------------------------------------------------------
def unique_key_space(global_plan, local_plan, world):
    """Key space of the sender-side unique (requester side): per slot (row_base, ids) with  key = row_base + id  for an id of the
    GLOBAL space [0, V_s), laid out so that  key % world = id % world = the owner  and  key // world = the owner's local row
    (local_row_base(slot) + id // world): row_base = world * local_row_base(slot).  A slot's keys end before the next slot's
    begin because every rank holds ceil(V_s / world) rows of it."""
    W = int(world)
    return [(W * int(local_plan.row_base[i]), int(s.num_buckets)) for i, s in enumerate(global_plan.slots)]


class ShardedWideDeepEngine(WideDeepEngine):
    """WideDeepEngine whose tables hold rows id % world == rank; batches are this rank's examples.

    One step = the single-GPU step with three fixed-size all-to-alls in the sparse part and one all-reduce of the
    flat dense gradient (csrc/dist_exchange.hip has the device side).  Nothing in the step reads a value back to
    the host: the per-peer segments have a static capacity `cap = slack * expected_nnz / world`; `check_overflow()`
    (one device->host read, call it every few hundred steps) raises if a peer ever received more than that.
    """

    def __init__(self, spec: ModelSpec, max_batch=8192, max_nnz=None, device="cuda", seed=0, group=None, slack=1.5,
                 expected_nnz=None, expected_unique=None, dedup=None):
        if not dist.is_initialized():
            raise RuntimeError("ShardedWideDeepEngine needs torch.distributed to be initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.global_spec = spec
        self.global_plan = FeaturePlan(spec)
        gp = self.global_plan
        # replicated columns (small tables): whole on every rank, never exchanged
        self.rep_idx = replicated_slots(gp, spec, self.world)
        rep_names = {gp.slots[i].name for i in self.rep_idx}
        dims = ({int(s.dim) for i, s in enumerate(gp.slots) if s.deep == "embedding" and i not in self.rep_idx}
                if spec.has_deep else set())
        if any(d not in (4, 8, 16, 32, 64, 128) for d in dims):
            raise NotImplementedError("sharded engine: embedding dims must be 4, 8, ..., 128 (got %s)" % sorted(dims))
        # Mixed embedding dims (the reference's default rule, build_estimator.py:57-59, gives every hashed column its own):
        # every exchanged record is [Dmax floats | wide], and the OWNER keeps its embedding rows padded to Dmax, so that
        # "embedding row = local fused row" holds and one slot descriptor covers the whole local row space.  A slot of a
        # smaller dim sends / receives zeros in the pad columns (zero gradient: they never move).
        self.mixed_dims = len(dims) > 1
        S = gp.S
        mn = int(max_nnz) if max_nnz else int(max_batch) * max(S, 1) * 8
        W = self.world
        # entries per peer segment: the all-to-alls always move FULL segments, so size them from the occurrences a
        # batch really has (expected_nnz), not from the worst case the buffers could hold
        # Sender-side unique (one id per bag, row records): a batch asks its owners for every DISTINCT row once and sends one
        # pre-summed gradient per row -- the segments are then sized from the distinct rows a batch holds (expected_unique).
        # dedup: True / False, or None = WD_SHARD_DEDUP (1 / 0 / auto: on when the first batch holds < 80 % distinct rows -- with
        # uniform ids over 1M-row tables 99.6 % are distinct and the requester-side sort would buy nothing).
        if dedup is None:
            env = os.environ.get("WD_SHARD_DEDUP", "auto")
            dedup = True if env == "1" else False if env == "0" else (
                expected_unique is not None and expected_unique < 0.8 * int(expected_nnz or mn))
        # ... and only where the unique path is GUARANTEED to be taken: a batch routed per occurrence into segments sized for its
        # distinct rows would lose occurrences until the next check_overflow().  Everything static is decided here, before the
        # segments are sized (record-shaped model -- engine.py's rule --, key space below 2^32, at most 128 slots); what is left
        # to the batch (one id per bag) makes _route raise instead of falling back.
        dset = sorted(dims)
        static_ok = (spec.has_deep and spec.has_wide and len(dset) == 1 and dset[0] in (4, 8, 16) and 0 < S <= 128
                     and not self.rep_idx and all(s.deep == "embedding" and s.wide for s in gp.slots)
                     and os.environ.get("WD_ROW_RECORDS", "1") != "0"
                     and W * sum(shard_rows(int(s.num_buckets), W) for s in gp.slots) < (1 << 32))
        self.dedup = bool(dedup) and static_ok
        self._cap_from_unique = bool(self.dedup and expected_unique)
        need = int(expected_unique) if self._cap_from_unique else int(expected_nnz or mn)
        self.cap = ((int(math.ceil(need / W * slack)) + 63) // 64) * 64
        # the all-to-alls move [W][cap] on every rank: a rank that sized its segments from a different first batch would
        # hang or corrupt the exchange -- agree on the largest
        capt = torch.tensor([self.cap], dtype=torch.int64)
        if dist.get_backend(group) != "gloo":
            capt = capt.to(device)
        dist.all_reduce(capt, op=dist.ReduceOp.MAX, group=group)
        self.cap = int(capt.item())
        self.n_req = W * self.cap
        # owner side: the received list has n_req entries (padding included) -> capacity of the update workspaces
        # the owner keeps its LOCAL rows as records [emb | w z n - | pad] when the model is record-shaped (engine.py): a request
        # is answered from one 128-byte line, the update touches two lines per row (WD_ROW_RECORDS=0: separate tables)
        self._records_ok = not self.mixed_dims and not self.rep_idx       # (the small-table kernels take separate tables)
        if self.dedup and not spec_default_opts(spec):
            self.dedup = self._cap_from_unique = False      # (the sender-side unique works on row records: default optimizers)
        self._small_forced = list(self.rep_idx)
        super().__init__(local_spec(spec, W, keep=rep_names), max_batch=max_batch, max_nnz=max(mn, self.n_req), device=device,
                         seed=seed, expected_nnz=self.n_req, table_seed=int(seed) * 1000003 + 7919 * (self.rank + 1))
        # any optimizer python/lib/utils/model_util.py:84-90 accepts on either scope: the defaults (Adagrad / Ftrl) keep their
        # specialised kernels, the others take the generic owner update (wd_sparse_apply_opt) on separate tables
        self._segs = None
        self._work_c = None
        self._work_d = None
        self._train_fwd = False       # forward() of a train step: start the owner-side bucketing early
        self._bucketized = False
        self.req_max_nnz = mn
        self.dim = max(dims) if dims else 0       # width of the embedding part of an exchanged record
        dev = self.device
        lp = self.plan
        assert [s.name for s in lp.slots] == [s.name for s in gp.slots]      # bags are indexed by the GLOBAL plan's slot order
        # EXCHANGED embedding slots come first in the fused row space: embedding row == fused local row (rows padded to self.dim)
        self.n_emb_slots = (lp.n_emb - sum(1 for i in self.rep_idx if lp.slots[i].deep == "embedding")) if spec.has_deep else 0
        self.n_emb_rows = sum(int(s.num_buckets) for s in lp.slots[: self.n_emb_slots])
        if self.rep_idx:
            self._sync_replicas()
        if self.mixed_dims:
            self._pad_tables()
        else:
            assert all(lp.emb_off[i] == lp.row_base[i] * self.dim for i in range(self.n_emb_slots))
        # indicator columns (vocabulary / identity columns in the deep input): computed by the requester from its own ids --
        # nothing to exchange; descriptors with the GLOBAL vocabulary size as the width of the multi-hot vector
        self.ind_xslots_dev = None
        has_emb, has_wide = self.n_emb_slots > 0, spec.has_wide
        self.RS = (self.dim if has_emb else 0) + ((4 if has_emb else 1) if has_wide else 0)   # floats per exchanged row
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        # two sets of routing state: in a pipelined multi-step graph (ShardedStepGraph) the requests of step t+1 are routed
        # and exchanged (A) while step t still reads its own set
        self._xsets = [dict(send_rows=torch.full((self.n_req,), -1, **i32), recv_rows=torch.full((self.n_req,), -1, **i32),
                            pos=torch.zeros(self.max_nnz, **i32),
                            route_ws=torch.zeros(int(call("wd_route_chunks")) * W, **i32),
                            peer_counts=torch.zeros(W, **i32)) for _ in range(2)]
        self._pset = 0
        self._rq = None
        if self.dedup:
            self._setup_dedup()
        self._skip_exchange = False
        self.overflow = torch.zeros(1, **i32)
        self.fwd_send = torch.zeros(self.n_req * self.RS, **f32)
        self.fwd_recv = torch.zeros(self.n_req * self.RS, **f32)
        self.bwd_send = torch.zeros(self.n_req * self.RS, **f32)
        self.bwd_recv = torch.zeros(self.n_req * self.RS, **f32)
        self.req_offs = torch.arange(self.n_req + 1, **i32)             # one "bag" per received request

        def make_slots(entries):
            arr = (capi.WdSlot * len(entries))()
            for i, e in enumerate(entries):
                for k, v in e.items():
                    setattr(arr[i], k, v)
            return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)

        # requester side: "table" = the rows that came back (fwd_recv), "id" = position in it
        xs = []
        for i, s in enumerate(lp.slots):
            is_emb = bool(s.deep == "embedding" and spec.has_deep)
            xs.append(dict(emb_off=0 if is_emb else -1, row_base=0, num_buckets=1 << 30, dim=int(s.dim) if is_emb else 0,
                           out_col=lp.out_col[i], kind=capi.SLOT_EMBEDDING if is_emb else capi.SLOT_NONE,
                           wide=1 if (spec.has_wide and s.wide) else 0))
        self.xslots_dev = make_slots(xs)
        for i in self.rep_idx:           # pooling / wide sum of the received rows skip the replicated columns
            xs[i]["flags"] = capi.SLOT_F_SMALL
        self.xslots_small_dev = make_slots(xs) if self.rep_idx else None
        if spec.has_deep and lp.ind_slots:
            self.ind_xslots_dev = make_slots([dict(emb_off=-1, row_base=0, num_buckets=s.ind_width, dim=0,
                                                   out_col=lp.out_col[i], kind=capi.SLOT_INDICATOR, wide=0)
                                              for i, s in enumerate(lp.slots)])
        # owner side: the whole local fused row space is one slot; only rows below n_emb_rows carry embeddings
        # (its row-range buckets therefore span slots: one geometry over all local rows)
        osh, _, self.n_buckets = bucket_geometry([max(lp.total_rows, 1)], self.n_req, int(call("wd_bucket_max")),
                                                 float(os.environ.get("WD_BUCKET_TARGET", "64")))
        # owner-side bucketing scratch, two sets like the routing state (the base engine's sets follow its own geometry)
        self._obsets = [dict(cnt=torch.zeros((2 * int(call("wd_bucket_chunks")) + 1) * self.n_buckets, **i32),
                             start=torch.zeros(2 * self.n_buckets + 2, **i32),   # starts [nb+1] + launch order [nb]
                             rank=torch.zeros(self.n_req, **i32),
                             long_list=torch.zeros(2 * (self.n_req // 32 + 2) + 2, **i32), big_list=torch.zeros(self.n_buckets, **i32),
                             pairs=torch.zeros(self.n_req, dtype=torch.int64, device=dev)) for _ in range(2)]
        # default optimizers on row records, every local row embedded: the owner's update is the single engine's flat one -- the
        # buckets of the received list sorted ahead (one wavefront per bucket, beside the gradient exchange), wd_row_update_ragged
        # over the sorted pairs (WD_OWNER_FLAT=0: wd_sparse_apply_rec sorts inside its update workgroups)
        self.owner_flat = (self.rec is not None and self.default_opts and self.dim in (4, 8, 16) and has_emb
                           and self.n_emb_rows == lp.total_rows and os.environ.get("WD_OWNER_FLAT", "1") != "0")
        self.oslot_dev = make_slots([dict(emb_off=0, row_base=0, num_buckets=max(self.n_emb_rows, 1), dim=self.dim,
                                          out_col=0, kind=capi.SLOT_EMBEDDING if has_emb else capi.SLOT_NONE, wide=1,
                                          bucket_shift=osh[0], bucket_base=0)])
        # the local row space as two row ranges for wd_adam_untouched: rows with an embedding (padded to self.dim) + a wide line,
        # then the rows of the wide-only columns
        self.aslots_dev = make_slots([dict(emb_off=0 if has_emb else -1, row_base=0, num_buckets=max(self.n_emb_rows, 1) if has_emb else 0,
                                           dim=self.dim, out_col=0, kind=capi.SLOT_EMBEDDING if has_emb else capi.SLOT_NONE, wide=1),
                                      dict(emb_off=-1, row_base=self.n_emb_rows if has_emb else 0,
                                           num_buckets=max(lp.total_rows - (self.n_emb_rows if has_emb else 0), 0), dim=0, out_col=0,
                                           kind=capi.SLOT_NONE, wide=1)])
        # bias gradient = sum_b dlogit[b] = bias gradient of the logits layer -> rides in the flat all-reduce
        self._logits_b_off = self.towers[0]["metas"][-1]["b_off"] if spec.has_deep else None
        # (several towers -- python/lib/dnn.py:260-274 -- share the input layer and every tower's logits bias sees the same
        # gradient sum_b dlogit[b]: tower 0's rides for the linear bias)
        self._gsum = torch.zeros(1, **f32)
        # feature hashing happens in the GLOBAL id space (id = Fingerprint64 % global buckets); only then is an id split
        # into (owner, local row).  Batches are therefore built / hashed against global_plan + these descriptors.
        garr = (capi.WdSlot * max(gp.S, 1))()
        for i, sl in enumerate(gp.slots):
            garr[i].emb_off, garr[i].row_base = -1, gp.row_base[i]
            garr[i].num_buckets, garr[i].dim, garr[i].out_col = int(sl.num_buckets), 0, -1
            garr[i].kind, garr[i].wide = capi.SLOT_NONE, 0
        self.hash_slots_dev = torch.from_numpy(np.frombuffer(bytes(garr), dtype=np.uint8).copy()).to(dev)
        self.hash_plan = gp

    def _pad_tables(self):
        """Mixed embedding dims: re-lay the local tables (and the Adagrad accumulator) as [n_emb_rows][Dmax]."""
        lp, D, n = self.plan, self.dim, self.n_emb_rows
        from .plan import opt_slot_init
        ia, ib = opt_slot_init(self.spec.dnn_opt)
        for name, fill in (("emb", 0.0), ("emb_a", ia), ("emb_acc", ib), ("emb_c", 0.0)):
            old = getattr(self, name, None)
            if old is None:
                continue
            new = torch.full((max(n * D, 4),), float(fill), dtype=torch.float32, device=self.device)
            v = new[: n * D].view(n, D)
            for i in range(self.n_emb_slots):
                s, r0 = lp.slots[i], lp.row_base[i]
                v[r0: r0 + s.num_buckets, : s.dim] = old[lp.emb_off[i]: lp.emb_off[i] + s.num_buckets * s.dim].view(
                    s.num_buckets, s.dim)
            setattr(self, name, new)
            if name == "emb_c" and "dnn" in self.opt_c:
                self.opt_c["dnn"].slot_c = new.data_ptr()      # (the descriptor held the flat buffer's address)
        self._padded = True

    def _emb_view(self, buf, i):
        if not self.mixed_dims or not getattr(self, "_padded", False):     # (the base constructor fills the flat layout first)
            return super()._emb_view(buf, i)
        s, r0 = self.plan.slots[i], self.plan.row_base[i]
        return buf[: self.n_emb_rows * self.dim].view(self.n_emb_rows, self.dim)[r0: r0 + s.num_buckets, : s.dim]

    # current routing / owner-bucketing set (see _xsets)
    send_rows = property(lambda self: self._xsets[self._pset]["send_rows"])
    recv_rows = property(lambda self: self._xsets[self._pset]["recv_rows"])
    pos = property(lambda self: self._xsets[self._pset]["pos"])
    route_ws = property(lambda self: self._xsets[self._pset]["route_ws"])
    peer_counts = property(lambda self: self._xsets[self._pset]["peer_counts"])

    def _collective(self, fn):
        """Run a collective now, or -- while a step is being captured as graph segments -- close the current segment and
        record it (full-graph capture: the collective itself is captured, `fn` simply runs)."""
        if self._segs is not None:
            self._segs.boundary(fn)
        else:
            fn()

    def check_overflow(self):
        """Collective: every rank learns the largest segment any rank was asked to fill, and all raise together (a rank that
        raised alone would leave its peers hanging in the next all-to-all)."""
        flag = self.overflow.clone()
        if dist.get_backend(self.group) == "gloo":
            flag = flag.cpu()
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        n = int(flag.item())
        if n:
            raise capi.WdError("all-to-all segment overflow: a peer received %d requests, capacity %d; "
                               "raise `slack` or `max_nnz`" % (n, self.cap))

    def _fold_at_end(self):
        # one-launch tower: the dense tail (Adagrad on the all-reduced gradient + the packed kernels of the next step) runs
        # right behind the all-reduce, so forward() never launches a fold / pack
        return bool(self.chain) and os.environ.get("WD_FOLD_AT_END", "1") == "1"

    # ---- overrides ------------------------------------------------------------------------------
    def _sparse_forward(self, bt: DeviceBatch, st):
        self._sparse_exchange(bt, st)
        self._pool(bt, st)

    def _chain_input_ok(self, bt):
        # the tower kernel pools the received rows itself: embedding rows + wide weight travel together (needs both)
        return super()._chain_input_ok(bt) and self.n_emb_slots > 0 and self.n_emb_slots == self.plan.S

    def _chain_input(self, bt, tw):
        """x tile from the rows that came back through the exchange: `pos` indexes fwd_recv (row stride RS)."""
        ci = super()._chain_input(bt, tw)
        ci.emb, ci.slots, ci.ids = ptr(self.fwd_recv), ptr(self.xslots_dev), ptr(self.pos)
        ci.row_stride = self.RS
        ci.wide, ci.wide_in_row = None, 0
        if self.spec.has_wide:
            ci.wide, ci.wide_in_row = ptr(self.fwd_recv), 1
        return ci

    def _setup_dedup(self):
        """Requester-side state of the sender-side unique: slot descriptors over the key space  W * local_row_base(slot) + id
        (owner = key % W, local row = key / W), their row-range bucket geometry, and sort scratch per routing set."""
        lp, gp, W, dev = self.plan, self.global_plan, self.world, self.device
        S = lp.S
        ks = unique_key_space(gp, lp, W)
        vs = [v for _, v in ks]                                          # ids of slot s: [0, V_s) in the GLOBAL space
        if S > 128 or W * int(lp.total_rows) >= (1 << 32) or self.rec is None:
            if self._cap_from_unique:      # (cannot happen: __init__ checks the same conditions before it sizes the segments)
                raise RuntimeError("sharded engine: segments sized for distinct rows, but the unique path is not available")
            self.dedup = False
            return
        shifts, bases, nb = bucket_geometry(vs, self.max_batch, int(call("wd_bucket_max")),
                                            float(os.environ.get("WD_BUCKET_TARGET", "64")))
        arr = (capi.WdSlot * S)()
        for i, sl in enumerate(lp.slots):
            is_emb = bool(sl.deep == "embedding" and self.spec.has_deep)
            arr[i].emb_off, arr[i].row_base = 0, ks[i][0]
            arr[i].num_buckets, arr[i].dim, arr[i].out_col = vs[i], int(sl.dim) if is_emb else 0, lp.out_col[i]
            arr[i].kind, arr[i].wide = capi.SLOT_EMBEDDING if is_emb else capi.SLOT_NONE, 1 if sl.wide else 0
            arr[i].bucket_shift, arr[i].bucket_base = shifts[i], bases[i]
        i32 = dict(dtype=torch.int32, device=dev)
        n = self.max_batch * S
        self._rq = dict(slots=torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev), nb=nb,
                        max_slot_buckets=max([((v + (1 << sh) - 1) >> sh) for v, sh in zip(vs, shifts)] + [1]),
                        long_cap=n // 32 + 2)
        for xs in self._xsets:
            xs.update(rq_start=torch.zeros(2 * nb + 2, **i32), rq_pairs=torch.zeros(n, dtype=torch.int64, device=dev),
                      rq_long=torch.zeros(2 * (n // 32 + 2) + 2, **i32), rq_big=torch.zeros(nb, **i32), unique=False)

    def _dedup_ok(self, bt):
        lp = self.plan
        return (self.dedup and self._rq is not None and bt.one_hot and bt.nnz == bt.B * lp.S and self.rec is not None
                and self.dim in (4, 8, 16) and self.n_emb_slots == lp.S and self.spec.has_wide and self.spec.has_deep)

    def _route(self, bt: DeviceBatch, st):
        """A, requester side: every occurrence -> (owner segment, position); needs the ids only."""
        xs = self._xsets[self._pset]
        xs["unique"] = False
        if self._cap_from_unique and not self._dedup_ok(bt):
            raise ValueError("sharded engine: the exchange segments were sized for the DISTINCT rows of a batch "
                             "(expected_unique), which needs one id per bag in every batch; build the engine with dedup=False "
                             "(or WD_SHARD_DEDUP=0) for multi-hot batches")
        if self._dedup_ok(bt):
            # sort the batch's occurrences on (key, occurrence), then one segment entry per distinct key
            rq, S = self._rq, self.plan.S
            cols = bt.ids_cols is not None and bt.ids_cols_valid
            call("wd_bucket_onehot", ptr(rq["slots"]), S, ptr(bt.ids_cols if cols else bt.ids), 1 if cols else 0, bt.B,
                 ptr(xs["rq_start"]), ptr(xs["rq_pairs"]), rq["nb"], rq["max_slot_buckets"], None, ptr(xs["rq_long"]), st)
            call("wd_bucket_sort", ptr(xs["rq_start"]), ptr(xs["rq_pairs"]), rq["nb"], ptr(xs["rq_long"]), rq["long_cap"],
                 ptr(xs["rq_big"]), bt.B, S, None, None, None, st)
            call("wd_route_unique", ptr(xs["rq_pairs"]), bt.B * S, self.world, self.cap, ptr(self.send_rows), ptr(self.pos),
                 ptr(self.route_ws), ptr(self.peer_counts), ptr(self.overflow), st)
            xs["unique"] = True
            return
        call("wd_route_build", ptr(self.slots_small_dev if self.rep_idx else self.slots_dev), self.plan.S, self.world, ptr(bt.ids), ptr(bt.bag_offs), bt.B, self.cap,
             ptr(self.send_rows), ptr(self.pos), ptr(self.route_ws), ptr(self.peer_counts), ptr(self.overflow), st)

    def _exchange_rows(self):
        """A: the requested local rows travel to their owners."""
        send, recv = self.send_rows, self.recv_rows
        self._collective(lambda: _a2a(recv, send, None, None, self.group))

    def _owner_bucketize(self, st):
        """Owner side of the backward pass, part 1 (needs only the received row list)."""
        ob = self._obsets[self._pset]
        call("wd_sparse_bucketize", ptr(self.oslot_dev), 1, ptr(self.recv_rows), ptr(self.req_offs), self.n_req,
             self.n_req, ptr(ob["cnt"]), ptr(ob["start"]), ptr(ob["rank"]), ptr(ob["pairs"]), self.n_buckets, st)
        if self.owner_flat:
            call("wd_bucket_sort_ragged", ptr(ob["start"]), ptr(ob["pairs"]), self.n_buckets, ptr(ob["long_list"]),
                 (ob["long_list"].numel() - 2) // 2, ptr(ob["big_list"]), self.n_req, 1, self.n_req, st)

    def _owner_gather(self, st):
        """B: owners read the requested rows (+ wide weight) and send them back."""
        self._owner_gather_kernel(st)
        self._collective(lambda: _a2a(self.fwd_recv, self.fwd_send, None, None, self.group))

    def _owner_gather_kernel(self, st):
        """the owner-side gather alone (this rank's share of the embedding gather: one record per received request)"""
        spec = self.spec
        has_emb = self.n_emb_slots > 0
        if self.rec is not None:
            call("wd_owner_gather_rec", ptr(self.rec), self.rec_stride, self.dim, ptr(self.recv_rows), self.n_req,
                 ptr(self.fwd_send), self.RS, st)
        else:
            call("wd_owner_gather", ptr(self.emb) if has_emb else None, self.n_emb_rows, self.dim,
                 ptr(self.wide) if spec.has_wide else None, ptr(self.recv_rows), self.n_req, ptr(self.fwd_send), self.RS, st)

    def measure_exchange(self, iters=20):
        """Diagnostics (bench.py, after the timed region; collective -- every rank calls it): each collective of a step issued
        alone on the buffers of the last step, `iters` times back to back between two stream events, and the owner-side gather
        kernel the same way.  Returns {name: (microseconds per call, bytes this rank puts on the wire per call)}; with W ranks
        (W - 1) / W of an all-to-all's payload leaves the GPU, an all-reduce of n bytes moves 2 (W - 1) / W n."""
        W, RS, cap = self.world, self.RS, self.cap
        wire = (W - 1) / W if W > 1 else 1.0       # (one rank: RCCL's local copies, reported against the same payload)
        st = torch.cuda.current_stream().cuda_stream
        legs = [("A_rows_int32", lambda: _a2a(self.recv_rows, self.send_rows, None, None, self.group), 4 * W * cap * wire),
                ("B_records_f32", lambda: _a2a(self.fwd_recv, self.fwd_send, None, None, self.group), 4 * W * cap * RS * wire),
                ("C_gradients_f32", lambda: _a2a(self.bwd_recv, self.bwd_send, None, None, self.group), 4 * W * cap * RS * wire),
                ("D_dense_allreduce_f32", lambda: _all_reduce_sum(self._gdiag, self.group),
                 4 * self.G.numel() * (2 * wire if W > 1 else 1.0)),
                ("owner_gather_kernel", lambda: self._owner_gather_kernel(st), 0)]
        self._gdiag = torch.zeros_like(self.G)      # (a scratch copy: the gradient buffer itself stays what the step left)
        out = {}
        for name, fn, nbytes in legs:
            fn()
            torch.cuda.synchronize()
            dist.barrier(group=self.group)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            out[name] = (e0.elapsed_time(e1) / iters * 1e3, nbytes)
        del self._gdiag
        return out

    def _sparse_exchange(self, bt: DeviceBatch, st):
        if self._skip_exchange:      # ShardedStepGraph has already issued route / A / B of this step on its own branches
            return
        self._route(bt, st)
        self._exchange_rows()
        self._owner_gather(st)
        if self._train_fwd:
            # bucket the received requests on a side stream under the tower; joined at the top of backward_and_update
            main, side = torch.cuda.current_stream(), self._side(0)
            side.wait_stream(main)
            self._owner_bucketize(side.cuda_stream)
            self._bucketized = True

    def _pool(self, bt: DeviceBatch, st):
        """Requester side of the forward: pool the received rows into x, numeric columns, wide logit."""
        lp, spec = self.plan, self.spec
        B, S = bt.B, lp.S
        has_emb = self.n_emb_slots > 0
        if spec.has_deep:
            tw0 = self.towers[0]
            ld = tw0["layout"].ld
            xp = self._x_ptr(tw0)
            groups = (self.group_slots_big if self.rep_idx else self.group_slots) if has_emb else {}
            for d, gs in groups.items():      # one launch per embedding dim (exchanged columns)
                call("wd_embag_fwd_strided", ptr(self.fwd_recv), self.RS, ptr(self.xslots_dev), S, ptr(gs), gs.numel(),
                     d, ptr(self.pos), ptr(bt.bag_offs), B, xp, ld, st)
            if self.ind_xslots_dev is not None:
                call("wd_indicator_fwd", ptr(self.ind_xslots_dev), S, ptr(self.ind_slots_dev), self.ind_slots_dev.numel(),
                     ptr(bt.ids), ptr(bt.bag_offs), B, xp, ld, st)
            if self.dense_cols_dev is not None:
                call("wd_dense_fwd", ptr(bt.dense), bt.dense.stride(0), ptr(self.dense_cols_dev),
                     len(lp.dense_cols), B, xp, ld, st)
        if spec.has_wide:
            w_ptr = self.fwd_recv.data_ptr() + 4 * (self.dim if has_emb else 0)
            call("wd_wide_fwd", w_ptr, self.RS, ptr(self.bias), ptr(self.xslots_small_dev if self.rep_idx else self.xslots_dev), S,
                 ptr(self.pos), ptr(bt.bag_offs), B, ptr(self.wide_logit), st)
        if self.rep_idx:
            # replicated columns: pooled / summed from this rank's own copy of their tables (tables in LDS, csrc/small_tables.hip)
            call("wd_small_tables_fwd", ptr(self.emb) if spec.has_deep else None, ptr(self.wide) if spec.has_wide else None,
                 ptr(self.slots_dev), S, ptr(self.small_idx_dev), len(self.small_idx), self.small_rows, self.small_dim,
                 ptr(bt.ids), ptr(bt.bag_offs), B, self._x_ptr(self.towers[0]) if spec.has_deep else None,
                 self.towers[0]["layout"].ld if spec.has_deep else 0, ptr(self.wide_logit) if spec.has_wide else None, 0, st)

    def _reduce_dense_grads(self):
        self._collective(lambda: _all_reduce_sum(self.G, self.group))

    def _pack_grads(self, bt: DeviceBatch, st):
        """C: per-occurrence gradients to the owners.  Issued asynchronously: RCCL moves them while this stream goes on
        with the dense branch; `_owner_update` waits for them."""
        lp, spec = self.plan, self.spec
        has_emb = self.n_emb_slots > 0
        dx_ptr, ld = None, 0
        if has_emb:
            tw0 = self.towers[0]
            tl0 = tw0["layout"]
            dx_ptr, ld = tw0["dact"].data_ptr() + 4 * tl0.seg_start[0], tl0.ld
        xs = self._xsets[self._pset]
        if xs.get("unique"):
            # one record per distinct row: the sum over the row's occurrences in this batch (sorted pairs of _route)
            call("wd_row_grad_presum", ptr(self._rq["slots"]), lp.S, bt.B, dx_ptr, ld, ptr(self.dlogit), self.dim,
                 ptr(xs["rq_pairs"]), ptr(xs["rq_long"]), self._rq["long_cap"], ptr(self.pos), ptr(self.bwd_send), self.RS, st)
            return
        call("wd_grad_pack", ptr(self.slots_small_dev if self.rep_idx else self.slots_dev), lp.S, ptr(bt.bag_offs), ptr(self.pos), bt.B, dx_ptr, ld,
             ptr(self.dlogit) if spec.has_wide else None, self.dim, self.RS, ptr(self.bwd_send), st)

    def _send_grads(self):
        def send():
            self._work_c = _a2a(self.bwd_recv, self.bwd_send, None, None, self.group, async_op=True)
        self._collective(send)

    def _grads_to_owners(self, bt: DeviceBatch, st):
        self._pack_grads(bt, st)
        self._send_grads()

    def _wait_grads(self):
        def wait():
            if self._work_c is not None:
                self._work_c.wait()      # stream-level wait: the received gradients are complete for what follows
                self._work_c = None
        self._collective(wait)

    def _owner_apply(self, st, bucketized):
        """Owner: dedup by row + Adagrad / FTRL on the received (row, gradient) list; bias handled elsewhere."""
        spec = self.spec
        has_emb = self.n_emb_slots > 0
        lr, l1, l2 = (spec.lin_opt[1], spec.lin_opt[2], spec.lin_opt[3]) if (spec.has_wide and self.default_opts) else (0.0, 0.0, 0.0)
        g_ptr = self.bwd_recv.data_ptr()
        dl_ptr = g_ptr + 4 * (self.dim if has_emb else 0)
        ob = self._obsets[self._pset]
        if not bucketized:
            self._owner_bucketize(st)
        if not self.default_opts:
            # the generic optimizers (SGD / Adagrad / Ftrl with any lr_power / RMSProp / Adam) on the received (row, gradient) list;
            # Adam moves every row of a sparsely updated variable (tf 1.x _apply_sparse_shared): the rows of this shard no request
            # named are stepped by wd_adam_untouched (the touched bitmap is over LOCAL fused rows)
            import ctypes
            od = ctypes.byref(self.opt_c["dnn"]) if has_emb else None
            ol = ctypes.byref(self.opt_c["linear"]) if spec.has_wide else None
            call("wd_sparse_apply_opt", ptr(self.emb) if has_emb else None, ptr(self.emb_a) if has_emb else None,
                 ptr(self.emb_acc) if has_emb else None, ptr(self.wide) if spec.has_wide else None, None, ptr(self.oslot_dev), 1,
                 ptr(self.req_offs), self.n_req, g_ptr if has_emb else None, self.RS, dl_ptr if spec.has_wide else None, self.RS,
                 od, ol, ptr(ob["start"]), ptr(ob["pairs"]), self.n_buckets, ptr(self.touched), st)
            if self.pow:
                call("wd_adam_untouched", ptr(self.emb) if (has_emb and "dnn" in self.pow) else None, ptr(self.emb_a),
                     ptr(self.emb_acc), ptr(self.wide) if "linear" in self.pow else None, ptr(self.aslots_dev), 2,
                     max(self.plan.total_rows, 1), max(self.plan.total_rows, 1), ptr(self.touched), od, ol, st)
            return
        if self.owner_flat:
            call("wd_row_update_ragged", ptr(self.rec), self.rec_stride, self.dim, ptr(self.emb_acc), None, ptr(self.oslot_dev), 1,
                 self.n_req, ptr(self.req_offs), g_ptr, self.RS, dl_ptr, self.RS, float(spec.dnn_opt[1]), float(lr), float(l1),
                 float(l2), ptr(ob["pairs"]), self.n_req, ob["start"].data_ptr() + 4 * self.n_buckets, ptr(ob["long_list"]),
                 (ob["long_list"].numel() - 2) // 2, st)
            return
        if self.rec is not None:
            call("wd_sparse_apply_rec", ptr(self.rec), self.rec_stride, self.dim, ptr(self.emb_acc), None, ptr(self.oslot_dev),
                 1, ptr(self.req_offs), self.n_req, g_ptr, self.RS, dl_ptr, self.RS, float(spec.dnn_opt[1]), float(lr),
                 float(l1), float(l2), ptr(ob["start"]), ptr(ob["pairs"]), self.n_buckets, None, st)
            return
        call("wd_sparse_apply", ptr(self.emb) if has_emb else None, ptr(self.emb_acc) if has_emb else None,
             ptr(self.wide) if spec.has_wide else None, None, ptr(self.oslot_dev), 1, ptr(self.req_offs), self.n_req,
             g_ptr if has_emb else None, self.RS, dl_ptr if spec.has_wide else None, self.RS,
             float(spec.dnn_opt[1]) if spec.has_deep else 0.0, float(lr), float(l1), float(l2),
             ptr(ob["start"]), ptr(ob["pairs"]), self.n_buckets, st)

    def _owner_update(self, bt: DeviceBatch, st, do_bias=True):
        self._wait_grads()
        bucketized, self._bucketized = self._bucketized, False
        self._owner_apply(st, bucketized)
        if do_bias:
            self._bias_update(bt, st)

    def _bias_update(self, bt: DeviceBatch, st):
        spec = self.spec
        if not spec.has_wide:
            return
        # bias_weights: dense FTRL on the GLOBAL sum of dlogit
        lr, l1, l2 = (spec.lin_opt[1], spec.lin_opt[2], spec.lin_opt[3]) if self.default_opts else (0.0, 0.0, 0.0)
        if self._logits_b_off is not None:
            gsum = self.G[self._logits_b_off: self._logits_b_off + 1]     # already all-reduced with G
        else:
            torch.sum(self.dlogit[:bt.B], dim=0, keepdim=True, out=self._gsum)
            self._collective(lambda: _all_reduce_sum(self._gsum, self.group))
            gsum = self._gsum
        if self.default_opts:
            call("wd_bias_ftrl", ptr(self.bias), ptr(gsum), 1, float(lr), float(l1), float(l2), st)
        else:       # bias_weights is a dense [1] variable of the linear scope: {w, slot a, slot b, slot c} = self.bias
            import ctypes
            ob = capi.WdOpt()
            ctypes.memmove(ctypes.byref(ob), ctypes.byref(self.opt_c["linear"]), ctypes.sizeof(ob))
            ob.slot_c = self.bias.data_ptr() + 12
            self._bias_opt = ob
            call("wd_opt_dense", self.bias.data_ptr(), self.bias.data_ptr() + 4, self.bias.data_ptr() + 8, ptr(gsum), 1,
                 ctypes.byref(ob), st)

    def _sparse_backward(self, bt: DeviceBatch, st):
        self._grads_to_owners(bt, st)
        self._owner_update(bt, st)
        self._replicated_update(bt, st)

    def _small_on(self, bt):
        return bool(self.rep_idx)        # replicated columns have no owner: every batch takes the small-table path for them

    def _chain_windows_ok(self):
        return True

    def _sync_replicas(self):
        """Every rank draws its tables from its own seed stream: the replicated ones take rank 0's numbers."""
        lp = self.plan
        for i in self.rep_idx:
            if lp.emb_off[i] >= 0:
                v = self._emb_view(self.emb, i)
                t = v.contiguous()
                if dist.get_backend(self.group) == "gloo":
                    c = t.cpu()
                    dist.broadcast(c, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                    t = c.to(v.device)
                else:
                    dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                v.copy_(t)

    def _replicated_update(self, bt: DeviceBatch, st):
        """Replicated columns: this rank's gradient sums + hit counts per row -> all-reduce(SUM) -> the same Adagrad / Ftrl
        update on every rank (the copies stay identical: every rank adds the same numbers to the same state)."""
        if not self.rep_idx:
            return
        lp, spec = self.plan, self.spec
        has_emb = any(lp.emb_off[i] >= 0 for i in self.rep_idx)
        dx_ptr, ld = None, 0
        if has_emb:
            tw0 = self.towers[0]
            dx_ptr, ld = tw0["dact"].data_ptr() + 4 * tw0["layout"].seg_start[0], tw0["layout"].ld
        n = len(self.small_idx)
        if self.small_ws is None:
            nws = int(call("wd_small_tables_ws_floats", n, self.small_rows, self.small_dim, self.max_batch))
            self.small_ws = torch.zeros(max(nws, 1), dtype=torch.float32, device=self.device)
            self.small_g = torch.zeros(n * self.small_rows * (self.small_dim + 2), dtype=torch.float32, device=self.device)
        call("wd_small_tables_grad", ptr(self.slots_dev), lp.S, ptr(self.small_idx_dev), n, self.small_rows, self.small_dim,
             ptr(bt.ids), ptr(bt.bag_offs), bt.B, dx_ptr, ld, ptr(self.dlogit) if spec.has_wide else None,
             ptr(self.small_ws), self.small_ws.numel(), ptr(self.small_g), st)
        self._collective(lambda: _all_reduce_sum(self.small_g, self.group))
        lr, l1, l2 = (spec.lin_opt[1], spec.lin_opt[2], spec.lin_opt[3]) if spec.has_wide else (0.0, 0.0, 0.0)
        call("wd_small_tables_apply", ptr(self.emb) if has_emb else None, ptr(self.emb_acc) if has_emb else None,
             ptr(self.wide) if spec.has_wide else None, ptr(self.slots_dev), lp.S, ptr(self.small_idx_dev), n, self.small_rows,
             self.small_dim, ptr(self.small_g), float(spec.dnn_opt[1]) if spec.has_deep else 0.0, float(lr), float(l1), float(l2), 0, st)

    def _dense_tail(self, bt, st):
        """Products -> this rank's dense gradient -> D (all-reduce) -> Adagrad + the packed kernels of the next step."""
        tw = self.towers[0]
        self._tower_backward(tw, bt.B, st, need_dx=False, head_done=True)
        self._chain_tail(capi.WD_TAIL_GRAD, st)      # this rank's dense gradient from the split-K partials -> G

        def reduce_async():
            self._work_d = _all_reduce_sum(self.G, self.group, async_op=True)
        self._collective(reduce_async)

    def _dense_finish(self, bt, st):
        def wait_d():
            if self._work_d is not None:
                self._work_d.wait()
                self._work_d = None
        self._collective(wait_d)
        if not self.default_opts:
            import ctypes
            call("wd_opt_dense", ptr(self.P), ptr(self.Pa), ptr(self.Pacc), ptr(self.G), self.P.numel(),
                 ctypes.byref(self.opt_c["dnn_dense"]), st)
            if self._fold_at_end():
                self._chain_tail(capi.WD_TAIL_PACK, st)      # the MFMA-packed kernels the next tower launch reads
                self._folded = True
        elif self._fold_at_end():
            # Adagrad on the all-reduced gradient + the MFMA-packed kernels the next tower launch reads, one launch
            self._chain_tail(capi.WD_TAIL_UPDATE | capi.WD_TAIL_PACK, st)
            self._folded = True
        else:
            call("wd_adagrad_dense", ptr(self.P), ptr(self.Pacc), ptr(self.G), self.P.numel(), float(self.spec.dnn_opt[1]), st)
        self._bias_update(bt, st)
        for scope, pw in self.pow.items():     # Adam: beta1^t, beta2^t -> t + 1 (the base engine ticks them in backward_and_update)
            o = self.spec.dnn_opt if scope == "dnn" else self.spec.lin_opt
            call("wd_adam_tick", ptr(pw), float(o[2]), float(o[3]), st)

    def backward_and_update(self, bt: DeviceBatch, bucketized=False, pset=0, lookahead=None, before_join=None):
        """With the one-launch tower dx exists when forward() returns: the gradient exchange starts first and overlaps
        with the dense branch (weight-gradient GEMMs, all-reduce, dense tail)."""
        spec, st = self.spec, _stream()
        if self._bucketized:
            torch.cuda.current_stream().wait_stream(self._side(0))     # the early bucketing (see _sparse_exchange)
        if not (spec.has_deep and self.chain):
            return super().backward_and_update(bt, False, before_join=before_join)
        self._grads_to_owners(bt, st)
        self._dense_tail(bt, st)         # D (all-reduce of the flat dense gradient) in the background of the owners' update
        self._owner_update(bt, st, do_bias=False)
        self._replicated_update(bt, st)
        self._dense_finish(bt, st)
        if before_join is not None:
            before_join()

    def train_step(self, bt: DeviceBatch):
        self._train_fwd = True
        try:
            return super().train_step(bt)
        finally:
            self._train_fwd = False

    def _graph_mode(self):
        """How a captured step treats the collectives: "full" = RCCL calls are captured into the hipGraph with everything else
        (one launch per replay; verified on ROCm 7.2 / RCCL 2.26 with a one-rank group, scripts/rccl_graph_probe.py),
        "segments" = graphs between the collectives, which stay ordinary calls (gloo staging; WD_DIST_GRAPH=segments)."""
        mode = os.environ.get("WD_DIST_GRAPH", "full")
        return mode if dist.get_backend(self.group) == "nccl" else "segments"

    def _quiesce(self):
        torch.cuda.synchronize()
        if dist.get_backend(self.group) == "nccl":
            # The process group's watchdog thread polls the completion events of the eager collectives every 100 ms until it
            # has seen them complete.  A poll that lands inside a capture intermittently fails with hipErrorCapturedEvent on
            # ROCm 7.2 and takes the process down; the works are complete after the synchronize, so two poll periods later
            # none is left to poll.
            import time
            time.sleep(0.25)

    def capture_train_step(self, bt, warmup=2, pre=None):
        """Capture the step on `bt`'s buffers; `pre` = extra launches in front of it (e.g. the token hashing that fills
        bt.ids).  Returns a replay callable; every rank must capture and replay in lock-step."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                if pre:
                    pre()
                self.train_step(bt)
        torch.cuda.current_stream().wait_stream(side)
        if warmup:
            self._quiesce()
        bump = 3 if self.spec.model_type == "wide_deep" else 2
        if self._graph_mode() == "full":
            graph = hipgraph.new_graph()
            gs = self.global_step
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                if pre:
                    pre()
                self.train_step(bt)
            self.global_step = gs
            torch.cuda.synchronize()
            self._graph = graph

            def replay_full():
                graph.replay()
                self.global_step += bump
                return self.loss

            return replay_full
        segs = _Segments()
        with torch.cuda.stream(side):
            self._segs = segs
            segs.begin()
            try:
                if pre:
                    pre()
                self.train_step(bt)
            finally:
                segs.end()
                self._segs = None
        torch.cuda.synchronize()

        def replay():
            for op in segs.ops:
                op()
            self.global_step += bump
            return self.loss

        return replay

    # ---- state: shard <-> full tables ---------------------------------------------------------------
    def import_full_state(self, state):
        """Load a FULL (unsharded) state dict in the reference's naming, keeping rows id % world == rank."""
        W, r = self.world, self.rank
        sub = {}
        rep = self._replicated_names()
        for k, v in state.items():
            if any(k == n or k.startswith(n + "/") for n in rep):
                sub[k] = v                  # a replicated column: whole on every rank
            elif ("embedding_weights" in k or k.startswith("linear/linear_model/")) and "bias_weights" not in k:
                sh = v[r::W]
                need = shard_rows(v.shape[0], W)
                if sh.shape[0] < need:   # ranks whose last local row does not exist globally: pad (never addressed)
                    pad = torch.zeros((need - sh.shape[0],) + tuple(v.shape[1:]), dtype=v.dtype)
                    if k.endswith("/Adagrad") or k.endswith("/Ftrl") or k.endswith("/RMSProp"):
                        pad += 0.1          # (rows that exist on no rank's id space: never addressed; any positive accumulator)
                    sh = torch.cat([sh, pad], 0)
                sub[k] = sh.contiguous()
            else:
                sub[k] = v
        self.import_state(sub)

    def _replicated_names(self):
        gp = self.global_plan
        out = []
        for i in self.rep_idx:
            s = gp.slots[i]
            out += ["dnn/input_from_feature_columns/input_layer/%s/embedding_weights" % s.deep_name, "linear/linear_model/%s/weights" % s.name]
        return out

    def export_full_state(self):
        """Gather every rank's shard; returns the FULL state dict on every rank."""
        W = self.world
        local = self.export_state()
        out = {}
        gp = self.global_plan
        full_rows = {}
        for i, s in enumerate(gp.slots):
            if i in self.rep_idx:
                continue                    # whole on every rank: the local copy is the variable
            full_rows["dnn/input_from_feature_columns/input_layer/%s/embedding_weights" % s.deep_name] = s.num_buckets
            full_rows["linear/linear_model/%s/weights" % s.name] = s.num_buckets
        for k in sorted(local):
            v = local[k]
            base = k
            for suf in ("/Adagrad", "/Ftrl_1", "/Ftrl", "/RMSProp_2", "/RMSProp_1", "/RMSProp", "/Adam_1", "/Adam"):
                if base.endswith(suf):
                    base = base[: -len(suf)]
                    break
            if base in full_rows:
                if dist.get_backend(self.group) != "gloo":
                    v = v.to(self.device)                       # RCCL moves device tensors only
                parts = [torch.empty_like(v) for _ in range(W)]
                dist.all_gather(parts, v.contiguous(), group=self.group)
                parts, v = [q.cpu() for q in parts], v.cpu()
                V = full_rows[base]
                full = torch.empty((V,) + tuple(v.shape[1:]), dtype=v.dtype)
                for r in range(W):
                    n = len(range(r, V, W))
                    full[r::W] = parts[r][:n]
                out[k] = full
            else:
                out[k] = v
        return out


class ShardedStepGraph:
    """`len(token_batches)` consecutive SHARDED train steps in one hipGraph, RCCL collectives captured with the kernels
    (one launch per replay on every rank; ranks must build and replay in lock-step -- the captured collectives meet their
    peers' in issue order).  Two graph branches:

        main     owner gather(t) -> B(t): rows back to the requesters -> tower(t) (pools the received rows itself) -> pack ->
                 C(t): gradients to the owners (async) -> products -> tail(GRAD) -> D(t): all-reduce (async) ->
                 tail(UPDATE|PACK) + bias -> hash(t+1) -> route(t+1) -> A(t+1): requested rows to their owners
        sparse   owner-side bucketing of the requests of step t (under tower(t)) | wait C(t) -> owner update(t): Adagrad /
                 Ftrl on the received (row, gradient) list, beside the products and the input work of step t+1

    Needs the one-launch tower with the fused input (Criteo-shaped model, one id per bag) and the RCCL backend
    (`eng._graph_mode() == "full"`); dist.py's per-step capture_train_step covers everything else."""

    def __init__(self, eng, token_batches, ids_input=False, stream=None):
        from . import synth
        if eng._graph_mode() != "full":
            raise capi.WdError("ShardedStepGraph needs the RCCL backend with captured collectives (WD_DIST_GRAPH=full)")
        if not all(eng.chain and eng._chain_input_ok(tb.batch) and tb.batch.labels is not None for tb in token_batches):
            raise capi.WdError("ShardedStepGraph: one-launch tower with the fused input layer only")
        if not eng._folded:
            raise capi.WdError("ShardedStepGraph: run one eager train step first (packed kernels, lazy allocations)")
        self.eng, self.n = eng, len(token_batches)
        self.stream = stream or torch.cuda.Stream()
        self.graph = hipgraph.new_graph()
        self._events = []
        self.stream.wait_stream(torch.cuda.current_stream())
        eng._quiesce()
        gs = eng.global_step
        try:
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                self._capture(token_batches, ids_input, synth)
        finally:
            eng._pset, eng._skip_exchange = 0, False
            eng.global_step = gs
        torch.cuda.synchronize()
        self._bump = (3 if eng.spec.model_type == "wide_deep" else 2) * self.n

    def _capture(self, tbs, ids_input, synth):
        # (a third layout with the whole chain B -> tower -> pack -> A -> C -> owner update on the main branch was measured in round 6
        # and removed: 0.2818 against 0.2636 ms/step, profiles/r6_sharded_layout_v3_experiment.txt)
        if os.environ.get("WD_SHARD_LAYOUT", "v2") == "v2" and os.environ.get("WD_SHARD_PIPE", "1") != "0":
            return self._capture_v2(tbs, ids_input, synth)
        eng = self.eng
        main = torch.cuda.current_stream()
        n = len(tbs)
        if os.environ.get("WD_SHARD_PIPE", "1") == "0":     # plain stream order, collectives captured: no cross-step overlap
            for tb in tbs:
                if not ids_input:
                    synth.hash_tokens(eng, tb)
                eng.train_step(tb.batch)
            return
        # Two branches: main and sparse (owner-side bucketing + owner update).  What hipStreamEndCapture of ROCm 7.2 / torch 2.10
        # / RCCL 2.26 survives was found by bisection (round 3): every collective is issued from the MAIN branch; the requested
        # rows (A) travel with a SYNCHRONOUS call -- an asynchronous A whose Work is waited for on another branch segfaults the
        # end of the capture, while C and D (asynchronous, waited for once) are fine; a third branch for hash / route joined
        # in front of A segfaults as well.  The input work of step t+1 (hash, route, A) therefore sits on the main branch
        # behind the dense tail of step t, where main would otherwise idle until the owner update of step t is done, and the
        # owner-side bucketing of what arrived goes to the sparse branch, under the tower of step t+1.
        s_sp = eng._side(0)
        s_sp.wait_stream(main)
        keep = self._events

        def event(stream):
            ev = torch.cuda.Event()
            ev.record(stream)
            keep.append(ev)
            return ev

        ev_upd = [None] * n

        early = os.environ.get("WD_SHARD_ROUTE", "early") == "early"
        ev_route = [None] * n

        def route_work(t, stream):
            """hash -> route of step t (routing / bucketing set t & 1: step t-2, its last reader, is complete -- main has
            joined update(t-2) before the owner gather of step t-1, which precedes every caller)."""
            tb = tbs[t]
            eng._pset = t & 1
            with torch.cuda.stream(stream):
                if not ids_input:
                    synth.hash_tokens(eng, tb)
                eng._route(tb.batch, stream.cuda_stream)
                if stream is not main:
                    ev_route[t] = event(stream)

        def exchange_rows(t):
            """A of step t (synchronous, from the main branch), owner-side bucketing of what arrived on the sparse branch."""
            eng._pset = t & 1
            if ev_route[t] is not None:
                main.wait_event(ev_route[t])
            eng._exchange_rows()                              # A
            s_sp.wait_event(event(main))
            with torch.cuda.stream(s_sp):
                eng._owner_bucketize(s_sp.cuda_stream)

        route_work(0, main)
        exchange_rows(0)
        for t, tb in enumerate(tbs):
            bt = tb.batch
            eng._pset = t & 1
            st = main.cuda_stream
            if t >= 1:
                main.wait_event(ev_upd[t - 1])                # the rows this step reads are final
            eng._owner_gather(st)                             # + B
            if early and t + 1 < n:
                # hash / route of the NEXT step on the sparse branch, under this step's tower (behind the bucketing of this
                # step's requests, in front of the wait for C)
                s_sp.wait_event(event(main))
                route_work(t + 1, s_sp)
                eng._pset = t & 1
            eng._skip_exchange = True
            eng.forward(bt, need_loss=True)                   # the tower launch (packed kernels in place)
            eng._skip_exchange = False
            eng._grads_to_owners(bt, st)                      # pack + C (async)
            ev_pack = event(main)
            tw = eng.towers[0]
            eng._tower_backward(tw, bt.B, st, need_dx=False, head_done=True)      # weight-gradient products
            # the owner update is captured BEHIND the products (ready nodes launch in capture order; its thousands of small
            # workgroups would otherwise occupy the CUs first -- pipeline.py)
            s_sp.wait_event(ev_pack)
            with torch.cuda.stream(s_sp):
                eng._wait_grads()
                eng._owner_apply(s_sp.cuda_stream, True)
                ev_upd[t] = event(s_sp)
            eng._chain_tail(capi.WD_TAIL_GRAD, st)
            eng._work_d = _all_reduce_sum(eng.G, eng.group, async_op=True)      # D
            eng._dense_finish(bt, st)
            if t + 1 < n:
                if not early:
                    route_work(t + 1, main)                   # beside the owner update of this step
                exchange_rows(t + 1)
        main.wait_stream(s_sp)

    def _capture_v2(self, tbs, ids_input, synth):
        """Three branches; the main one carries ONLY the exchange (collectives are issued from it) and what the owner does:

            main     owner gather(t) -> B(t) | A(t+1) (under the tower) | C(t) (async) -> wait C -> owner update(t) -> D(t) (async)
            compute  tower(t) -> pack | products(t) -> tail(GRAD) | wait D -> tail(UPDATE|PACK) + bias
            input    hash(t+1) -> route(t+1) | owner-side bucketing of the requests A(t+1) delivered

        so that what separates two towers is  pack, C, owner update, owner gather, B  and nothing else."""
        eng = self.eng
        main = torch.cuda.current_stream()
        comp, s_r = eng._side(0), eng._side(1)
        comp.wait_stream(main)
        s_r.wait_stream(main)
        keep = self._events
        n = len(tbs)

        def event(stream):
            ev = torch.cuda.Event()
            ev.record(stream)
            keep.append(ev)
            return ev

        def route_work(t, stream):
            tb = tbs[t]
            eng._pset = t & 1
            with torch.cuda.stream(stream):
                if not ids_input:
                    synth.hash_tokens(eng, tb)
                eng._route(tb.batch, stream.cuda_stream)
                return event(stream)

        def exchange_rows(t, ev_route):
            """A of step t from the main branch; the owner-side bucketing of what arrived goes to the input branch."""
            eng._pset = t & 1
            if ev_route is not None:
                main.wait_event(ev_route)
            eng._exchange_rows()                              # A (synchronous)
            s_r.wait_event(event(main))
            with torch.cuda.stream(s_r):
                eng._owner_bucketize(s_r.cuda_stream)
                return event(s_r)

        route_work(0, main)
        ev_buck = exchange_rows(0, None)
        ev_dense = None
        for t, tb in enumerate(tbs):
            bt = tb.batch
            eng._pset = t & 1
            eng._owner_gather(main.cuda_stream)               # + B   (main order: behind the owner update of step t-1)
            ev_b = event(main)
            # ---- compute branch: tower(t) -> pack ---------------------------------------------------------------------------
            comp.wait_event(ev_b)                             # (its own order: behind the dense tail of step t-1)
            with torch.cuda.stream(comp):
                eng._skip_exchange = True
                eng.forward(bt, need_loss=True)               # the tower launch
                eng._skip_exchange = False
                eng._pack_grads(bt, comp.cuda_stream)
                ev_pack = event(comp)
            # ---- input branch + A of the NEXT step, under the tower -----------------------------------------------------------
            ev_buck_next = None
            if t + 1 < n:
                s_r.wait_event(ev_b)                          # routing set (t+1) & 1: step t-1 is complete
                ev_route = route_work(t + 1, s_r)
                ev_buck_next = exchange_rows(t + 1, ev_route)
                eng._pset = t & 1
            # ---- C, products, owner update ---------------------------------------------------------------------------------------
            main.wait_event(ev_pack)
            eng._send_grads()                                 # C (async)
            with torch.cuda.stream(comp):
                eng._tower_backward(eng.towers[0], bt.B, comp.cuda_stream, need_dx=False, head_done=True)   # products
                eng._chain_tail(capi.WD_TAIL_GRAD, comp.cuda_stream)
                ev_tg = event(comp)
            main.wait_event(ev_buck)
            eng._wait_grads()
            eng._owner_apply(main.cuda_stream, True)          # owner update (captured behind the products)
            main.wait_event(ev_tg)
            eng._work_d = _all_reduce_sum(eng.G, eng.group, async_op=True)      # D
            with torch.cuda.stream(comp):
                eng._dense_finish(bt, comp.cuda_stream)       # wait D, Adagrad + packed kernels, bias
            ev_buck = ev_buck_next
        main.wait_stream(comp)
        main.wait_stream(s_r)

    def replay(self):
        self.graph.replay()
        self.eng.global_step += self._bump
        return self.eng.loss
