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

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .capi import call, ptr
from .engine import DeviceBatch, WideDeepEngine, _stream
from .plan import CatSlot, FeaturePlan, ModelSpec


# ---------------------------------------------------------------------------------------------
# exchange plumbing (any device, any backend)
# ---------------------------------------------------------------------------------------------
def _a2a(out, inp, out_splits, in_splits, group=None):
    """all_to_all_single that also works for CUDA tensors on the gloo backend (host staging)."""
    backend = dist.get_backend(group)
    if inp.is_cuda and backend == "gloo":
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_to_all_single(o, i, out_splits, in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)
    return out


def _all_reduce_sum(t, group=None):
    backend = dist.get_backend(group)
    if t.is_cuda and backend == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


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


def local_spec(spec: ModelSpec, world):
    """Same model, every categorical column holding only this rank's rows (ceil(V / world))."""
    slots = []
    for s in spec.slots:
        d = CatSlot(**{k: getattr(s, k) for k in s.__dataclass_fields__})
        d.num_buckets = shard_rows(s.num_buckets, world)
        slots.append(d)
    return ModelSpec(model_type=spec.model_type, slots=slots, dense_cols=list(spec.dense_cols),
                     towers=list(spec.towers), activation=spec.activation, batch_norm=spec.batch_norm,
                     dropout=spec.dropout, dnn_opt=spec.dnn_opt, lin_opt=spec.lin_opt,
                     use_weight_column=spec.use_weight_column, pos_weight=spec.pos_weight, neg_weight=spec.neg_weight)


# ---------------------------------------------------------------------------------------------
# HIP engine
# ---------------------------------------------------------------------------------------------
class ShardedWideDeepEngine(WideDeepEngine):
    """WideDeepEngine whose tables hold rows id % world == rank; batches are this rank's examples."""

    def __init__(self, spec: ModelSpec, max_batch=8192, max_nnz=None, device="cuda", seed=0, group=None):
        if not dist.is_initialized():
            raise RuntimeError("ShardedWideDeepEngine needs torch.distributed to be initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.global_spec = spec
        self.global_plan = FeaturePlan(spec)
        gp = self.global_plan
        dims = {int(s.dim) for s in gp.slots if s.deep == "embedding"} if spec.has_deep else set()
        if len(dims) > 1:
            raise NotImplementedError("sharded engine: embedding slots must share one dim (got %s)" % sorted(dims))
        if spec.has_deep and gp.ind_slots:
            raise NotImplementedError("sharded engine: indicator columns are not exchanged yet")
        S = gp.S
        mn = int(max_nnz) if max_nnz else int(max_batch) * max(S, 1) * 8
        # owner side may receive up to world * (requester nnz) occurrences in the worst case; size for 2x the mean
        super().__init__(local_spec(spec, self.world), max_batch=max_batch, max_nnz=2 * mn, device=device, seed=seed)
        self.req_max_nnz = mn
        self.dim = dims.pop() if dims else 0
        dev = self.device
        self.local_row_base = torch.tensor(self.plan.row_base, dtype=torch.int64, device=dev)
        self.n_emb_slots = self.plan.n_emb if spec.has_deep else 0
        self.emb_slot_mask = torch.tensor([1 if (s.deep == "embedding" and spec.has_deep) else 0 for s in self.plan.slots],
                                          dtype=torch.bool, device=dev)
        self.wide_slot_mask = torch.tensor([1 if (spec.has_wide and s.wide) else 0 for s in self.plan.slots],
                                           dtype=torch.bool, device=dev)
        self.out_col_t = torch.tensor(self.plan.out_col, dtype=torch.int64, device=dev)

        # requester-side descriptors: "table" = rows that came back, ids = position in that buffer
        def make_slots(entries):
            arr = (capi.WdSlot * len(entries))()
            for i, e in enumerate(entries):
                for k, v in e.items():
                    setattr(arr[i], k, v)
            return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)

        xs = []
        for i, s in enumerate(self.plan.slots):
            is_emb = bool(s.deep == "embedding" and spec.has_deep)
            xs.append(dict(emb_off=0 if is_emb else -1, row_base=0, num_buckets=1 << 30, dim=int(s.dim) if is_emb else 0,
                           out_col=self.plan.out_col[i], kind=capi.SLOT_EMBEDDING if is_emb else capi.SLOT_NONE,
                           wide=1 if (spec.has_wide and s.wide) else 0))
        self.xslots_dev = make_slots(xs)
        # owner-side pseudo slot: the whole local fused row space is one slot
        self.oslot_dev = make_slots([dict(emb_off=0, row_base=0, num_buckets=1 << 30, dim=self.dim, out_col=0,
                                          kind=capi.SLOT_EMBEDDING, wide=1)])
        self.zero1 = torch.zeros(4, dtype=torch.float32, device=dev)
        self.one_slot = torch.zeros(1, dtype=torch.int32, device=dev)
        self._ex_emb = self._ex_wide = None

    # ---- requester / owner kernels ---------------------------------------------------------
    def _owner_gather_emb(self, req):
        n = req.numel()
        out = torch.empty(n, self.dim, dtype=torch.float32, device=self.device)
        if n:
            offs = torch.arange(n + 1, dtype=torch.int32, device=self.device)
            call("wd_embag_fwd", ptr(self.emb), ptr(self.oslot_dev), 1, ptr(self.one_slot), 1, self.dim, ptr(req),
                 ptr(offs), n, ptr(out), self.dim, _stream())
        return out

    def _owner_gather_wide(self, req):
        n = req.numel()
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        if n:
            offs = torch.arange(n + 1, dtype=torch.int32, device=self.device)
            call("wd_wide_fwd", ptr(self.wide), 4, ptr(self.zero1), ptr(self.oslot_dev), 1, ptr(req), ptr(offs), n,
                 ptr(out), _stream())
        return out

    def _owner_update(self, req, grads, is_emb):
        """Deduplicate by row (sort + segment sum) and apply Adagrad (embedding rows) or FTRL (wide rows)."""
        n = req.numel()
        if n == 0:
            return
        if n > self.max_nnz:
            raise ValueError("owner received %d occurrences, capacity %d" % (n, self.max_nnz))
        st = _stream()
        offs = torch.arange(n + 1, dtype=torch.int32, device=self.device)
        call("wd_build_sort_keys", ptr(self.oslot_dev), 1, ptr(req), ptr(offs), n, n, ptr(self.keys), ptr(self.vals), st)
        call("wd_sort_pairs", ptr(self.keys), ptr(self.vals), ptr(self.keys_sorted), ptr(self.vals_sorted), n,
             self.plan.key_bits, ptr(self.sort_ws), self.sort_ws_bytes, st)
        if is_emb:
            call("wd_embag_bwd_adagrad", ptr(self.emb), ptr(self.emb_acc), ptr(self.oslot_dev), 1, self.dim,
                 ptr(self.keys_sorted), ptr(self.vals_sorted), n, ptr(offs), ptr(grads), self.dim,
                 float(self.spec.dnn_opt[1]), st)
        else:
            _, lr, l1, l2, _ = self.spec.lin_opt
            call("wd_wide_bwd_ftrl", ptr(self.wide), ptr(self.oslot_dev), 1, ptr(self.keys_sorted), ptr(self.vals_sorted),
                 n, ptr(grads), float(lr), float(l1), float(l2), st)

    # ---- overrides ------------------------------------------------------------------------------
    def _sparse_forward(self, bt: DeviceBatch, st):
        plan, spec = self.plan, self.spec
        B, S = bt.B, plan.S
        ids = bt.ids[: bt.nnz]
        slot_of, ex_of, lens = occurrence_slots(bt.bag_offs, B, S, bt.nnz)
        self._occ = (slot_of, ex_of, lens)
        if spec.has_deep:
            tw0 = self.towers[0]
            ld = tw0["layout"].ld
            xp = self._x_ptr(tw0)
            if self.n_emb_slots:
                m = self.emb_slot_mask[slot_of]
                all_emb = bool(self.n_emb_slots == S)
                sel = None if all_emb else m.nonzero().squeeze(1)
                ex = ExchangePlan(ids if all_emb else ids[sel], slot_of if all_emb else slot_of[sel],
                                  self.local_row_base, self.world, self.group)
                rows = ex.to_requester(self._owner_gather_emb(ex.req_rows))         # [n, D], bucketed order
                # position of occurrence j inside `rows`
                pos = torch.zeros(bt.nnz, dtype=torch.int32, device=self.device)
                if all_emb:
                    pos = ex.inv.to(torch.int32)
                else:
                    pos[sel] = ex.inv.to(torch.int32)
                self._ex_emb = (ex, sel)
                gs = next(iter(self.group_slots.values()))
                call("wd_embag_fwd", ptr(rows), ptr(self.xslots_dev), S, ptr(gs), gs.numel(), self.dim, ptr(pos),
                     ptr(bt.bag_offs), B, xp, ld, st)
            if self.dense_cols_dev is not None:
                call("wd_dense_fwd", ptr(bt.dense), bt.dense.stride(0), ptr(self.dense_cols_dev),
                     len(plan.dense_cols), B, xp, ld, st)
        if spec.has_wide:
            all_wide = bool(self.wide_slot_mask.all())
            sel = None if all_wide else self.wide_slot_mask[slot_of].nonzero().squeeze(1)
            reuse = spec.has_deep and all_wide and self.n_emb_slots == S
            if reuse:   # same occurrence set as the embedding exchange: reuse its routing
                ex = self._ex_emb[0]
            else:
                ex = ExchangePlan(ids if all_wide else ids[sel], slot_of if all_wide else slot_of[sel],
                                  self.local_row_base, self.world, self.group)
            w = ex.to_requester(self._owner_gather_wide(ex.req_rows))               # [n]
            pos = torch.zeros(bt.nnz, dtype=torch.int32, device=self.device)
            if all_wide:
                pos = ex.inv.to(torch.int32)
            else:
                pos[sel] = ex.inv.to(torch.int32)
            self._ex_wide = (ex, sel)
            call("wd_wide_fwd", ptr(w), 1, ptr(self.bias), ptr(self.xslots_dev), S, ptr(pos), ptr(bt.bag_offs), B,
                 ptr(self.wide_logit), st)

    def _reduce_dense_grads(self):
        _all_reduce_sum(self.G, self.group)

    def _sparse_backward(self, bt: DeviceBatch, st):
        plan, spec = self.plan, self.spec
        B = bt.B
        slot_of, ex_of, lens = self._occ
        if spec.has_deep and self.n_emb_slots:
            ex, sel = self._ex_emb
            tw0 = self.towers[0]
            tl0 = tw0["layout"]
            so = slot_of if sel is None else slot_of[sel]
            eo = ex_of if sel is None else ex_of[sel]
            bag = eo * plan.S + so
            scale = 1.0 / lens[bag].clamp_min(1).to(torch.float32)
            cols = (tl0.seg_start[0] + self.out_col_t[so])[:, None] + torch.arange(self.dim, device=self.device)[None, :]
            g = tw0["dact"][eo[:, None], cols] * scale[:, None]                       # [n, D] per-occurrence row grads
            self._owner_update(ex.req_rows, ex.to_owner(g[ex.order].contiguous()), True)
        if spec.has_wide:
            ex, sel = self._ex_wide
            eo = ex_of if sel is None else ex_of[sel]
            gw = self.dlogit[eo][ex.order].contiguous()
            self._owner_update(ex.req_rows, ex.to_owner(gw), False)
            # bias_weights: dense FTRL on the GLOBAL sum of dlogit
            gsum = self.dlogit[:B].sum().reshape(1)
            _all_reduce_sum(gsum, self.group)
            _, lr, l1, l2, _ = spec.lin_opt
            call("wd_bias_ftrl", ptr(self.bias), ptr(gsum), 1, float(lr), float(l1), float(l2), st)

    def capture_train_step(self, bt, warmup=2):
        raise NotImplementedError("the sharded step has host-visible all-to-all split sizes; it is launched eagerly")

    # ---- state: shard <-> full tables ---------------------------------------------------------------
    def import_full_state(self, state):
        """Load a FULL (unsharded) state dict in the reference's naming, keeping rows id % world == rank."""
        W, r = self.world, self.rank
        sub = {}
        for k, v in state.items():
            if ("embedding_weights" in k or k.startswith("linear/linear_model/")) and "bias_weights" not in k:
                sh = v[r::W]
                need = shard_rows(v.shape[0], W)
                if sh.shape[0] < need:   # ranks whose last local row does not exist globally: pad (never addressed)
                    pad = torch.zeros((need - sh.shape[0],) + tuple(v.shape[1:]), dtype=v.dtype)
                    if k.endswith("/Adagrad") or k.endswith("/Ftrl"):
                        pad += 0.1
                    sh = torch.cat([sh, pad], 0)
                sub[k] = sh.contiguous()
            else:
                sub[k] = v
        self.import_state(sub)

    def export_full_state(self):
        """Gather every rank's shard; returns the FULL state dict on every rank."""
        W = self.world
        local = self.export_state()
        out = {}
        gp = self.global_plan
        full_rows = {}
        for s in gp.slots:
            full_rows["dnn/input_from_feature_columns/input_layer/%s/embedding_weights" % s.deep_name] = s.num_buckets
            full_rows["linear/linear_model/%s/weights" % s.name] = s.num_buckets
        for k in sorted(local):
            v = local[k]
            base = k.replace("/Adagrad", "").replace("/Ftrl_1", "").replace("/Ftrl", "")
            if base in full_rows:
                parts = [torch.empty_like(v) for _ in range(W)]
                dist.all_gather(parts, v.contiguous(), group=self.group)
                V = full_rows[base]
                full = torch.empty((V,) + tuple(v.shape[1:]), dtype=v.dtype)
                for r in range(W):
                    n = len(range(r, V, W))
                    full[r::W] = parts[r][:n]
                out[k] = full
            else:
                out[k] = v
        return out
