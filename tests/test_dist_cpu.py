"""CPU (gloo, world_size 2): the all-to-all exchange plumbing of wide_deep_amd.dist against the oracle.

Each rank owns rows id % world == rank of every table.  The forward exchange must reproduce the
full-table embedding bag of the oracle, the backward exchange must deliver to every owner exactly the
(duplicate-summed) row gradients the oracle computes on the GLOBAL batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


V = [11, 20, 7]
D = 4
S = 3
B_LOC = 16


def _batch(rank):
    rng = np.random.default_rng(100 + rank)
    lens = rng.integers(0, 4, size=(B_LOC, S))
    nnz = int(lens.sum())
    slot_of = np.repeat(np.tile(np.arange(S), B_LOC), lens.reshape(-1))
    ids = (rng.integers(0, 1 << 30, size=nnz) % np.asarray(V)[slot_of]).astype(np.int64)
    offs = np.zeros(B_LOC * S + 1, dtype=np.int32)
    np.cumsum(lens.reshape(-1), out=offs[1:])
    dx = rng.standard_normal((B_LOC, S * D)).astype(np.float32)
    return ids, offs, lens, slot_of, dx


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from wide_deep_amd.dist import ExchangePlan, occurrence_slots, shard_rows
        g = torch.Generator().manual_seed(7)
        full = [torch.randn(v, D, generator=g) for v in V]
        loc_rows = [shard_rows(v, world) for v in V]
        base = np.concatenate([[0], np.cumsum(loc_rows)])[:-1]
        local = torch.zeros(int(sum(loc_rows)), D)
        for s in range(S):
            sh = full[s][rank::world]
            local[base[s]: base[s] + sh.shape[0]] = sh
        ids, offs, lens, slot_np, dx = _batch(rank)
        ids_t, offs_t = torch.from_numpy(ids), torch.from_numpy(offs)
        slot_of, ex_of, lens_t = occurrence_slots(offs_t, B_LOC, S, len(ids))
        assert np.array_equal(slot_of.numpy(), slot_np)
        ex = ExchangePlan(ids_t, slot_of, torch.from_numpy(base.astype(np.int64)), world)
        # ---- forward: owners gather, requester pools ----
        rows = ex.to_requester(local[ex.req_rows.long()])      # bucketed order
        occ_rows = rows[ex.inv]                                # back to occurrence order
        for s in range(S):
            m = slot_np == s
            # per-slot CSR of this rank's batch
            l = lens[:, s]
            o = np.zeros(B_LOC + 1, np.int32); np.cumsum(l, out=o[1:])
            exp = O.embag_fwd(full[s], ids[m], o, mean=True)
            got = torch.zeros(B_LOC, D)
            r = occ_rows[torch.from_numpy(m)]
            for b in range(B_LOC):
                if l[b]:
                    got[b] = r[o[b]:o[b + 1]].sum(0) / float(l[b]) if l[b] > 1 else r[o[b]]
            assert torch.allclose(got, exp, atol=1e-6), "fwd slot %d" % s
        # ---- backward: per-occurrence grads to owners, owners dedupe-sum ----
        bag = ex_of * S + slot_of
        scale = 1.0 / lens_t[bag].clamp_min(1).float()
        dxt = torch.from_numpy(dx)
        cols = (slot_of * D)[:, None] + torch.arange(D)[None, :]
        gocc = dxt[ex_of[:, None], cols] * scale[:, None]
        recv = ex.to_owner(gocc[ex.order].contiguous())
        acc = torch.zeros_like(local)
        acc.index_add_(0, ex.req_rows.long(), recv)
        # expectation: oracle row grads on the GLOBAL batch, restricted to my rows
        for s in range(S):
            exp = torch.zeros(V[s], D)
            for r in range(world):
                i2, o2, l2, sl2, dx2 = _batch(r)
                m = sl2 == s
                o = np.zeros(B_LOC + 1, np.int32); np.cumsum(l2[:, s], out=o[1:])
                uniq, rg = O.embag_row_grads(D, i2[m], o, torch.from_numpy(dx2[:, s * D:(s + 1) * D]).contiguous(), mean=True)
                exp[torch.from_numpy(uniq)] += rg
            mine = exp[rank::world]
            assert torch.allclose(acc[base[s]: base[s] + mine.shape[0]], mine, atol=1e-5), "bwd slot %d" % s
        # ---- dense gradient all-reduce is a SUM ----
        from wide_deep_amd.dist import _all_reduce_sum
        t = torch.full((5,), float(rank + 1))
        _all_reduce_sum(t)
        assert torch.equal(t, torch.full((5,), world * (world + 1) / 2.0))
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 4])
def test_exchange_world2_gloo(world):
    """world 2 and 4 (three peers per all-to-all, four owners per table, vocabularies not divisible by the world size)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_local_spec_and_full_state_roundtrip_shapes():
    from wide_deep_amd.dist import local_spec, shard_rows
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(n_sparse=3, buckets=101, dim=8, hidden=(8,))
    ls = local_spec(spec, 4)
    assert [s.num_buckets for s in ls.slots] == [26, 26, 26] and shard_rows(101, 4) == 26
    assert spec.slots[0].num_buckets == 101   # the global spec is untouched


def test_local_spec_keeps_the_deep_input_of_the_default_conf():
    """The repo-default conf (47 embedding columns of widths 4..32, 20 indicator columns, bucketized wide-only columns) on a
    row-sharded rank: wide / embedding rows are cut to ceil(V / world), the deep input -- and with it every tower kernel --
    keeps its shape: an indicator column still spans the whole vocabulary."""
    from wide_deep_amd.build_estimator import build_model_spec
    from wide_deep_amd.dist import local_spec, shard_rows
    from wide_deep_amd.plan import FeaturePlan
    spec = build_model_spec()
    gp = FeaturePlan(spec)
    for world in (2, 4, 8):
        lp = FeaturePlan(local_spec(spec, world))
        assert [s.name for s in lp.slots] == [s.name for s in gp.slots]
        assert lp.deep_dim == gp.deep_dim and lp.out_col == gp.out_col and lp.tf_deep_dim == gp.tf_deep_dim
        assert np.array_equal(lp.tf_input_perm, gp.tf_input_perm)
        assert [m["K"] for m in lp.layer_meta[0]] == [m["K"] for m in gp.layer_meta[0]]
        for a, b in zip(lp.slots, gp.slots):
            assert a.num_buckets == shard_rows(b.num_buckets, world)
            if b.deep == "indicator":
                assert a.ind_width == b.num_buckets
        assert lp.total_rows <= gp.total_rows // world + gp.S


# ---- the Estimator-shaped object under torch.distributed (train.py launched one process per GPU), on CPU ------------------
def _estimator_worker(rank, world, port, q, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import types
        from tests.test_estimator_host_cpu import StandInEngine, FIXTURE
        from tests.test_featurizer_host_cpu import _fake_call
        from wide_deep_amd import build_estimator as BE, dataset as DS, estimator as E, features as F
        from wide_deep_amd.read_conf import Config
        F.call = _fake_call
        torch.cuda.current_stream = lambda *a: types.SimpleNamespace(cuda_stream=0)
        torch.cuda.synchronize = lambda *a: None

        class DataParallelStandIn(StandInEngine):
            """every rank applies the SUM of all ranks' updates (what the sharded engine's exchange amounts to)"""

            def train_step(self, bt):
                before = self.w.clone()
                loss = StandInEngine.train_step(self, bt)
                delta = self.w - before
                dist.all_reduce(delta)
                self.w.copy_(before + delta)
                return loss

            def export_full_state(self):            # collective in the real engine; every rank must call it
                dist.barrier()
                return self.export_state()

            def import_full_state(self, st):
                self.w.copy_(st["w"])

        spec = BE.build_model_spec(Config(), "wide_deep")
        model_dir = os.path.join(tmp, "model")
        m = E.WideAndDeepClassifier(spec, model_dir=model_dir, runconfig=Config().runconfig, engine=DataParallelStandIn(spec, 64))
        lines = open(FIXTURE, "rb").read().splitlines()[:301]          # odd count: the last line is dropped, 150 per rank
        path = os.path.join(tmp, "rows.tsv")
        if rank == 0:
            open(path, "wb").write(b"\n".join(lines) + b"\n")
        dist.barrier()
        seen = [r.B for r in DS.input_fn(path, None, "eval", 64)]
        assert seen == [64, 64, 22], seen                                # 150 lines on every rank
        m.train(input_fn=lambda: DS.input_fn(path, None, "train", 64))
        assert m.engine.global_step == 9 and m.last_train["examples"] == 150
        ws = [torch.zeros_like(m.engine.w) for _ in range(world)]
        dist.all_gather(ws, m.engine.w)
        assert torch.equal(ws[0], ws[1])                                 # one model on both ranks
        dist.barrier()
        ck = sorted(f for f in os.listdir(model_dir) if f.startswith("model.ckpt-"))
        assert ck == ["model.ckpt-9.pt"]                                 # written once, by rank 0
        # a fresh object on every rank resumes from that checkpoint
        m2 = E.WideAndDeepClassifier(spec, model_dir=model_dir, runconfig=Config().runconfig, engine=DataParallelStandIn(spec, 64))
        m2.train(input_fn=lambda: DS.input_fn(path, None, "train", 64), steps=1)
        assert m2.engine.global_step == 12
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_estimator_world2_gloo_lines_checkpoints_and_resume(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_estimator_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_unique_key_space_encodes_owner_and_local_row():
    """Sender-side unique: key = world * local_row_base(slot) + id  must give  key % world = owner  and  key // world = the
    owner's local row for every id of every slot, slots must not overlap, for ragged vocabularies and every world size."""
    import numpy as np
    from wide_deep_amd.dist import local_spec, unique_key_space
    from wide_deep_amd.plan import FeaturePlan, criteo_spec
    spec = criteo_spec(n_dense=2, n_sparse=6, buckets=101, dim=16, hidden=(32, 16))
    for sl, v in zip(spec.slots, (101, 57, 300, 23, 1, 4096)):
        sl.num_buckets = v
    gp = FeaturePlan(spec)
    for W in (1, 2, 3, 4, 8, 16):
        lp = FeaturePlan(local_spec(spec, W))
        ks = unique_key_space(gp, lp, W)
        hi = 0
        for i, (base, V) in enumerate(ks):
            assert V == int(gp.slots[i].num_buckets) and base >= hi, "slot %d starts inside the previous slot's keys" % i
            ids = np.arange(V, dtype=np.int64)
            key = base + ids
            assert np.array_equal(key % W, ids % W), "owner"
            assert np.array_equal(key // W, lp.row_base[i] + ids // W), "local row on the owner"
            assert (ids // W).max() < int(lp.slots[i].num_buckets), "local row inside the owner's share of the slot"
            hi = int(key.max()) + 1
        assert hi <= W * int(lp.total_rows)
