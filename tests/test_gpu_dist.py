"""GPU: the sharded HIP engine with world_size 2 (both ranks share cuda:0, gloo staging through host)
must train exactly like ONE engine on the concatenated global batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spec(kind):
    from wide_deep_amd.plan import criteo_spec
    if kind == "multihot":
        return criteo_spec(n_dense=2, n_sparse=4, buckets=101, dim=16, hidden=(16, 8), mode="resnet"), 3
    if kind == "wideonly":
        return criteo_spec(n_dense=0, n_sparse=6, buckets=97, dim=16, hidden=(8,), model_type="wide"), 2
    if kind == "deeponly":
        return criteo_spec(n_dense=2, n_sparse=3, buckets=50, dim=32, hidden=(16,), model_type="deep"), 4
    if kind in ("chain", "chain_graph"):   # one-launch tower (widths % 32 == 0): gradient exchange overlapped with the dense branch
        return criteo_spec(n_dense=16, n_sparse=3, buckets=300, dim=16, hidden=(64, 32)), 1
    return criteo_spec(n_dense=3, n_sparse=5, buckets=300, dim=16, hidden=(32, 16)), 1


def _batches(plan, kind, steps, B_loc, world):
    from wide_deep_amd import synth
    _, mean_len = _spec(kind)
    out = []
    for st in range(steps):
        out.append([synth.make_raw_batch(plan, B_loc, seed=1000 * st + r, mean_len=mean_len, pos_rate=0.3) for r in range(world)])
    return out


def _worker(rank, world, port, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from wide_deep_amd import synth
        from wide_deep_amd.dist import ShardedWideDeepEngine
        from wide_deep_amd.engine import WideDeepEngine
        from tests.helpers import assert_close
        spec, _ = _spec(kind)
        B_loc, steps = 48, 3
        ref = WideDeepEngine(spec, max_batch=B_loc * world, seed=11)
        full0 = ref.export_state()
        sh = ShardedWideDeepEngine(spec, max_batch=B_loc, seed=11)
        sh.import_full_state(full0)
        bs = _batches(ref.plan, kind, steps, B_loc, world)
        if kind.startswith("chain"):
            assert sh.chain and ref.chain
        replay = None
        for st in range(steps):
            hbs = bs[st]
            # single engine: global batch = rank0's examples then rank1's
            glob = {"B": B_loc * world, "lens": np.concatenate([h["lens"] for h in hbs], 0),
                    "raw": np.concatenate([h["raw"] for h in hbs]), "dense": None if hbs[0]["dense"] is None else np.concatenate([h["dense"] for h in hbs], 0),
                    "labels": np.concatenate([h["labels"] for h in hbs])}
            ref.train_step(synth.to_device_ids(ref.plan, glob))
            if kind == "chain_graph":
                # graph segments between the collectives: capture once on fixed buffers, refresh their contents per step
                nb = synth.to_device_ids(sh.global_plan, hbs[rank])
                if replay is None:
                    bt = nb
                    snap = {k: v.clone() for k, v in sh.export_state().items()}
                    replay = sh.capture_train_step(bt, warmup=1)
                    sh.import_state(snap)           # undo the warm-up step
                    sh.global_step = int(snap["global_step"])
                else:
                    assert nb.ids.numel() == bt.ids.numel()
                    bt.ids.copy_(nb.ids); bt.bag_offs.copy_(nb.bag_offs); bt.labels.copy_(nb.labels)
                    if nb.dense is not None:
                        bt.dense.copy_(nb.dense)
                replay()
            else:
                sh.train_step(synth.to_device_ids(sh.global_plan, hbs[rank]))
            torch.cuda.synchronize()
            assert_close(sh.logit[:B_loc], ref.logit[rank * B_loc:(rank + 1) * B_loc], 1e-4, 1e-5, "logits step %d" % st)
            sh.check_overflow()
        full1, exp = sh.export_full_state(), ref.export_state()
        for k, v in exp.items():
            if k == "global_step":
                continue
            assert_close(full1[k], v, 2e-4, 1e-5, k)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def _overflow_worker(rank, world, port, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from wide_deep_amd import synth
        from wide_deep_amd.capi import WdError
        from wide_deep_amd.dist import ShardedWideDeepEngine
        spec, _ = _spec("onehot")
        # capacity sized for 1/10 of the real occurrences: every peer segment overflows
        sh = ShardedWideDeepEngine(spec, max_batch=64, seed=1, expected_nnz=32, slack=1.0)
        hb = synth.make_raw_batch(sh.global_plan, 64, seed=rank)
        sh.train_step(synth.to_device_ids(sh.global_plan, hb))
        torch.cuda.synchronize()
        try:
            sh.check_overflow()
            q.put((rank, "FAIL: overflow not reported"))
        except WdError:
            q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def _run(worker, kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_exchange_overflow_is_reported():
    _run(_overflow_worker, "onehot")


@pytest.mark.parametrize("kind", ["onehot", "multihot", "wideonly", "deeponly", "chain", "chain_graph"])
def test_sharded_world2_equals_single_engine(kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
