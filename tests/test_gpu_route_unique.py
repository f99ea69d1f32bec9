"""GPU: the sender-side unique of the sharded row exchange at kernel level (csrc/dist_exchange.hip wd_route_unique,
csrc/onehot_path.hip wd_row_grad_presum) against a numpy restatement: one segment entry per DISTINCT key in key order,
every occurrence of the key pointing at it, overflow beyond the segment capacity reported and dropped (pos = -1), and one
pre-summed gradient record per entry -- short rows, rows of more than 32 occurrences (the long-row workers), dropped rows."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _numpy_route(pairs, W, cap):
    keys = (pairs >> np.uint64(32)).astype(np.int64)
    occ = (pairs & np.uint64(0xFFFFFFFF)).astype(np.int64)
    n = len(pairs)
    send = np.full(W * cap, -1, dtype=np.int32)
    pos = np.full(n, -1, dtype=np.int32)
    cnt = np.zeros(W, dtype=np.int64)
    heads = np.flatnonzero(np.r_[True, keys[1:] != keys[:-1]])
    ends = np.r_[heads[1:], n]
    for h, e in zip(heads, ends):
        o = int(keys[h] % W)
        r = int(cnt[o])
        cnt[o] += 1
        if r < cap:
            send[o * cap + r] = keys[h] // W
            pos[occ[h:e]] = o * cap + r
    return send, pos, cnt, heads, ends


@pytest.mark.parametrize("W,tight", [(1, False), (2, False), (8, False), (4, True)])
def test_route_unique_and_grad_presum_match_numpy(W, tight):
    from wide_deep_amd import capi
    from wide_deep_amd.capi import call, ptr
    rng = np.random.default_rng(7 + W)
    B, S, D, RS = 700, 4, 16, 20
    vocab = [60, 9, 3, 4000]                                   # slot 2: three rows -> ~230 occurrences each (long rows)
    lbase = np.cumsum([0] + [-(-v // W) for v in vocab])        # local row base of a slot on every owner
    ids = np.stack([rng.integers(0, v, B) for v in vocab], axis=1)
    key = (W * lbase[:S][None, :] + ids).astype(np.uint64)
    occ = (np.arange(B)[:, None] * S + np.arange(S)[None, :]).astype(np.uint64)
    pairs = np.sort(((key << np.uint64(32)) | occ).reshape(-1))
    n = B * S
    nuniq = len(np.unique(pairs >> np.uint64(32)))
    cap = max(8, nuniq // W // 2) if tight else n
    want_send, want_pos, cnt, heads, ends = _numpy_route(pairs, W, cap)

    dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream
    d_pairs = torch.from_numpy(pairs.view(np.int64)).to(dev)
    send = torch.zeros(W * cap, dtype=torch.int32, device=dev)
    pos = torch.full((n,), -7, dtype=torch.int32, device=dev)
    ws = torch.zeros(int(call("wd_route_chunks")) * W, dtype=torch.int32, device=dev)
    peer = torch.zeros(W, dtype=torch.int32, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    call("wd_route_unique", ptr(d_pairs), n, W, cap, ptr(send), ptr(pos), ptr(ws), ptr(peer), ptr(ovf), st)
    torch.cuda.synchronize()
    assert np.array_equal(peer.cpu().numpy(), cnt.astype(np.int32)), "distinct rows per owner"
    assert np.array_equal(send.cpu().numpy(), want_send), "request segments"
    assert np.array_equal(pos.cpu().numpy(), want_pos), "entry of every occurrence"
    assert int(ovf.item()) == (int(cnt.max()) if cnt.max() > cap else 0), "overflow flag"
    if tight:
        assert (want_pos < 0).any() and (want_pos >= 0).any()

    # ---- one pre-summed gradient record per entry ---------------------------------------------------------------------
    ldx = S * D
    dx = rng.standard_normal((B, ldx)).astype(np.float32)
    dl = rng.standard_normal(B).astype(np.float32)
    arr = (capi.WdSlot * S)()
    for s in range(S):
        arr[s].emb_off, arr[s].row_base, arr[s].num_buckets = 0, int(W * lbase[s]), vocab[s]
        arr[s].dim, arr[s].out_col, arr[s].kind, arr[s].wide = D, s * D, capi.SLOT_EMBEDDING, 1
    slots = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
    lens = ends - heads
    longs = [(int(h), int(m)) for h, m in zip(heads, lens) if m > 32]
    assert longs, "the long-row path must be exercised"
    long_cap = n // 32 + 2
    ll = np.zeros(2 * long_cap + 2, dtype=np.int32)
    ll[0] = len(longs)
    for q, (h, m) in enumerate(longs):
        ll[2 + 2 * q], ll[3 + 2 * q] = h, m
    out = torch.full((W * cap * RS,), 123.0, dtype=torch.float32, device=dev)
    d_dx, d_dl, d_ll = torch.from_numpy(dx).to(dev), torch.from_numpy(dl).to(dev), torch.from_numpy(ll).to(dev)   # (kept alive)
    call("wd_row_grad_presum", ptr(slots), S, B, ptr(d_dx), ldx, ptr(d_dl), D, ptr(d_pairs), ptr(d_ll), long_cap, ptr(pos),
         ptr(out), RS, st)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(W * cap, RS)
    occs = (pairs & np.uint64(0xFFFFFFFF)).astype(np.int64)
    written = np.zeros(W * cap, dtype=bool)
    for h, e in zip(heads, ends):
        p = int(want_pos[occs[h]])
        if p < 0:
            continue
        written[p] = True
        b, s = occs[h:e] // S, int(occs[h] % S)
        g = dx[b, s * D:(s + 1) * D].astype(np.float64).sum(0)
        np.testing.assert_allclose(got[p, :D], g, rtol=2e-5, atol=2e-5, err_msg="row gradient, %d occurrences" % (e - h))
        np.testing.assert_allclose(got[p, D], dl[b].astype(np.float64).sum(), rtol=2e-5, atol=2e-5)
    assert np.all(got[~written] == 123.0), "entries no row owns must stay untouched"
