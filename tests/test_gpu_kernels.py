"""GPU: each hot-path kernel, called through the C ABI, against the CPU oracle.
Integer work is compared bit-exactly; fp32 work within the stated tolerance."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-5, 1e-6   # fp32 tolerance for sums whose order differs from the oracle's


def _dev(a, dt):
    return torch.as_tensor(np.asarray(a), dtype=dt).cuda()


def _rand_tokens(rnd, n, maxlen):
    toks = []
    for _ in range(n):
        L = rnd.choice([0, 1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 31, 32, 33, 36, 54, 63, 64, 65, 100, 127, 128, 129, 200]) if rnd.random() < 0.5 else rnd.randint(0, maxlen)
        toks.append(bytes(rnd.getrandbits(8) for _ in range(L)))
    return toks


def test_fingerprint64_bit_exact_all_length_branches():
    from wide_deep_amd.capi import call, ptr
    rnd = random.Random(3)
    toks = _rand_tokens(rnd, 5000, 300) + [str(i).encode() for i in range(2000)]
    data, offs = O.pack_tokens(toks)
    exp = O.fingerprint64_batch(data, offs)
    d = _dev(data if data.size else np.zeros(1, np.uint8), torch.uint8)
    o = _dev(offs, torch.int32)
    out = torch.zeros(len(toks), dtype=torch.int64, device="cuda")
    call("wd_fingerprint64", ptr(d), ptr(o), len(toks), ptr(out), torch.cuda.current_stream().cuda_stream)
    got = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, exp)


def test_golden_kat_on_device():
    import json
    from wide_deep_amd.capi import call, ptr
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_hash.json")))
    toks = list(g["fingerprint64"].keys())
    data, offs = O.pack_tokens(toks)
    out = torch.zeros(len(toks), dtype=torch.int64, device="cuda")
    d, o = _dev(data, torch.uint8), _dev(offs, torch.int32)
    call("wd_fingerprint64", ptr(d), ptr(o), len(toks), ptr(out), torch.cuda.current_stream().cuda_stream)
    assert out.cpu().numpy().view(np.uint64).tolist() == [g["fingerprint64"][t] for t in toks]


def test_cityhash_le32_vectors_on_device():
    """17..32-byte branch against the independent CityHash64 build (tests/golden/kat_city_le32.json)"""
    import json
    from wide_deep_amd.capi import call, ptr
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_city_le32.json")))["vectors"]
    toks = [bytes.fromhex(hx) for hx, _ in g]
    data, offs = O.pack_tokens(toks)
    out = torch.zeros(len(toks), dtype=torch.int64, device="cuda")
    d, o = _dev(data, torch.uint8), _dev(offs, torch.int32)
    call("wd_fingerprint64", ptr(d), ptr(o), len(toks), ptr(out), torch.cuda.current_stream().cuda_stream)
    assert out.cpu().numpy().view(np.uint64).tolist() == [e for _, e in g]


def test_published_fingerprint_chain_on_device():
    """Guava's testMultipleLengths (3200 chained fingerprints, lengths 0..3199, every branch) folded from DEVICE results"""
    from tests.helpers import GUAVA_SIMPLE, BIGQUERY_DOC, GUAVA_MULTIPLE_LENGTHS, guava_multiple_lengths
    from wide_deep_amd.capi import call, ptr
    rec = []
    guava_multiple_lengths(O.fingerprint64, rec)
    toks = rec + [s for s, _ in GUAVA_SIMPLE + BIGQUERY_DOC]
    data, offs = O.pack_tokens(toks)
    out = torch.zeros(len(toks), dtype=torch.int64, device="cuda")
    d, o = _dev(data, torch.uint8), _dev(offs, torch.int32)
    call("wd_fingerprint64", ptr(d), ptr(o), len(toks), ptr(out), torch.cuda.current_stream().cuda_stream)
    got = out.cpu().numpy().view(np.uint64).tolist()
    it = iter(got[:3200])
    assert guava_multiple_lengths(lambda m: next(it)) == GUAVA_MULTIPLE_LENGTHS
    assert got[3200:] == [e for _, e in GUAVA_SIMPLE + BIGQUERY_DOC]


def _engine(spec, **kw):
    from wide_deep_amd.engine import WideDeepEngine
    return WideDeepEngine(spec, **kw)


def test_hash_bucket_multi_slot_and_multi_token():
    from wide_deep_amd import synth
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(n_dense=2, n_sparse=5, buckets=1000, dim=8, hidden=(16,))
    spec.slots[1].num_buckets = 77
    spec.slots[3].num_buckets = 100003
    eng = _engine(spec, max_batch=64)
    for mean_len in (1, 3):
        hb = synth.make_raw_batch(eng.plan, 37, seed=5, mean_len=mean_len, n_raw=10**9)
        tb = synth.TokenBatch(eng.plan, hb)
        bt = synth.hash_tokens(eng, tb)
        torch.cuda.synchronize()
        S = eng.plan.S
        slot_of = np.repeat(np.tile(np.arange(S), hb["B"]), hb["lens"].reshape(-1))
        exp = np.zeros(len(hb["raw"]), dtype=np.int64)
        for si, s in enumerate(eng.plan.slots):
            m = slot_of == si
            exp[m] = O.hash_bucket([str(v) for v in hb["raw"][m]], s.num_buckets)
        assert np.array_equal(bt.ids.cpu().numpy().astype(np.int64), exp)


def test_cross_hash_bit_exact_ragged_and_empty():
    from wide_deep_amd import capi
    from wide_deep_amd.capi import call, ptr
    rnd = np.random.default_rng(11)
    B, S, slot = 50, 3, 1
    keys_host = []
    for k in range(3):
        lens = rnd.integers(0, 4, size=B)
        if k == 2:
            lens = np.maximum(lens, 1)
        offs = np.zeros(B + 1, np.int32); np.cumsum(lens, out=offs[1:])
        vals = rnd.integers(0, 2**63, size=int(offs[-1]), dtype=np.int64).astype(np.uint64)
        keys_host.append((vals, offs))
    for nb in (7, 100, 1000003):
        exp_ids, exp_offs = O.cross_hash(keys_host, nb)
        # bag offsets: slot 1 of 3 gets the cross, the others are empty
        lens_all = np.zeros((B, S), dtype=np.int64)
        lens_all[:, slot] = np.diff(exp_offs)
        bag_offs = np.zeros(B * S + 1, np.int32); np.cumsum(lens_all.reshape(-1), out=bag_offs[1:])
        ck = capi.WdCrossKeys()
        keep = []
        for k, (v, o) in enumerate(keys_host):
            vd = _dev(v.view(np.int64) if v.size else np.zeros(1, np.int64), torch.int64); od = _dev(o, torch.int32)
            keep += [vd, od]
            ck.vals[k] = vd.data_ptr(); ck.offs[k] = od.data_ptr()
        ck.nkeys = 3
        ids = torch.full((max(int(bag_offs[-1]), 1),), -1, dtype=torch.int32, device="cuda")
        bo = _dev(bag_offs, torch.int32)
        call("wd_cross_hash", ck, B, 0xDECAFCAFFE, nb, ptr(bo), S, slot, ptr(ids),
             torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(ids.cpu().numpy()[: len(exp_ids)].astype(np.int64), exp_ids)


@pytest.mark.parametrize("dim,records", [(4, False), (4, True), (8, False), (8, True), (16, False), (16, True), (32, False),
                                         (64, False), (6, False)])
def test_embag_fwd_mean_matches_oracle(dim, records):
    """records: the row-record table layout (embedding row + wide line of a fused row in one 32 / 64 / 128-byte record)"""
    from wide_deep_amd import synth
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(n_dense=3, n_sparse=4, buckets=500, dim=dim, hidden=(8,))
    eng = _engine(spec, max_batch=128, row_records=records)
    assert (eng.rec is not None) == records
    rng = np.random.default_rng(dim)
    hb = synth.make_raw_batch(eng.plan, 100, seed=dim, mean_len=3)
    hb["lens"][rng.random(hb["lens"].shape) < 0.2] = 0        # empty bags -> zero vector
    nnz = int(hb["lens"].sum()); hb["raw"] = hb["raw"][:nnz]
    bt = synth.to_device_ids(eng.plan, hb)
    eng.forward(bt)
    torch.cuda.synchronize()
    from tests.helpers import slot_csr, assert_close
    csr = slot_csr(eng.plan, bt.ids.cpu().numpy(), bt.bag_offs.cpu().numpy(), bt.B)
    st = eng.export_state()
    x = eng.towers[0]["act"][: bt.B].cpu()
    for si, s in enumerate(eng.plan.slots):
        tab = st["dnn/input_from_feature_columns/input_layer/%s/embedding_weights" % s.deep_name]
        exp = O.embag_fwd(tab, *csr[s.name], mean=True)
        c0 = eng.plan.out_col[si]
        assert_close(x[:, c0:c0 + dim], exp, RTOL, ATOL, "embag slot %s" % s.name)
    for j in range(3):
        assert torch.equal(x[:, eng.plan.dense_out_col[j]], torch.as_tensor(hb["dense"][:, j]))


def test_gemm_variants_match_torch_fp32():
    from wide_deep_amd.capi import call, ptr
    from tests.helpers import assert_close
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for (M, N, K) in [(100, 64, 48), (257, 130, 77), (64, 1, 64), (1000, 256, 429), (33, 7, 5)]:
        lda, ldb, ldc = K + 3, N, N + 5
        A = torch.randn(M, lda, device="cuda", generator=g); Bm = torch.randn(K, ldb, device="cuda", generator=g)
        bias = torch.randn(N, device="cuda", generator=g)
        C = torch.zeros(M, ldc, device="cuda")
        call("wd_gemm_nn_bias_act", ptr(A), lda, ptr(Bm), ldb, ptr(bias), 1, 1, ptr(C), ldc, M, N, K, st)
        ref = torch.relu(A[:, :K].double() @ Bm[:, :N].double() + bias.double())
        assert_close(C[:, :N], ref, 1e-5, 1e-4, "NN %s" % ((M, N, K),))
        assert float(C[:, N:].abs().max()) == 0.0
        # NT: C[M,K] (+)= dZ[M,N] W[K,N]^T
        dZ = torch.randn(M, N, device="cuda", generator=g)
        C2 = torch.ones(M, K + 1, device="cuda")
        call("wd_gemm_nt", ptr(dZ), N, ptr(Bm), ldb, ptr(C2), K + 1, M, K, N, 1, st)
        ref2 = 1.0 + dZ.double() @ Bm[:, :N].double().t()
        assert_close(C2[:, :K], ref2, 1e-5, 1e-4, "NT acc")
        call("wd_gemm_nt", ptr(dZ), N, ptr(Bm), ldb, ptr(C2), K + 1, M, K, N, 0, st)
        assert_close(C2[:, :K], ref2 - 1.0, 1e-5, 1e-4, "NT")
        # TN split-K with ones row: G[K+1,N] = [A|1]^T dZ
        for ns in (1, 3):
            Gp = torch.zeros(ns, K + 1, N, device="cuda")
            call("wd_gemm_tn_splitk", ptr(A), lda, ptr(dZ), N, ptr(Gp), K, N, M, ns, 1, st)
            G = Gp.double().sum(0)
            assert_close(G[:K], A[:, :K].double().t() @ dZ.double(), 1e-5, 2e-4, "TN")
            assert_close(G[K], dZ.double().sum(0), 1e-5, 2e-4, "TN ones row")


def test_grouped_weight_gradient_products():
    _tn_group_check()


def _tn_group_check():
    """wd_gemm_tn_splitk_group: even and odd row strides, odd widths, ragged last tiles, batches that are not a multiple of 16 or of
    2, many splits -- against fp64 torch, with column-sum jobs in the same launch."""
    import ctypes
    from wide_deep_amd import capi
    from wide_deep_amd.capi import call, ptr
    from tests.helpers import assert_close
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    cases = [   # (batch, [(K_l, N_l, lda, nsplit)])
        (8192, [(429, 256, 896, 13), (256, 128, 896, 13), (128, 64, 896, 13)]),
        (1000, [(77, 96, 80, 3), (64, 32, 64, 5)]),
        (999, [(45, 34, 46, 2)]),              # odd batch, odd width (reads column 45 of its last pair), N not a tile multiple
        (37, [(64, 64, 64, 4)]),               # two full sets + a masked one; splits without a set
        (400, [(130, 66, 131, 4)]),            # odd stride: the LDS-tiled kernel
    ]
    for B, layers in cases:
        jobs = (capi.WdTnJob * (len(layers) + 1))()
        keep, exp = [], []
        cs_src = torch.randn(B, 70, device="cuda", generator=g)
        cs_out = torch.zeros(70, device="cuda")
        jobs[0].A, jobs[0].lda, jobs[0].B, jobs[0].Cpart, jobs[0].N, jobs[0].K = ptr(cs_src), 70, None, ptr(cs_out), 70, B
        for j, (K, N, lda, ns) in enumerate(layers):
            A = torch.randn(B, lda, device="cuda", generator=g)
            dZ = torch.randn(B, N, device="cuda", generator=g)
            Gp = torch.full((ns, K, N), float("nan"), device="cuda")
            q = jobs[j + 1]
            q.A, q.lda, q.B, q.ldb, q.Cpart, q.M, q.N, q.K, q.nsplit, q.append_ones = ptr(A), lda, ptr(dZ), N, ptr(Gp), K, N, B, ns, 0
            keep.append((A, dZ, Gp))
            exp.append(A[:, :K].double().t() @ dZ.double())
        call("wd_gemm_tn_splitk_group", jobs, len(layers) + 1, st)
        torch.cuda.synchronize()
        for (A, dZ, Gp), e, (K, N, lda, ns) in zip(keep, exp, layers):
            assert not bool(torch.isnan(Gp).any()), "every partial of every split is written (B %d, layer %s)" % (B, (K, N))
            assert_close(Gp.double().sum(0), e, 1e-5, 3e-4 * (B / 1000.0) ** 0.5, "TN group B %d layer %s" % (B, (K, N, lda, ns)))
        assert_close(cs_out, cs_src.double().sum(0), 1e-5, 1e-4, "column-sum job beside the products")


def test_a_identity_asymmetric_b_detects_transposes():
    from wide_deep_amd.capi import call, ptr
    st = torch.cuda.current_stream().cuda_stream
    n = 64
    A = torch.eye(n, device="cuda")
    Bm = (torch.arange(n, device="cuda").float()[:, None] * 100 + torch.arange(n, device="cuda").float()[None, :]).contiguous()
    C = torch.zeros(n, n, device="cuda")
    call("wd_gemm_nn_bias_act", ptr(A), n, ptr(Bm), n, None, 0, 0, ptr(C), n, n, n, n, st)
    assert torch.equal(C, Bm)


def test_sort_and_sparse_updates_match_oracle():
    from wide_deep_amd import synth
    from wide_deep_amd.plan import criteo_spec
    from tests.helpers import slot_csr, assert_close
    spec = criteo_spec(n_dense=0, n_sparse=3, buckets=50, dim=16, hidden=(8,))   # tiny tables -> many duplicates
    eng = _engine(spec, max_batch=256, row_records=False)     # the per-op entry points work on separate tables
    hb = synth.make_raw_batch(eng.plan, 200, seed=9, mean_len=4)
    bt = synth.to_device_ids(eng.plan, hb)
    st0 = eng.export_state()
    B, ld = bt.B, eng.towers[0]["layout"].ld
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    dx = torch.randn(B, ld, device="cuda", generator=g)
    eng.towers[0]["dact"][:B] = dx
    eng.dlogit[:B] = torch.randn(B, device="cuda", generator=g)
    s = torch.cuda.current_stream().cuda_stream
    eng.sort_occurrences(bt, s)
    torch.cuda.synchronize()
    ks = eng.keys_sorted[: bt.nnz].cpu().numpy().view(np.uint32)
    assert np.all(np.diff(ks.astype(np.int64)) >= 0)
    from wide_deep_amd.capi import call, ptr
    call("wd_embag_bwd_adagrad", ptr(eng.emb), ptr(eng.emb_acc), ptr(eng.slots_dev), eng.plan.S, 16, ptr(eng.keys_sorted),
         ptr(eng.vals_sorted), bt.nnz, ptr(bt.bag_offs), ptr(eng.towers[0]["dact"]), ld, 0.05, s)
    call("wd_wide_bwd_ftrl", ptr(eng.wide), ptr(eng.slots_dev), eng.plan.S, ptr(eng.keys_sorted), ptr(eng.vals_sorted),
         bt.nnz, ptr(eng.dlogit), 0.1, 0.5, 1.0, s)
    call("wd_bias_ftrl", ptr(eng.bias), ptr(eng.dlogit), B, 0.1, 0.5, 1.0, s)
    torch.cuda.synchronize()
    st1 = eng.export_state()
    csr = slot_csr(eng.plan, bt.ids.cpu().numpy(), bt.bag_offs.cpu().numpy(), B)
    dxc, dl = dx.cpu(), eng.dlogit[:B].cpu()
    for si, sl in enumerate(eng.plan.slots):
        nm = "dnn/input_from_feature_columns/input_layer/%s/embedding_weights" % sl.deep_name
        tab, acc = st0[nm].clone(), st0[nm + "/Adagrad"].clone()
        c0 = eng.plan.out_col[si]
        uniq, rg = O.embag_row_grads(16, *csr[sl.name], dxc[:, c0:c0 + 16].contiguous(), mean=True)
        O.adagrad_rows(tab, acc, uniq, rg, 0.05)
        assert_close(st1[nm], tab, RTOL, ATOL, "emb " + sl.name)
        assert_close(st1[nm + "/Adagrad"], acc, RTOL, ATOL, "emb acc " + sl.name)
        wn = "linear/linear_model/%s/weights" % sl.name
        w, z, n = st0[wn].clone(), st0[wn + "/Ftrl_1"].clone(), st0[wn + "/Ftrl"].clone()
        uniq, rg = O.embag_row_grads(1, *csr[sl.name], dl.reshape(-1, 1).contiguous(), mean=False)
        O.ftrl_rows(w, z, n, uniq, rg, 0.1, 0.5, 1.0)
        assert_close(st1[wn], w, RTOL, ATOL, "wide w " + sl.name)
        assert_close(st1[wn + "/Ftrl_1"], z, RTOL, ATOL, "wide z " + sl.name)
        assert_close(st1[wn + "/Ftrl"], n, RTOL, ATOL, "wide n " + sl.name)
    b = "linear/linear_model/bias_weights"
    w, z, n = st0[b].clone(), st0[b + "/Ftrl_1"].clone(), st0[b + "/Ftrl"].clone()
    O.ftrl_dense(w, z, n, dl.sum().reshape(1), 0.1, 0.5, 1.0)
    assert_close(st1[b], w, 1e-4, 1e-6, "bias")


def test_bce_head_matches_oracle():
    from wide_deep_amd.capi import call, ptr
    from tests.helpers import assert_close
    n = 1000
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    a = torch.randn(n, device="cuda", generator=g) * 8; b = torch.randn(n, device="cuda", generator=g)
    y = (torch.rand(n, device="cuda", generator=g) < 0.3).float(); w = torch.rand(n, device="cuda", generator=g)
    logit = torch.zeros(n, device="cuda"); p = torch.zeros(n, device="cuda"); dl = torch.zeros(n, device="cuda"); loss = torch.zeros(1, device="cuda")
    call("wd_bce_sum_fwd_bwd", ptr(a), ptr(b), ptr(y), ptr(w), n, ptr(logit), ptr(p), ptr(dl), ptr(loss), torch.cuda.current_stream().cuda_stream)
    el, edl, ep = O.bce_sum((a + b).cpu(), y.cpu(), w.cpu())
    assert_close(dl, edl, 1e-5, 1e-6, "dlogit"); assert_close(p, ep, 1e-5, 1e-7, "prob")
    assert abs(float(loss) - el) <= 1e-4 * abs(el)


@pytest.mark.parametrize("B,mean_len,dim", [(200, 4, 16), (2000, 4, 16), (8192, 8, 8), (512, 3, 6)])
def test_fused_sparse_backward_matches_sorted_path(B, mean_len, dim):
    """wd_sparse_bwd_fused (bucket + LDS sort + fused updates) against the device-sort path of the same ABI:
    short segments, long (workgroup-reduced) segments, buckets beyond the LDS capacity (sorted in HBM)."""
    from wide_deep_amd import synth
    from wide_deep_amd.plan import criteo_spec
    from tests.helpers import assert_close
    spec = criteo_spec(n_dense=0, n_sparse=3, buckets=50, dim=dim, hidden=(8,))   # tiny tables -> heavy duplicates
    # engine 0: fused kernel on the row-record layout where the width allows it (16, 8); engine 1: separate tables, sorted path
    engs = [_engine(spec, max_batch=B, max_nnz=B * 3 * 16, row_records=None if i == 0 else False) for i in range(2)]
    assert (engs[0].rec is not None) == (dim in (8, 16)) and engs[1].rec is None
    hb = synth.make_raw_batch(engs[0].plan, B, seed=B + dim, mean_len=mean_len)
    bt = synth.to_device_ids(engs[0].plan, hb)
    ld = engs[0].towers[0]["layout"].ld
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    dx = torch.randn(B, ld, device="cuda", generator=g)
    dl = torch.randn(B, device="cuda", generator=g)
    s = torch.cuda.current_stream().cuda_stream
    for e in engs:
        e.towers[0]["dact"][:B] = dx
        e.dlogit[:B] = dl
    engs[0]._sparse_backward(bt, s)
    engs[1]._sparse_backward_unfused(bt, s)
    torch.cuda.synchronize()
    # rows hit ~1000x: the two paths add the same terms in different (each fixed) orders -> absolute slack
    atol = 1e-6 if B * mean_len < 10000 else 3e-4
    a, b = engs[0].export_state(), engs[1].export_state()
    for k in b:
        if k.startswith("dnn/input_from") or k.startswith("linear/"):
            assert_close(a[k], b[k], 2e-5, atol, k)
    # second step on the same engines: the workspaces are reusable
    engs[0]._sparse_backward(bt, s)
    engs[1]._sparse_backward_unfused(bt, s)
    torch.cuda.synchronize()
    a, b = engs[0].export_state(), engs[1].export_state()
    for k in b:
        if k.startswith("dnn/input_from") or k.startswith("linear/"):
            assert_close(a[k], b[k], 5e-5, 2 * atol, k)


@pytest.mark.parametrize("B,mean_len", [(512, 1), (8192, 1), (4096, 3)])
def test_fused_sparse_backward_mixed_vocabularies(B, mean_len):
    """Per-slot bucket geometry: 2-row and 7-row vocabularies (one bucket per row, no sort, thousands of duplicates per
    row -- the reference's conf/feature.yaml shapes at large batch) beside a 100k-row table (row-range buckets, LDS
    sort).  Checked against the device-sort path, and bit-identical when repeated on a fresh engine (deterministic
    summation order)."""
    from wide_deep_amd import synth
    from wide_deep_amd.plan import criteo_spec
    from tests.helpers import assert_close
    spec = criteo_spec(n_dense=0, n_sparse=4, buckets=50, dim=8, hidden=(8,))
    for sl, v in zip(spec.slots, (2, 7, 100000, 300)):
        sl.num_buckets = v
    engs = [_engine(spec, max_batch=B, max_nnz=B * 4 * 16, row_records=None if i != 1 else False) for i in range(3)]
    assert engs[0].rec is not None and engs[1].rec is None
    assert engs[0].bucket_shifts[0] == 0 and engs[0].bucket_shifts[1] == 0 and engs[0].bucket_shifts[2] > 0
    hb = synth.make_raw_batch(engs[0].plan, B, seed=B + 11, mean_len=mean_len)
    bt = synth.to_device_ids(engs[0].plan, hb)
    ld = engs[0].towers[0]["layout"].ld
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    dx = torch.randn(B, ld, device="cuda", generator=g)
    dl = torch.randn(B, device="cuda", generator=g)
    s = torch.cuda.current_stream().cuda_stream
    for e in engs:
        e.towers[0]["dact"][:B] = dx
        e.dlogit[:B] = dl
    engs[0]._sparse_backward(bt, s)
    engs[1]._sparse_backward_unfused(bt, s)
    engs[2]._sparse_backward(bt, s)
    torch.cuda.synchronize()
    a, b, c = (e.export_state() for e in engs)
    for k in b:
        if k.startswith("dnn/input_from") or k.startswith("linear/"):
            assert_close(a[k], b[k], 5e-5, 1e-3, k)     # rows hit thousands of times: summation order differs
            assert torch.equal(torch.as_tensor(a[k]), torch.as_tensor(c[k])), k
