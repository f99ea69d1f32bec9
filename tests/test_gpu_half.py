"""GPU tests of the fp16-input MFMA tower (csrc/mlp_half.hip) against torch references that see the SAME half-rounded
operands (fp32 accumulate), so the tolerance only covers summation order and the final rounding to half."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _st():
    return torch.cuda.current_stream().cuda_stream


def _close(a, b, rtol, atol, what):
    a, b = a.double().cpu(), b.double().cpu()
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bool(bad.any()), "%s: %d bad, max err %.3e" % (what, int(bad.sum()), float(err.max()))


@pytest.mark.parametrize("M,N,K,pad", [(256, 128, 64, 0), (1000, 300, 200, 0), (130, 70, 33, 3), (8192, 1024, 448, 0), (64, 8, 16, 0)])
def test_hgemm_nn_nt_tn_match_torch(M, N, K, pad):
    from wide_deep_amd.capi import call, ptr
    g = torch.Generator(device="cuda"); g.manual_seed(M + N + K)
    lda = K + pad + (8 - (K + pad) % 8) % 8 if pad == 0 else K + pad          # pad != 0: pitch NOT a multiple of 8 (scalar path)
    ldw = lda
    A = torch.zeros(M, lda, dtype=torch.float16, device="cuda"); A[:, :K] = torch.randn(M, K, device="cuda", generator=g).half()
    WT = torch.zeros(N, ldw, dtype=torch.float16, device="cuda"); WT[:, :K] = (torch.randn(N, K, device="cuda", generator=g) * 0.1).half()
    bias = torch.randn(2, N, device="cuda", generator=g)
    ldc, ldct = N + 8, ((M + 63) // 64) * 64
    C = torch.zeros(M, ldc, dtype=torch.float16, device="cuda"); CT = torch.zeros(N, ldct, dtype=torch.float16, device="cuda")
    call("wd_hgemm_nn", ptr(A), lda, ptr(WT), ldw, ptr(bias), 2, 1, ptr(C), ldc, ptr(CT), ldct, M, N, K, _st())
    ref = torch.relu(A[:, :K].float() @ WT[:, :K].float().t() + bias.sum(0))
    _close(C[:, :N], ref, 2e-3, 2e-3, "NN")
    assert torch.equal(CT[:, :M], C[:, :N].t().contiguous()), "transposed copy differs"
    assert float(C[:, N:].abs().max()) == 0.0 and (ldct == M or float(CT[:, M:].abs().max()) == 0.0)

    # NT: X[M, K] = dZ[M, N] W[K, N]^T   (W rows = output columns k, reduction n contiguous)
    lddz = N + (8 - N % 8) % 8 + pad
    dZ = torch.zeros(M, lddz, dtype=torch.float16, device="cuda"); dZ[:, :N] = torch.randn(M, N, device="cuda", generator=g).half()
    W = torch.zeros(K, lddz, dtype=torch.float16, device="cuda"); W[:, :N] = WT[:, :K].t()
    refx = dZ[:, :N].float() @ W[:, :N].float().t()
    C32 = torch.ones(M, K + 5, device="cuda")
    call("wd_hgemm_nt", ptr(dZ), lddz, ptr(W), lddz, M, K, N, ptr(C32), K + 5, 1, None, 0, None, 0, None, 0, 0, _st())
    _close(C32[:, :K], refx + 1.0, 1e-3, 1e-3, "NT accumulate")
    call("wd_hgemm_nt", ptr(dZ), lddz, ptr(W), lddz, M, K, N, ptr(C32), K + 5, 0, None, 0, None, 0, None, 0, 0, _st())
    _close(C32[:, :K], refx, 1e-3, 1e-3, "NT store")
    assert float((C32[:, K:] - 1.0).abs().max()) == 0.0
    act_src = torch.randn(M, K, device="cuda", generator=g).half()
    Dh = torch.zeros(M, K, dtype=torch.float16, device="cuda"); DT = torch.zeros(K, ldct, dtype=torch.float16, device="cuda")
    call("wd_hgemm_nt", ptr(dZ), lddz, ptr(W), lddz, M, K, N, None, 0, 0, ptr(Dh), K, ptr(DT), ldct, ptr(act_src), K, 1, _st())
    _close(Dh, refx * (act_src.float() > 0).float(), 2e-3, 2e-3, "NT fused relu'")
    assert torch.equal(DT[:, :M], Dh.t().contiguous())

    # TN split-K with the ones row: G[K+1, N] = [A | 1]^T dZ, operands given TRANSPOSED (batch contiguous)
    AT = torch.zeros(K, ldct, dtype=torch.float16, device="cuda"); AT[:, :M] = A[:, :K].t()
    dZT = torch.zeros(N, ldct, dtype=torch.float16, device="cuda"); dZT[:, :M] = dZ[:, :N].t()
    for ns in (1, 3):
        Gp = torch.zeros(ns, K + 1, N, device="cuda")
        call("wd_hgemm_tn_splitk", ptr(AT), ldct, ptr(dZT), ldct, ptr(Gp), K, N, M, ns, _st())
        G = Gp.double().sum(0)
        _close(G[:K], A[:, :K].double().t() @ dZ[:, :N].double(), 1e-3, 2e-3 * np.sqrt(M), "TN")
        _close(G[K], dZ[:, :N].double().sum(0), 1e-3, 2e-3 * np.sqrt(M), "TN ones row")


def test_cast_transpose_with_activation_derivative():
    from wide_deep_amd.capi import call, ptr
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    R, C = 300, 170
    src = torch.randn(R, C + 6, device="cuda", generator=g)
    a = torch.randn(R, C, device="cuda", generator=g).half()
    dst = torch.zeros(R, C + 2, dtype=torch.float16, device="cuda"); dstT = torch.zeros(C, 320, dtype=torch.float16, device="cuda")
    call("wd_cast_transpose_h", ptr(src), C + 6, R, C, None, 0, 0, ptr(dst), C + 2, ptr(dstT), 320, _st())
    assert torch.equal(dst[:, :C], src[:, :C].half()) and torch.equal(dstT[:, :R], src[:, :C].half().t().contiguous())
    call("wd_cast_transpose_h", ptr(src), C + 6, R, C, ptr(a), C, 1, ptr(dst), C + 2, ptr(dstT), 320, _st())
    exp = (src[:, :C] * (a.float() > 0).float()).half()
    assert torch.equal(dst[:, :C], exp) and torch.equal(dstT[:, :R], exp.t().contiguous())


@pytest.mark.parametrize("mode,model_type,dim,hidden", [("dense", "deep", 64, (128, 64, 32, 16)), ("simple", "wide_deep", 16, (96, 48)),
                                                        ("resnet", "wide_deep", 8, (40, 24))])
def test_fp16_tower_tracks_fp32_tower_and_oracle(mode, model_type, dim, hidden):
    """BASELINE configs[4] shape at test size: deep-only DenseDnn, emb 64, fp16 MFMA tower with fp32 embeddings (and the
    other connection modes).  Same weights and batches through tower_dtype='fp16', 'fp32' and the fp32 CPU oracle."""
    from tests.helpers import oracle_batch, oracle_from_engine
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(n_dense=5, n_sparse=6, buckets=400, dim=dim, hidden=hidden, mode=mode, model_type=model_type)
    B = 200
    e16 = WideDeepEngine(spec, max_batch=B, seed=5, tower_dtype="fp16")
    e32 = WideDeepEngine(spec, max_batch=B, seed=5, tower_dtype="fp32")
    ora = oracle_from_engine(e32)
    first = None
    for step in range(4):
        hb = synth.make_raw_batch(e32.plan, B, seed=100 + step, mean_len=2, pos_rate=0.3)
        bt = synth.to_device_ids(e32.plan, hb)
        l16 = float(e16.train_step(bt)); l32 = float(e32.train_step(bt))
        torch.cuda.synchronize()
        ob = oracle_batch(e32.plan, bt.ids.cpu().numpy(), bt.bag_offs.cpu().numpy(), B, hb["dense"], hb["labels"])
        oloss, ologits = ora.train_step(ob)
        # fp16 operands: ~1e-3 relative per product; logits are O(1) sums of a few hundred of them
        _close(e16.logit[:B], e32.logit[:B], 3e-2, 2e-2, "fp16 vs fp32 logits step %d" % step)
        _close(e16.logit[:B], ologits, 3e-2, 2e-2, "fp16 vs oracle logits step %d" % step)
        assert abs(l16 - oloss) <= 2e-2 * max(1.0, abs(oloss)), (step, l16, l32, oloss)
        first = first if first is not None else l16
    # the fp16 tower trains: the embedding tables moved exactly where the fp32 tower moved them (up to fp16 noise)
    a, b = e16.export_state(), e32.export_state()
    k = [n for n in b if n.endswith("embedding_weights")][0]
    moved = (b[k] - oracle_from_engine(WideDeepEngine(spec, max_batch=B, seed=5)).state[k]).abs().sum()
    # Adagrad normalises tiny gradients to ~lr-sized moves, so an fp16-noise sign flip can show as ~1e-2 on single elements
    assert float(moved) > 0 and float((a[k] - b[k]).abs().max()) < 3e-2 and float((a[k] - b[k]).abs().mean()) < 1e-3
