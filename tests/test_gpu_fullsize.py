"""GPU: the BASELINE configurations AT THEIR OWN SIZE against the CPU oracle, through the path bench.py times.

  C2  batch 8192, 13 dense + 26 slots x 1M buckets, D 16, Dnn [256,128,64]: raw tokens -> wd_hash_bucket (ids bit-exact
      against the oracle's Fingerprint64) -> one-launch tower with the fused input layer -> fused sparse update; three eager
      steps (logits + loss checked after EVERY step) and then the 8-steps-per-hipGraph replay bench.py uses
      (wide_deep_amd/pipeline.StepGraph), uniform and Zipf(1.05) ids;
  C3  the same model over ONE 100M-row table (26 x 3,846,154 rows) resident on one GPU;
  C4  multi-hot (mean 5 ids per slot), ResDnn tower, weight column;
  C5  deep-only DenseDnn [1024,512,256,128], D 64, fp16-operand MFMA tower (fp32 accumulate / embeddings).

The oracle sees the SAMPLED rows of the tables (tests/helpers.CompactOracle: exactly the rows the batches touch, same
arithmetic, same summation orders), every other row of the device tables must stay bit-identical.

Tolerances (north_star "logits within stated fp32 tolerance"): fp32 tower |dlogit| <= 2e-4 + 2e-4 |logit| per step, loss
1e-3 relative of the batch SUM, touched rows / dense parameters 5e-4 relative + 1e-5; fp16-operand tower (C5): logits
3e-2 + 2e-2 |logit| (half rounding of the GEMM operands), parameters 2e-2 relative + 2e-3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

L_RTOL, L_ATOL = 2e-4, 2e-4
P_RTOL, P_ATOL = 5e-4, 1e-5


def _weights(spec, hb):
    if not spec.use_weight_column:
        return None
    return np.where(hb["labels"] > 0, spec.pos_weight, spec.neg_weight).astype(np.float32)


def _hash_and_check(eng, tbs, hbs):
    """tokens -> ids on the device; bit-exact against the oracle's Fingerprint64 % buckets."""
    from oracle import oracle as O
    from wide_deep_amd import synth
    plan = eng.plan
    nb = np.asarray([s.num_buckets for s in plan.slots], dtype=np.uint64)
    out = []
    for tb, hb in zip(tbs, hbs):
        bt = synth.hash_tokens(eng, tb)
        torch.cuda.synchronize()
        ids = bt.ids.cpu().numpy()[: bt.nnz].copy()
        offs = bt.bag_offs.cpu().numpy()
        data, toffs = synth.pack_decimal_tokens(hb["raw"])
        slot_of = np.repeat(np.tile(np.arange(plan.S), hb["B"]), hb["lens"].reshape(-1))
        want = (O.fingerprint64_batch(data, toffs.astype(np.int64)) % nb[slot_of]).astype(np.int64)
        assert np.array_equal(ids.astype(np.int64), want), "hash ids differ from the oracle"
        bt.ids.zero_()           # the step has to produce them itself (hash-in-step)
        out.append((ids, offs, hb["B"]))
    return out


def _fullsize(spec, B, mean_len, dist, n_eager, n_graph, tower_dtype="fp32", tol=None, seed=20260925):
    from tests.helpers import CompactOracle, assert_close
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    from wide_deep_amd.pipeline import StepGraph, step_eager
    l_rtol, l_atol, p_rtol, p_atol = tol or (L_RTOL, L_ATOL, P_RTOL, P_ATOL)
    S = len(spec.slots)
    eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * S * (2 * mean_len + 2), seed=0, tower_dtype=tower_dtype)
    n = n_eager + n_graph
    hbs = [synth.make_raw_batch(eng.plan, B, seed=seed + i, mean_len=mean_len, dist=dist) for i in range(n)]
    tbs = [synth.TokenBatch(eng.plan, hb, weights=_weights(spec, hb)) for hb in hbs]
    dev_ids = _hash_and_check(eng, tbs, hbs)
    co = CompactOracle(eng, dev_ids)
    touched = co.touched_mask()
    emb0 = eng.emb.clone() if eng.emb is not None else None
    wide0 = eng.wide.clone() if eng.wide is not None else None

    def oracle_step(i):
        ids, offs, _ = dev_ids[i]
        return co.ora.train_step(co.batch(ids, offs, B, hbs[i]["dense"], hbs[i]["labels"], _weights(spec, hbs[i])))

    # ---- eager steps: logits + loss after every step --------------------------------------------------------------
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for i in range(n_eager):
            loss = step_eager(eng, tbs[i])
            side.synchronize()
            oloss, ologits = oracle_step(i)
            assert_close(eng.logit[:B], ologits, l_rtol, l_atol, "logits eager step %d" % i)
            assert abs(float(loss) - oloss) <= 1e-3 * max(1.0, abs(oloss)), (i, float(loss), oloss)
    torch.cuda.synchronize()
    # ---- the replay bench.py times: n_graph steps in ONE hipGraph, each on its own batch ----------------------------
    if n_graph:
        g = StepGraph(eng, tbs[n_eager:], stream=side)
        loss = g.replay()
        torch.cuda.synchronize()
        for i in range(n_eager, n):
            oloss, ologits = oracle_step(i)
        assert_close(eng.logit[:B], ologits, l_rtol, l_atol, "logits after the %d-step graph" % n_graph)
        assert abs(float(loss) - oloss) <= 1e-3 * max(1.0, abs(oloss)), (float(loss), oloss)
    # ---- state: touched rows + dense parameters vs the oracle, untouched rows bit-identical ---------------------------
    co.assert_state_matches(p_rtol, p_atol)
    if emb0 is not None:
        (dim, _), = list(eng.plan.emb_groups.items())[:1]
        if len(eng.plan.emb_groups) == 1 and all(s.deep == "embedding" for s in eng.plan.slots):
            assert torch.equal(emb0.view(-1, dim)[: touched.numel()][~touched], eng.emb.view(-1, dim)[: touched.numel()][~touched])
    if wide0 is not None:
        assert torch.equal(wide0[~touched], eng.wide[~touched])
    return eng


def _c2(buckets=1_000_000):
    from wide_deep_amd.plan import criteo_spec
    return criteo_spec(n_dense=13, n_sparse=26, buckets=buckets, dim=16, hidden=(256, 128, 64), mode="simple")


@pytest.mark.parametrize("dist", ["uniform", "zipf"])
def test_c2_full_size_bench_path_matches_oracle(dist):
    eng = _fullsize(_c2(), 8192, 1, dist, n_eager=3, n_graph=8)
    assert eng.chain and eng._fused_input_layer       # the path bench.py times: one-launch tower, input layer fused


def test_c3_100m_row_table_one_gpu_matches_oracle():
    eng = _fullsize(_c2(buckets=3_846_154), 8192, 1, "uniform", n_eager=1, n_graph=2)
    assert eng.plan.total_rows == 26 * 3_846_154


def test_c4_full_size_multi_hot_resnet_weights_matches_oracle():
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=16, hidden=(256, 128, 64), mode="resnet",
                       use_weight_column=True)
    _fullsize(spec, 8192, 5, "zipf", n_eager=2, n_graph=2)


def test_c5_full_size_fp16_tower_matches_oracle():
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=64, hidden=(1024, 512, 256, 128), mode="dense",
                       model_type="deep")
    eng = _fullsize(spec, 8192, 1, "uniform", n_eager=2, n_graph=2, tower_dtype="fp16", tol=(2e-2, 3e-2, 2e-2, 2e-3))
    assert eng.half
