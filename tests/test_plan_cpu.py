"""CPU: host-side planning logic of the engine (no GPU): activation-buffer layouts of the connected modes, the row-range
bucket geometry of the fused sparse update, optimizer slot naming."""
import pytest

from wide_deep_amd.plan import (FeaturePlan, OPT_SLOT_NAMES, TowerLayout, adam_pow_names, bucket_geometry, criteo_spec,
                                opt_params, opt_slot_init, opt_slot_names)


@pytest.mark.parametrize("mode", ["simple", "dense", "resnet", "last_dense", "first_dense"])
@pytest.mark.parametrize("hidden", [(8,), (16, 8), (32, 16, 8), (16, 8, 8, 4, 12)])
def test_every_layer_reads_one_contiguous_window(mode, hidden):
    """python/lib/dnn.py:92-193: the concat a layer consumes is ONE column window of the activation buffer (free concat),
    windows list exactly the segments the reference concatenates, and segments never overlap."""
    deep = 20
    tl = TowerLayout(deep, hidden, mode)
    L = len(hidden)
    spans = sorted((tl.seg_start[j], tl.seg_start[j] + tl.seg_width[j]) for j in range(len(tl.seg_width)))
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0, "segments overlap"
    assert spans[-1][1] <= tl.ld
    for l in range(L + 1):
        segs = tl.in_segs[l]
        assert sum(tl.seg_width[j] for j in segs) == tl.in_K[l]                       # contiguous: no holes
        canon = [tl.canon(j) for j in segs]
        if mode == "simple" or (mode == "last_dense" and l < L):
            want = [l]
        elif mode == "first_dense":
            want = [0] if l == 0 else [l, 0]                                             # [h_{l-1} | x]
        elif mode in ("dense", "last_dense"):
            want = list(range(l + 1))                                                    # [x | h_0 | ... | h_{l-1}]
        else:
            want = list(range(l, -1, -1))                                                # resnet: newest first
        assert canon == want, (mode, l, canon, want)
        cols = tl.window_cols(l)
        assert len(cols) == tl.in_K[l] and all(seg >= 0 for seg, _ in cols)
    if mode == "first_dense":
        assert len(tl.x_copies) == max(0, (L + 1) // 2 - 1)                             # one copy of x serves two consumers


def test_first_dense_kernel_rows_follow_the_reference_concat_order():
    spec = criteo_spec(n_dense=3, n_sparse=2, buckets=10, dim=8, hidden=(8, 4, 4), mode="first_dense")
    plan = FeaturePlan(spec)
    tl = plan.towers[0]
    for l in range(1, 4):
        rows = plan.tf_rows_of_layer(0, l)
        K_tf = tl.seg_width[l] + plan.tf_deep_dim                     # [h_{l-1} | x] in TF's order
        assert len(rows) == K_tf and len(set(rows.tolist())) == K_tf
        h0 = tl.seg_start[l] - tl.in_start[l]
        assert rows[: tl.seg_width[l]].tolist() == list(range(h0, h0 + tl.seg_width[l]))


@pytest.mark.parametrize("vocab,occ", [([1_000_000] * 26, 8192), ([2, 7, 55, 1000, 100_000, 12_000_000], 8192),
                                       ([50] * 3, 200), ([3_846_154] * 26, 8192 * 5), ([1], 10)])
def test_bucket_geometry_invariants(vocab, occ):
    nb_max = 8192
    shifts, bases, total = bucket_geometry(vocab, occ, nb_max)
    assert total <= nb_max and len(shifts) == len(bases) == len(vocab)
    run = 0
    for v, sh, base in zip(vocab, shifts, bases):
        assert base == run                                            # buckets of a slot are contiguous, slots in order
        nb = (v + (1 << sh) - 1) >> sh
        run += nb
        if sh == 0:
            assert nb == v                                            # one bucket per row (no sort in the update kernel)
        else:
            assert ((v - 1) >> sh) == nb - 1                          # last row lands in the last bucket
    assert run == total
    small = [i for i, v in enumerate(vocab) if v <= occ / 64.0]
    assert all(shifts[i] == 0 for i in small) or total == nb_max      # tiny vocabularies: per-row buckets


def test_optimizer_slot_tables_are_consistent():
    opts = [("SGD", 0.1), ("Adagrad", 0.05, 0.1), ("Ftrl", 0.1, 0.5, 1.0, 0.2), ("RMSProp", 0.01, 0.9, 0.1, 1e-10),
            ("Adam", 0.001, 0.9, 0.999, 1e-8)]
    for o in opts:
        a, b = opt_slot_init(o)
        sa, sb = OPT_SLOT_NAMES[o[0]]
        assert (a is None) == (sa is None) and (b is None) == (sb is None)
        assert len(opt_params(o)) == 4
    assert opt_slot_init(opts[1]) == (None, 0.1) and opt_slot_init(opts[2]) == (0.0, 0.2)
    assert opt_slot_init(opts[3]) == (1.0, 0.0)                       # rms slot starts at ones
    assert opt_params(opts[2]) == (0.5, 1.0, -0.5, 0.0) and opt_params(opts[4]) == (0.9, 0.999, 1e-8, 0.0)
    assert opt_params(("Ftrl", 0.1, 0.5, 1.0, 0.2, -0.7)) == (0.5, 1.0, -0.7, 0.0)     # p2 = learning_rate_power
    assert opt_params(("Ftrl", 0.1, 0.5, 1.0, 0.2, -0.5, 0.25)) == (0.5, 1.0, -0.5, 0.25)   # p3 = l2_shrinkage_regularization_strength
    # RMSPropOptimizer._create_slots: rms, [mg,] momentum -> /RMSProp, [/RMSProp_1,] /RMSProp_1 or _2
    assert opt_slot_names(opts[3]) == ("/RMSProp", "/RMSProp_1", None)
    assert opt_slot_names(opts[3] + (True,)) == ("/RMSProp", "/RMSProp_2", "/RMSProp_1")
    assert opt_slot_names(opts[1]) == (None, "/Adagrad", None)
    assert adam_pow_names(opts[4], opts[4], True, True) == {"dnn": ("beta1_power", "beta2_power"),
                                                            "linear": ("beta1_power_1", "beta2_power_1")}
    assert adam_pow_names(opts[4], opts[4], False, True) == {"linear": ("beta1_power", "beta2_power")}
    assert adam_pow_names(opts[1], opts[2], True, True) == {}


def test_deep_input_is_padded_for_aligned_windows():
    """x is padded to a multiple of 4 (64 beyond 128 columns); pad columns map to no TF row and stay zero."""
    for n_dense, n_sparse, dim in [(3, 5, 16), (13, 26, 16), (1, 1, 4), (0, 3, 8)]:
        plan = FeaturePlan(criteo_spec(n_dense=n_dense, n_sparse=n_sparse, buckets=10, dim=dim, hidden=(8,)))
        c = n_dense + n_sparse * dim
        assert plan.tf_deep_dim == c and plan.deep_dim >= c and plan.deep_dim % 4 == 0
        assert plan.deep_dim % 64 == 0 or c <= 128
        assert len(set(plan.tf_input_perm.tolist())) == c and int(plan.tf_input_perm.max()) < plan.deep_dim


@pytest.mark.parametrize("conn,hidden", [(["0-1", "0-3", "1-2"], (16, 8, 8)), ("0-1,1-2,2-3", (16, 8, 4)), (["0-2"], (8, 8)),
                                         ([(0, 3), (1, 3), (2, 3)], (8, 4, 12)), (["1-2", "1-2"], (8, 8))])
def test_connection_list_windows(conn, hidden):
    """python/lib/dnn.py:195-224 (as documented at :65-66): layer j reads [net_i for every i -> j, ascending | h_{j-1}], each
    net_i itself such a concat; every window is contiguous, ends in its own segment, and repeats the others as copies."""
    from wide_deep_amd.plan import parse_connections
    deep = 20
    tl = TowerLayout(deep, hidden, conn)
    L = len(hidden)
    assert tl.mode == "list" and tl.connections == parse_connections(conn, L)
    into = {}
    for i, j in tl.connections:
        into.setdefault(j, []).append(i)
    flat = [[0]]
    for l in range(1, L + 1):
        flat.append([s for i in sorted(into.get(l, [])) for s in flat[i]] + [l])
    spans = sorted((tl.seg_start[j], tl.seg_start[j] + tl.seg_width[j]) for j in range(len(tl.seg_width)))
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0, "segments overlap"
    for l in range(L + 1):
        segs = tl.in_segs[l]
        assert [tl.canon(j) for j in segs] == flat[l] and segs[-1] == l
        assert all(j > L for j in segs[:-1])                                   # everything but the tail is a copy
        assert sum(tl.seg_width[j] for j in segs) == tl.in_K[l] and tl.in_start[l] % 4 == 0
        assert all(seg >= 0 for seg, _ in tl.window_cols(l))
    assert sorted(tl.x_copies) == sorted(cs for cs, src in tl.copies.items() if src == 0)
    for l in range(1, L + 1):
        assert all(tl.seg_width[cs] == hidden[l - 1] for cs in tl.copies_of(l))


def test_connection_list_errors_and_chain_list_is_dense():
    from wide_deep_amd.build_estimator import tower_specs
    from wide_deep_amd.plan import is_connection_list
    for bad in (["1-0"], ["0-4"], ["2-2"], ["a-b"], ["0-1-2"]):
        with pytest.raises(ValueError):
            TowerLayout(16, (8, 8, 8), bad)
    with pytest.raises(ValueError):
        TowerLayout(16, (8, 8), "densenet")
    assert not is_connection_list("simple") and is_connection_list("0-1, 1-2") and is_connection_list(["0-1"])
    # python/lib/dnn.py:253-258: one mode for every DNN when it is a name or a list of 'i-j' items, else one per DNN
    t = tower_specs([[8, 8], [8, 4, 4]], ["0-1", "0-2"])
    assert [x.mode for x in t] == [((0, 1), (0, 2)), ((0, 1), (0, 2))]
    t = tower_specs([[8, 8], [8, 4, 4]], ["simple", ["0-1", "1-3"]])
    assert [x.mode for x in t] == ["simple", ((0, 1), (1, 3))]
    assert tower_specs([8, 4], "0-1,0-2")[0].mode == ((0, 1), (0, 2)) and tower_specs([8, 4], "resnet")[0].mode == "resnet"
    # kernel rows: the chain list concatenates exactly what `dense` does, in the same order
    a = FeaturePlan(criteo_spec(n_dense=3, n_sparse=2, buckets=10, dim=8, hidden=(8, 4, 4), mode="dense"))
    b = FeaturePlan(criteo_spec(n_dense=3, n_sparse=2, buckets=10, dim=8, hidden=(8, 4, 4), mode=((0, 1), (1, 2), (2, 3))))
    for l in range(4):
        ra, rb = a.tf_rows_of_layer(0, l), b.tf_rows_of_layer(0, l)
        ca, cb = a.towers[0].window_cols(l), b.towers[0].window_cols(l)
        assert [(a.towers[0].canon(ca[r][0]), ca[r][1]) for r in ra] == [(b.towers[0].canon(cb[r][0]), cb[r][1]) for r in rb]


def test_tower_specs_refuse_mode_lists_that_do_not_match_the_towers():
    """build_estimator.tower_specs: a silent zip() used to drop towers (or modes) -- an empty dnn_connected_mode gave a deep model
    without a tower; a connection list with items longer than three characters ('0-10') was taken for one mode per tower."""
    import pytest
    from wide_deep_amd.build_estimator import tower_specs
    assert [t.mode for t in tower_specs([[8, 4], [6]], ["simple", "resnet"])] == ["simple", "resnet"]
    assert len(tower_specs([[8, 4], [6]], "dense")) == 2
    with pytest.raises(ValueError):
        tower_specs([8, 4], [])
    with pytest.raises(ValueError):
        tower_specs([[8, 4], [6]], ["simple"])
    hidden = [4] * 12
    t, = tower_specs(hidden, ["0-10", "1-3"])           # ONE tower: a list of 'i-j' items is its connection list whatever their length
    assert (0, 10) in t.mode and (1, 3) in t.mode


def test_decayed_lr_starts_from_the_initial_rate_it_is_given():
    from wide_deep_amd.plan import criteo_spec
    spec = criteo_spec(n_dense=1, n_sparse=1, buckets=10, dim=4, hidden=(4,))
    spec.lr_decay = {"dnn": (0.5, 10.0)}
    spec.dnn_opt = ("Adagrad", 0.0125, 0.1)      # what an engine's own copy holds after a few decayed steps
    assert abs(spec.decayed_lr("dnn", 20, lr0=0.05) - 0.0125) < 1e-12 and spec.decayed_lr("linear", 20, lr0=0.1) == 0.1


def test_bench_gather_contract_bytes_by_column_width():
    """bench.py's SURVEY 8(d) gather contract for models with several embedding widths (configs[3] with its crossed columns): every
    column priced at its own row width and its own ids -- by hand on a 2-example, 3-column batch."""
    import importlib.util
    import os
    import types
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("wd_bench_module", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    B = 2
    lens = torch.tensor([[1, 2, 5], [3, 0, 7]])
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), lens.reshape(-1).cumsum(0)]).to(torch.int32)
    bt = types.SimpleNamespace(B=B, nnz=int(lens.sum()), bag_offs=offs)
    mixed = types.SimpleNamespace(S=3, emb_groups={16: [0, 1], 4: [2]})
    offsets = (B * 3 + 1) * 4
    assert b.gather_alg_bytes(mixed, bt, 16, 2, whole_layer="all") == (6 * 64 + 12 * 16) + 18 * 4 + offsets + (B * 2 * 64 + B * 16)
    assert b.gather_alg_bytes(mixed, bt, 16, 2, whole_layer=True) == 6 * 64 + 6 * 4 + offsets + B * 2 * 64     # the width-16 launch alone
    one = types.SimpleNamespace(S=3, emb_groups={16: [0, 1, 2]})
    for mode in (False, True, "all"):      # one width: the plain formula
        assert b.gather_alg_bytes(one, bt, 16, 3, whole_layer=mode) == 18 * 64 + 18 * 4 + offsets + B * 3 * 64
