"""CPU: the HOST half of the featurizer (wide_deep_amd/features.py: bag CSR per (example, slot), the staged arrays, the
gather indices that pad crossed string keys with '', the vocabulary / identity / bucketize paths) on real rows of the
bundled click log, with the four device entry points it drives (wd_fingerprint64, wd_emit_hash_slot, wd_emit_int_slot,
wd_cross_hash) replaced by numpy stand-ins built on the oracle.  Every column's ids must equal oracle/columns.py -- the same
assertion tests/test_gpu_c1.py makes with the real kernels on the GPU."""
import ctypes
import os
import types

import numpy as np
import pytest
import torch

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "c1_rows.tsv")


def _addr(p):
    return p.value if isinstance(p, ctypes.c_void_p) else int(p)


def _view(p, n, dtype):
    dt = np.dtype(dtype)
    if n == 0:
        return np.zeros(0, dt)
    return np.frombuffer((ctypes.c_char * (n * dt.itemsize)).from_address(_addr(p)), dtype=dt)


def _fake_call(name, *a):
    if name == "wd_fingerprint64":
        data_p, offs_p, n, out_p, _ = a
        offs = _view(offs_p, n + 1, np.int32)
        _view(out_p, n, np.uint64)[:] = O.fingerprint64_batch(_view(data_p, int(offs[n]), np.uint8).copy(), offs.copy())
    elif name == "wd_emit_hash_slot":
        fp_p, fo_p, B, nb, bag_p, S, slot, ids_p, _ = a
        fo, bag = _view(fo_p, B + 1, np.int32), _view(bag_p, B * S + 1, np.int32)
        fp, ids = _view(fp_p, int(fo[B]), np.uint64), _view(ids_p, int(bag[-1]), np.int32)
        for b in range(B):
            n, d = fo[b + 1] - fo[b], bag[b * S + slot]
            assert bag[b * S + slot + 1] - d == n
            ids[d: d + n] = fp[fo[b]: fo[b + 1]] % np.uint64(nb)
    elif name == "wd_emit_int_slot":
        v_p, fo_p, B, bag_p, S, slot, ids_p, _ = a
        fo, bag = _view(fo_p, B + 1, np.int32), _view(bag_p, B * S + 1, np.int32)
        v, ids = _view(v_p, int(fo[B]), np.int32), _view(ids_p, int(bag[-1]), np.int32)
        for b in range(B):
            n, d = fo[b + 1] - fo[b], bag[b * S + slot]
            assert bag[b * S + slot + 1] - d == n
            ids[d: d + n] = v[fo[b]: fo[b + 1]]
    elif name == "wd_cross_hash":
        ck, B, key, nb, bag_p, S, slot, ids_p, _ = a
        ck = ck._obj
        cols = []
        for k in range(ck.nkeys):
            offs = _view(ck.offs[k], B + 1, np.int32).copy()
            cols.append((_view(ck.vals[k], int(offs[B]), np.uint64).copy(), offs))
        out, oo = O.cross_hash(cols, nb, hash_key=key)
        bag = _view(bag_p, B * S + 1, np.int32)
        ids = _view(ids_p, int(bag[-1]), np.int32)
        for b in range(B):
            n, d = oo[b + 1] - oo[b], bag[b * S + slot]
            assert bag[b * S + slot + 1] - d == n
            ids[d: d + n] = out[oo[b]: oo[b + 1]]
    else:
        raise AssertionError("unexpected entry point " + name)
    return 0


@pytest.fixture
def host_featurizer(monkeypatch):
    from wide_deep_amd import build_estimator as BE, features as F, plan as PL
    from wide_deep_amd.read_conf import Config
    monkeypatch.setattr(F, "call", _fake_call)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: types.SimpleNamespace(cuda_stream=0))

    def make(model_type, padding, max_batch=512):
        spec = BE.build_model_spec(Config(), model_type)
        eng = types.SimpleNamespace(plan=PL.FeaturePlan(spec), spec=spec, device=torch.device("cpu"), max_batch=max_batch,
                                    max_nnz=max_batch * len(spec.slots) * 16)
        return eng, F.Featurizer(eng, cross_padding=padding, mode="host")
    return make


@pytest.mark.parametrize("padding", ["tf_dense", "ragged"])
def test_host_half_of_the_featurizer_on_real_rows(tmp_path, host_featurizer, padding):
    from oracle import columns as OC
    from tests.helpers import slot_csr
    from tests.test_conf_dataset import _with_na_rows
    from wide_deep_amd import dataset as DS, features as F
    from wide_deep_amd.read_conf import Config, conf_dir
    lines = _with_na_rows(open(FIXTURE, "rb").read().splitlines(), Config().read_schema())
    path = tmp_path / "rows.tsv"
    path.write_bytes(b"\n".join(lines) + b"\n")
    eng, fz = host_featurizer("wide_deep", padding)
    oc = OC.Columns(conf_dir())
    k = 0
    for raw in DS.input_fn(str(path), None, "eval", 200):
        bt = fz.to_device(raw)
        assert bt.B == raw.B and bt.nnz == int(bt.bag_offs[-1]) and bt.labels is not None
        got = slot_csr(eng.plan, bt.ids.numpy(), bt.bag_offs.numpy(), raw.B)
        exp = oc.transform(oc.parse(lines[k:k + raw.B]), cross_padding=padding)
        k += raw.B
        assert set(got) == set(exp["ids"])
        for name, (eids, eoffs) in exp["ids"].items():
            gids, goffs = got[name]
            assert np.array_equal(goffs, eoffs), name
            assert np.array_equal(gids, np.asarray(eids, dtype=np.int64)), name
        for j, d in enumerate(eng.plan.dense_cols):   # raw value travels; wd_dense_fwd applies (kind, p0, p1) on the device
            x = bt.dense[:, j].numpy()
            assert np.array_equal(x, raw.floats[d.feature])
            kind = {0: None, 1: "min_max", 2: "standard", 3: "log"}[d.kind]
            np.testing.assert_allclose(F._normalize(x, (kind, d.p0, d.p1) if kind else None),
                                       np.asarray(exp["dense"][d.name], np.float32), rtol=1e-6)
        assert bt.one_hot == bool((np.diff(bt.bag_offs.numpy()) == 1).all())
    assert k == len(lines)


def test_capacity_check_and_single_copy_staging(tmp_path, host_featurizer):
    from wide_deep_amd import dataset as DS, features as F
    lines = open(FIXTURE, "rb").read().splitlines()[:64]
    path = tmp_path / "rows.tsv"
    path.write_bytes(b"\n".join(lines) + b"\n")
    eng, fz = host_featurizer("wide", "tf_dense", max_batch=32)
    raw = next(iter(DS.input_fn(str(path), None, "eval", 64)))
    with pytest.raises(ValueError, match="exceeds engine capacity"):
        fz.to_device(raw)
    # one staged array per distinct host array: an array staged twice keeps one handle, the single-valued CSR is shared
    st = F._Stage()
    a = np.arange(5, dtype=np.int32)
    assert st.add(a, np.int32) == st.add(a, np.int32) and st.add(a, np.int64) != st.add(a, np.int32)
    h = st.add(np.zeros(0, np.float32), np.float32)
    st.upload(torch.device("cpu"))
    assert st.tensor(h).numel() == 0 and st.tensor(0).tolist() == list(range(5)) and st.tensor(1).dtype == torch.int64
    assert all(off % 16 == 0 for off, _ in st.items)
