"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/wd_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "wd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from wide_deep_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    # binding table covers the header exactly
    assert sorted(names) == capi.EXPORTED_SYMBOLS


def test_abi_version_and_error_string():
    from wide_deep_amd import capi
    lib = capi.load()
    assert lib.wd_abi_version() == 1
    assert isinstance(lib.wd_last_error(), bytes)


def test_ingest_library_exports_its_header():
    from wide_deep_amd import dataset
    src = open(os.path.join(ROOT, "include", "wd_ingest.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(wd_[a-z0-9_]+)\s*\(", src)))
    assert names == ["wd_crc32c", "wd_tsv_count", "wd_tsv_fill", "wd_tsv_scan", "wd_vocab_lookup"]
    if not os.path.exists(dataset._INGEST_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(dataset._INGEST_PATH)
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    buf = b"a\tb\nc\td\n"
    out = (ctypes.c_int64 * 8)()
    lib.wd_tsv_scan.restype = ctypes.c_int64
    assert lib.wd_tsv_scan(buf, ctypes.c_int64(len(buf)), out, ctypes.c_int64(7)) == 2 and list(out[:3]) == [0, 4, 8]


def test_missing_extension_and_bad_arguments_fail_loudly(monkeypatch):
    """No CPU fallback: without the HIP library the product path raises; with it, every entry point validates its arguments
    BEFORE touching the device and reports through the status code + wd_last_error (checked here without a GPU)."""
    import pytest
    from wide_deep_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", os.path.join(ROOT, "wide_deep_amd", "_lib", "no_such_lib.so"))
    with pytest.raises(capi.WdError, match="There is no CPU fallback"):
        capi.call("wd_fingerprint64", None, None, 1, None, None)
    monkeypatch.undo()
    capi.load()
    with pytest.raises(capi.WdError, match=r"wd_fingerprint64 failed .*null pointer"):
        capi.call("wd_fingerprint64", None, None, 4, None, None)
    with pytest.raises(capi.WdError, match="slot out of range"):
        capi.call("wd_emit_int_slot", 64, 64, 2, 64, 3, 7, 64, None)          # slot 7 of S = 3 (pointers never read)
    with pytest.raises(capi.WdError, match="num_buckets out of range"):
        capi.call("wd_emit_hash_slot", 64, 64, 2, 1 << 31, 64, 3, 0, 64, None)
    o = capi.WdOpt()
    o.kind = 99
    with pytest.raises(capi.WdError, match="bad optimizer"):
        capi.call("wd_opt_dense", 64, 64, 64, 64, 8, ctypes.byref(o), None)
    o.kind = capi.WD_OPT_KINDS["Ftrl"]
    with pytest.raises(capi.WdError, match="needs slot"):
        capi.call("wd_opt_dense", 64, None, None, 64, 8, ctypes.byref(o), None)
    assert b"needs slot" in capi.load().wd_last_error()


def test_a_library_built_from_other_sources_is_refused(monkeypatch):
    """csrc/build.sh embeds a sha256 of the sources (wd_build_stamp); capi.load() recomputes it from the tree and refuses a
    binary that does not match -- the prebuilt .so travels with the tree, a stale one must not load silently."""
    import pytest
    from wide_deep_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    lib.wd_build_stamp.restype = ctypes.c_char_p
    assert lib.wd_build_stamp().decode() == capi.source_stamp()
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.delenv("WD_HIP_LIB", raising=False)
    monkeypatch.setattr(capi, "source_stamp", lambda: "0" * 64)
    with pytest.raises(capi.WdError, match="stale HIP library"):
        capi.load()


def test_tower_lds_sizing_is_host_arithmetic_and_knows_the_launch_limit():
    """wd_tower_chain_lds_bytes / wd_tower_chain_windows_lds_bytes (host-side arithmetic, no launch): BASELINE's tower shapes fit a
    workgroup's dynamic LDS (147 KB: 160 KB minus the kernel's static arrays), shapes that do not are refused with -1 -- the engine
    falls back to per-layer launches on that answer, so a layout between the limit and 160 KB must never be reported as fitting."""
    from wide_deep_amd import capi
    from wide_deep_amd.plan import TowerLayout
    lib = capi.load()
    lim = 147 * 1024
    arr = lambda v: (ctypes.c_int32 * len(v))(*v)
    # C2 / C3: `simple`, x = 26 x 16 + 13 numeric columns (rounded to 32) -> 256-128-64
    b = int(lib.wd_tower_chain_lds_bytes(448, arr([256, 128, 64]), 3, 32))
    assert 0 < b <= lim
    assert int(lib.wd_tower_chain_lds_bytes(448, arr([256, 128, 64]), 3, 16)) == -1           # row tile 32 only
    assert int(lib.wd_tower_chain_lds_bytes(4096, arr([1024, 512]), 2, 32)) == -1             # a row tile that cannot fit
    # configs[3]: `resnet` over the same widths -- every layer reads one contiguous window of the activation row
    for mode in ("resnet", "dense"):
        tl = TowerLayout(448, [256, 128, 64], mode)
        w = capi.WdChainWindows()
        for sg in range(4):
            w.seg_col[sg], w.in_col[sg] = int(tl.seg_start[sg]), int(tl.in_start[sg])
        w.k_logits, w.cols = int(tl.in_K[3]), (int(tl.ld) + 31) // 32 * 32
        bw = int(lib.wd_tower_chain_windows_lds_bytes(ctypes.byref(w), 448, arr([256, 128, 64]), 3))
        assert 0 < bw <= lim, (mode, bw)
    # a concatenating tower whose row does not fit
    tl = TowerLayout(2048, [1024, 512, 256], "resnet")
    w = capi.WdChainWindows()
    for sg in range(4):
        w.seg_col[sg], w.in_col[sg] = int(tl.seg_start[sg]), int(tl.in_start[sg])
    w.k_logits, w.cols = int(tl.in_K[3]), (int(tl.ld) + 31) // 32 * 32
    assert int(lib.wd_tower_chain_windows_lds_bytes(ctypes.byref(w), 2048, arr([1024, 512, 256]), 3)) == -1
