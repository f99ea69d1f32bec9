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
