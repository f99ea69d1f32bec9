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
    assert names == ["wd_tsv_count", "wd_tsv_fill", "wd_tsv_scan", "wd_vocab_lookup"]
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
