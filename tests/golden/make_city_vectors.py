#!/usr/bin/env python
"""Known answers for Fingerprint64 on 0..32-byte inputs from an INDEPENDENT compiled implementation.

farmhashna::Hash64 (TF's Fingerprint64) and CityHash64 v1.1 share their HashLen0to16 and HashLen17to32
branches (they differ from 33 bytes on).  Abseil ships CityHash64 v1.1 (absl/hash/internal/city.cc) and pyarrow's
libarrow_compute.so exports it, so the 17..32-byte branch -- for which no TensorFlow known answer is obtainable offline --
can be pinned against code that is neither the oracle nor written in this repo.  The 0..16-byte vectors double the TF KATs.
Run where pyarrow is installed:  python tests/golden/make_city_vectors.py  ->  tests/golden/kat_city_le32.json
(33..64 and >64 bytes stay unpinned: asserted below that CityHash64 differs there, i.e. it is no evidence for them.)"""
import ctypes, glob, json, os, random, subprocess

import pyarrow
import pyarrow.compute  # noqa: F401  (loads the library's dependencies)

lib = sorted(glob.glob(os.path.join(os.path.dirname(pyarrow.__file__), "libarrow_compute.so*")))[0]
syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout.split()
name = [s for s in syms if "hash_internal" in s and s.endswith("10CityHash64EPKcm")][0]
city = getattr(ctypes.CDLL(lib), name)
city.restype, city.argtypes = ctypes.c_uint64, [ctypes.c_char_p, ctypes.c_size_t]

rnd = random.Random(20260926)
vec = []
for n in range(0, 33):
    for k in range(12):
        s = bytes(rnd.getrandbits(8) for _ in range(n)) if k else bytes([0x30 + (i % 10) for i in range(n)])
        vec.append([s.hex(), city(s, n)])
# real tokens of the bundled data that are longer than 16 bytes (device_model style strings)
for t in [b"Mozilla/5.0 (Linux; U;", b"com.tencent.mobileqq.x", b"samsung-sm-g9250-android", b"xiaomi_redmi_note_4x_pro_32gb"[:32]]:
    vec.append([t.hex(), city(t, len(t))])

if __name__ == "__main__":
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from oracle import oracle as O
    s40 = bytes(range(40))
    s100 = bytes(range(100))
    assert city(s40, 40) != O.fingerprint64(s40) and city(s100, 100) != O.fingerprint64(s100)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_city_le32.json")
    json.dump({"_source": "absl::hash_internal::CityHash64 (CityHash v1.1) exported by pyarrow %s libarrow_compute; "
                          "equal to farmhashna::Hash64 for len <= 32; made by tests/golden/make_city_vectors.py" % pyarrow.__version__,
               "vectors": vec}, open(out, "w"), indent=0)
    print(len(vec), "vectors ->", out)
