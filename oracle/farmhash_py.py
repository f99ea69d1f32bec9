"""oracle/farmhash_py.py -- TEST INFRASTRUCTURE ONLY.

Second, independent pure-Python transcription of TF ``Fingerprint64``
(= FarmHash ``farmhashna::Hash64``) and ``FingerprintCat64``, written from the
prose description in SURVEY.md Appendix A.1/A.3 rather than from
``wd_oracle.c``.  It cross-checks the C restatement on every length branch.
Both are pinned by published known answers of this function (tests/helpers.py:
Guava FarmHashFingerprint64Test incl. the 3200-message chain over lengths
0..3199, BigQuery FARM_FINGERPRINT doc examples), by the upstream-TF vectors in
tests/golden/kat_hash.json and by a compiled CityHash64 for <= 32 bytes
(tests/golden/kat_city_le32.json).

Reference call sites that select this arithmetic:
python/lib/build_estimator.py:86-88 (hash buckets) and :138-155 (crosses).
"""
M64 = (1 << 64) - 1
k0 = 0xC3A5C85C97CB3127
k1 = 0xB492B66FBE98F273
k2 = 0x9AE16A3B2F90404F


def _f64(s, i):
    return int.from_bytes(s[i:i + 8], "little")


def _f32(s, i):
    return int.from_bytes(s[i:i + 4], "little")


def _rot(v, n):
    v &= M64
    return v if n == 0 else ((v >> n) | (v << (64 - n))) & M64


def _smix(v):
    v &= M64
    return v ^ (v >> 47)


def _h16(u, v, mul):
    a = ((u ^ v) * mul) & M64
    a ^= a >> 47
    b = ((v ^ a) * mul) & M64
    b ^= b >> 47
    return (b * mul) & M64


def _weak(s, i, a, b):
    w, x, y, z = _f64(s, i), _f64(s, i + 8), _f64(s, i + 16), _f64(s, i + 24)
    a = (a + w) & M64
    b = _rot((b + a + z) & M64, 21)
    c = a
    a = (a + x) & M64
    a = (a + y) & M64
    b = (b + _rot(a, 44)) & M64
    return (a + z) & M64, (b + c) & M64


def fingerprint64(s: bytes) -> int:
    n = len(s)
    if n == 0:
        return k2
    if n <= 3:
        a, b, c = s[0], s[n >> 1], s[n - 1]
        y = (a + (b << 8)) & 0xFFFFFFFF
        z = (n + (c << 2)) & 0xFFFFFFFF
        return (_smix(((y * k2) & M64) ^ ((z * k0) & M64)) * k2) & M64
    if n <= 7:
        mul = (k2 + n * 2) & M64
        a = _f32(s, 0)
        return _h16((n + (a << 3)) & M64, _f32(s, n - 4), mul)
    if n <= 16:
        mul = (k2 + n * 2) & M64
        a = (_f64(s, 0) + k2) & M64
        b = _f64(s, n - 8)
        c = (_rot(b, 37) * mul + a) & M64
        d = ((_rot(a, 25) + b) * mul) & M64
        return _h16(c, d, mul)
    if n <= 32:
        mul = (k2 + n * 2) & M64
        a = (_f64(s, 0) * k1) & M64
        b = _f64(s, 8)
        c = (_f64(s, n - 8) * mul) & M64
        d = (_f64(s, n - 16) * k2) & M64
        return _h16((_rot((a + b) & M64, 43) + _rot(c, 30) + d) & M64,
                    (a + _rot((b + k2) & M64, 18) + c) & M64, mul)
    if n <= 64:
        mul = (k2 + n * 2) & M64
        a = (_f64(s, 0) * k2) & M64
        b = _f64(s, 8)
        c = (_f64(s, n - 8) * mul) & M64
        d = (_f64(s, n - 16) * k2) & M64
        y = (_rot((a + b) & M64, 43) + _rot(c, 30) + d) & M64
        z = _h16(y, (a + _rot((b + k2) & M64, 18) + c) & M64, mul)
        e = (_f64(s, 16) * mul) & M64
        f = _f64(s, 24)
        g = ((y + _f64(s, n - 32)) * mul) & M64
        h = ((z + _f64(s, n - 24)) * mul) & M64
        return _h16((_rot((e + f) & M64, 43) + _rot(g, 30) + h) & M64,
                    (e + _rot((f + a) & M64, 18) + g) & M64, mul)
    # > 64 bytes
    seed = 81
    x = seed
    y = (seed * k1 + 113) & M64
    z = (_smix((y * k2 + 113) & M64) * k2) & M64
    v = (0, 0)
    w = (0, 0)
    x = (x * k2 + _f64(s, 0)) & M64
    end = ((n - 1) // 64) * 64
    p = 0
    while True:
        x = (_rot((x + y + v[0] + _f64(s, p + 8)) & M64, 37) * k1) & M64
        y = (_rot((y + v[1] + _f64(s, p + 48)) & M64, 42) * k1) & M64
        x ^= w[1]
        y = (y + v[0] + _f64(s, p + 40)) & M64
        z = (_rot((z + w[0]) & M64, 33) * k1) & M64
        v = _weak(s, p, (v[1] * k1) & M64, (x + w[0]) & M64)
        w = _weak(s, p + 32, (z + w[1]) & M64, (y + _f64(s, p + 16)) & M64)
        z, x = x, z
        p += 64
        if p == end:
            break
    mul = (k1 + ((z & 0xFF) << 1)) & M64
    p = n - 64
    w = ((w[0] + ((n - 1) & 63)) & M64, w[1])
    v = ((v[0] + w[0]) & M64, v[1])
    w = ((w[0] + v[0]) & M64, w[1])
    x = (_rot((x + y + v[0] + _f64(s, p + 8)) & M64, 37) * mul) & M64
    y = (_rot((y + v[1] + _f64(s, p + 48)) & M64, 42) * mul) & M64
    x ^= (w[1] * 9) & M64
    y = (y + v[0] * 9 + _f64(s, p + 40)) & M64
    z = (_rot((z + w[0]) & M64, 33) * mul) & M64
    v = _weak(s, p, (v[1] * mul) & M64, (x + w[0]) & M64)
    w = _weak(s, p + 32, (z + w[1]) & M64, (y + _f64(s, p + 16)) & M64)
    z, x = x, z
    return _h16((_h16(v[0], w[0], mul) + ((_smix(y) * k0) & M64) + z) & M64,
                (_h16(v[1], w[1], mul) + x) & M64, mul)


def fingerprint_cat64(fp1: int, fp2: int) -> int:
    kmul = 0xC6A4A7935BD1E995
    r = (fp1 ^ kmul) & M64
    r ^= (_smix((fp2 * kmul) & M64) * kmul) & M64
    r = (r * kmul) & M64
    r = (_smix(r) * kmul) & M64
    return _smix(r)
