// wide_deep_amd/csrc/hash.hip -- bit-exact feature hashing on gfx950.
//
// Replaces the TF ops the reference selects at
//   python/lib/build_estimator.py:86-88   categorical_column_with_hash_bucket -> string_to_hash_bucket_fast
//   python/lib/build_estimator.py:138-155 crossed_column -> SparseCross (hashed)
// i.e. FarmHash farmhashna::Hash64 ("Fingerprint64") and TF FingerprintCat64.
//
// Integer, HBM/latency-bound work: one token per lane, tokens packed back to back so a wavefront
// touches one contiguous span of the byte buffer.  No LDS needed; 64-bit integer VALU only.
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace {

constexpr uint64_t k0 = 0xc3a5c85c97cb3127ULL;
constexpr uint64_t k1 = 0xb492b66fbe98f273ULL;
constexpr uint64_t k2 = 0x9ae16a3b2f90404fULL;

__device__ __forceinline__ uint64_t fetch64(const uint8_t *p) {
  uint64_t r;
  __builtin_memcpy(&r, p, 8);
  return r;
}
__device__ __forceinline__ uint64_t fetch32(const uint8_t *p) {
  uint32_t r;
  __builtin_memcpy(&r, p, 4);
  return (uint64_t)r;
}
__device__ __forceinline__ uint64_t rotr(uint64_t v, int s) { return s == 0 ? v : ((v >> s) | (v << (64 - s))); }
__device__ __forceinline__ uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }
__device__ __forceinline__ uint64_t hash_len16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul;
  a ^= (a >> 47);
  uint64_t b = (v ^ a) * mul;
  b ^= (b >> 47);
  return b * mul;
}

struct Pair {
  uint64_t first, second;
};

__device__ __forceinline__ Pair weak32(const uint8_t *s, uint64_t a, uint64_t b) {
  uint64_t w = fetch64(s), x = fetch64(s + 8), y = fetch64(s + 16), z = fetch64(s + 24);
  a += w;
  b = rotr(b + a + z, 21);
  uint64_t c = a;
  a += x;
  a += y;
  b += rotr(a, 44);
  return Pair{a + z, b + c};
}

__device__ uint64_t fingerprint64(const uint8_t *s, uint32_t len) {
  if (len <= 16) {
    if (len >= 8) {
      uint64_t mul = k2 + (uint64_t)len * 2;
      uint64_t a = fetch64(s) + k2;
      uint64_t b = fetch64(s + len - 8);
      uint64_t c = rotr(b, 37) * mul + a;
      uint64_t d = (rotr(a, 25) + b) * mul;
      return hash_len16(c, d, mul);
    }
    if (len >= 4) {
      uint64_t mul = k2 + (uint64_t)len * 2;
      uint64_t a = fetch32(s);
      return hash_len16((uint64_t)len + (a << 3), fetch32(s + len - 4), mul);
    }
    if (len > 0) {
      uint8_t a = s[0], b = s[len >> 1], c = s[len - 1];
      uint32_t y = (uint32_t)a + ((uint32_t)b << 8);
      uint32_t z = len + ((uint32_t)c << 2);
      return shift_mix((uint64_t)y * k2 ^ (uint64_t)z * k0) * k2;
    }
    return k2;
  }
  if (len <= 32) {
    uint64_t mul = k2 + (uint64_t)len * 2;
    uint64_t a = fetch64(s) * k1;
    uint64_t b = fetch64(s + 8);
    uint64_t c = fetch64(s + len - 8) * mul;
    uint64_t d = fetch64(s + len - 16) * k2;
    return hash_len16(rotr(a + b, 43) + rotr(c, 30) + d, a + rotr(b + k2, 18) + c, mul);
  }
  if (len <= 64) {
    uint64_t mul = k2 + (uint64_t)len * 2;
    uint64_t a = fetch64(s) * k2;
    uint64_t b = fetch64(s + 8);
    uint64_t c = fetch64(s + len - 8) * mul;
    uint64_t d = fetch64(s + len - 16) * k2;
    uint64_t y = rotr(a + b, 43) + rotr(c, 30) + d;
    uint64_t z = hash_len16(y, a + rotr(b + k2, 18) + c, mul);
    uint64_t e = fetch64(s + 16) * mul;
    uint64_t f = fetch64(s + 24);
    uint64_t g = (y + fetch64(s + len - 32)) * mul;
    uint64_t h = (z + fetch64(s + len - 24)) * mul;
    return hash_len16(rotr(e + f, 43) + rotr(g, 30) + h, e + rotr(f + a, 18) + g, mul);
  }
  const uint64_t seed = 81;
  uint64_t x = seed;
  uint64_t y = seed * k1 + 113;
  uint64_t z = shift_mix(y * k2 + 113) * k2;
  Pair v{0, 0}, w{0, 0};
  x = x * k2 + fetch64(s);
  const uint8_t *end = s + ((len - 1) / 64) * 64;
  const uint8_t *last64 = end + ((len - 1) & 63) - 63;
  do {
    x = rotr(x + y + v.first + fetch64(s + 8), 37) * k1;
    y = rotr(y + v.second + fetch64(s + 48), 42) * k1;
    x ^= w.second;
    y += v.first + fetch64(s + 40);
    z = rotr(z + w.first, 33) * k1;
    v = weak32(s, v.second * k1, x + w.first);
    w = weak32(s + 32, z + w.second, y + fetch64(s + 16));
    uint64_t t = z;
    z = x;
    x = t;
    s += 64;
  } while (s != end);
  uint64_t mul = k1 + ((z & 0xff) << 1);
  s = last64;
  w.first += ((len - 1) & 63);
  v.first += w.first;
  w.first += v.first;
  x = rotr(x + y + v.first + fetch64(s + 8), 37) * mul;
  y = rotr(y + v.second + fetch64(s + 48), 42) * mul;
  x ^= w.second * 9;
  y += v.first * 9 + fetch64(s + 40);
  z = rotr(z + w.first, 33) * mul;
  v = weak32(s, v.second * mul, x + w.first);
  w = weak32(s + 32, z + w.second, y + fetch64(s + 16));
  {
    uint64_t t = z;
    z = x;
    x = t;
  }
  return hash_len16(hash_len16(v.first, w.first, mul) + shift_mix(y) * k0 + z, hash_len16(v.second, w.second, mul) + x,
                    mul);
}

__device__ __forceinline__ uint64_t fingerprint_cat64(uint64_t fp1, uint64_t fp2) {
  const uint64_t kMul = 0xc6a4a7935bd1e995ULL;
  uint64_t r = fp1 ^ kMul;
  r ^= shift_mix(fp2 * kMul) * kMul;
  r *= kMul;
  r = shift_mix(r) * kMul;
  r = shift_mix(r);
  return r;
}

__global__ void k_fingerprint64(const uint8_t *__restrict__ bytes, const int32_t *__restrict__ tok_offs, int64_t ntok,
                                uint64_t *__restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntok) return;
  int32_t o0 = tok_offs[t], o1 = tok_offs[t + 1];
  out[t] = fingerprint64(bytes + o0, (uint32_t)(o1 - o0));
}

// first bag g with token_bag_offs[g+1] > t  (bags may be empty)
__device__ __forceinline__ int64_t bag_of_token(const int32_t *__restrict__ offs, int64_t nbags, int32_t t) {
  int64_t lo = 0, hi = nbags - 1;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (offs[mid + 1] > t) hi = mid; else lo = mid + 1;
  }
  return lo;
}

__global__ void k_hash_bucket(const uint8_t *__restrict__ bytes, const int32_t *__restrict__ tok_offs, int64_t ntok,
                              const int32_t *__restrict__ token_bag_offs, int64_t nbags,
                              const wd_slot_t *__restrict__ slots, int32_t S, int32_t *__restrict__ out_ids,
                              int32_t *__restrict__ out_cols) {
  WD_SIDE_PRIO();
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntok) return;
  int32_t o0 = tok_offs[t], o1 = tok_offs[t + 1];
  uint64_t fp = fingerprint64(bytes + o0, (uint32_t)(o1 - o0));
  int64_t bag = token_bag_offs ? bag_of_token(token_bag_offs, nbags, (int32_t)t) : t;
  uint64_t nb = (uint64_t)slots[bag % S].num_buckets;
  const int32_t id = (int32_t)(fp % nb);
  out_ids[t] = id;
  if (out_cols) out_cols[(bag % S) * (nbags / S) + bag / S] = id;   // slot-major copy (one token per bag): column s = ids of slot s
}

__global__ void k_emit_hash_slot(const uint64_t *__restrict__ fp, const int32_t *__restrict__ feat_offs, int64_t batch,
                                 uint64_t num_buckets, const int32_t *__restrict__ bag_offs, int32_t S, int32_t slot,
                                 int32_t *__restrict__ ids) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  int32_t f0 = feat_offs[b], f1 = feat_offs[b + 1];
  int32_t o = bag_offs[b * S + slot];
  for (int32_t j = f0; j < f1; ++j) ids[o + (j - f0)] = (int32_t)(fp[j] % num_buckets);
}

__global__ void k_emit_int_slot(const int32_t *__restrict__ vals, const int32_t *__restrict__ feat_offs, int64_t batch,
                                const int32_t *__restrict__ bag_offs, int32_t S, int32_t slot,
                                int32_t *__restrict__ ids) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  int32_t f0 = feat_offs[b], f1 = feat_offs[b + 1];
  int32_t o = bag_offs[b * S + slot];
  for (int32_t j = f0; j < f1; ++j) ids[o + (j - f0)] = vals[j];
}

struct CrossArgs {
  const uint64_t *vals[WD_MAX_CROSS_KEYS];
  const int32_t *offs[WD_MAX_CROSS_KEYS];
  int32_t nkeys;
};

// One lane per example; the product of a cross is tiny (1..prod Lmax) for CTR data.
__global__ void k_cross_hash(CrossArgs a, int64_t batch, uint64_t hash_key, uint64_t num_buckets,
                             const int32_t *__restrict__ bag_offs, int32_t S, int32_t slot, int32_t *__restrict__ ids) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  int32_t base[WD_MAX_CROSS_KEYS], cnt[WD_MAX_CROSS_KEYS];
  int64_t total = 1;
#pragma unroll
  for (int k = 0; k < WD_MAX_CROSS_KEYS; ++k) {
    if (k < a.nkeys) {
      base[k] = a.offs[k][b];
      cnt[k] = a.offs[k][b + 1] - base[k];
      total *= cnt[k];
    }
  }
  int32_t o = bag_offs[b * S + slot];
  const uint64_t m = num_buckets > 0 ? num_buckets : (uint64_t)INT64_MAX;
  for (int64_t j = 0; j < total; ++j) {
    // mixed-radix decode of j, last key fastest: digits consumed from the last key backwards,
    // FingerprintCat64 must be applied in key order, so decode first.
    int32_t idx[WD_MAX_CROSS_KEYS];
    int64_t r = j;
#pragma unroll
    for (int k = WD_MAX_CROSS_KEYS - 1; k >= 0; --k) {
      if (k < a.nkeys) {
        idx[k] = (int32_t)(r % cnt[k]);
        r /= cnt[k];
      }
    }
    uint64_t h = hash_key;
#pragma unroll
    for (int k = 0; k < WD_MAX_CROSS_KEYS; ++k) {
      if (k < a.nkeys) h = fingerprint_cat64(h, a.vals[k][base[k] + idx[k]]);
    }
    ids[o + j] = (int32_t)(h % m);
  }
}

// ---- featurizer on the device (python/lib/build_estimator.py:83-158 for a whole batch in four launches) --------------------
// One thread per (example, slot).  The slot table says what the column is; the batch descriptor where its raw values are:
// tokens of string features as fingerprints fp[] (+ vocabulary indices tok_val[] for vocabulary_list columns), integer and
// float features as [feature][example] matrices.  `feat_count` = how many ids the column gives the example, `feat_emit`
// writes them at the bag's offset -- the same walk twice, so that no per-slot compacted array ever exists.
__device__ __forceinline__ int32_t feat_bucketize(const float *__restrict__ bounds, int32_t n, float x) {
  int32_t lo = 0, hi = n;            // number of boundaries <= x (np.searchsorted(side='right'), tf bucketized_column)
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (bounds[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ float feat_normalize(float x, int32_t kind, float p0, float p1) {
  if (kind == 1) return (x - p0) / (p1 - p0);
  if (kind == 2) return (x - p0) / p1;
  return x;                           // (log-normalised columns arrive normalised: the host applies np.log)
}

// count of key k of a crossed column for example b, and its j-th value (uint64: fingerprint / int64 id)
__device__ __forceinline__ int32_t feat_key_count(const wd_feat_key_t &k, const wd_feat_batch_t &q, int64_t b) {
  if (k.kind == WD_FEAT_KEY_STRING) {
    if (q.lmax) return q.lmax[k.src];                              // tf_dense: padded_batch + '' padding (quirk C.16)
    const int32_t *eo = q.ex_offs + (int64_t)k.src * (q.batch + 1);
    return eo[b + 1] - eo[b];
  }
  if (k.kind == WD_FEAT_KEY_IDENTITY) return q.ints[(int64_t)k.src * q.batch + b] != -1 ? 1 : 0;
  return 1;
}

__device__ __forceinline__ uint64_t feat_key_value(const wd_feat_key_t &k, const wd_feat_batch_t &q, int64_t b, int32_t j) {
  if (k.kind == WD_FEAT_KEY_STRING) {
    const int32_t *eo = q.ex_offs + (int64_t)k.src * (q.batch + 1);
    const int32_t real = eo[b + 1] - eo[b];
    return j < real ? q.fp[q.tok_base[k.src] + eo[b] + j] : q.fp[q.empty_index];
  }
  if (k.kind == WD_FEAT_KEY_IDENTITY) {
    const int64_t v = q.ints[(int64_t)k.src * q.batch + b];
    return (uint64_t)((v >= 0 && v < k.num_buckets) ? v : 0);      // default_value = 0 for out-of-range ids
  }
  return (uint64_t)feat_bucketize(q.bounds + k.bound_off, k.nbound, q.floats[(int64_t)k.src * q.batch + b]);
}

template <bool EMIT>
__global__ void __launch_bounds__(256)
k_feat(const wd_feat_slot_t *__restrict__ slots, wd_feat_batch_t q, int32_t *__restrict__ lens,
       const int32_t *__restrict__ bag_offs, int32_t *__restrict__ ids) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= q.batch * q.S) return;
  const int64_t b = i / q.S;
  const wd_feat_slot_t &s = slots[i - b * q.S];
  int32_t *out = EMIT ? ids + bag_offs[i] : nullptr;
  int32_t n = 0;
  switch (s.kind) {
    case WD_FEAT_HASH: {
      const int32_t *eo = q.ex_offs + (int64_t)s.src * (q.batch + 1);
      n = eo[b + 1] - eo[b];
      if (EMIT) {
        const uint64_t *f = q.fp + q.tok_base[s.src] + eo[b];
        for (int32_t j = 0; j < n; ++j) out[j] = (int32_t)(f[j] % (uint64_t)s.num_buckets);
      }
      break;
    }
    case WD_FEAT_VOCAB: {
      const int32_t *eo = q.ex_offs + (int64_t)s.src * (q.batch + 1);
      const int32_t *tv = q.tok_val + q.tok_base[s.src] + eo[b];
      const int32_t m = eo[b + 1] - eo[b];
      for (int32_t j = 0; j < m; ++j) {
        const int32_t v = tv[j];
        if (v >= 0) {                  // out-of-vocabulary tokens are dropped (default_value = -1)
          if (EMIT) out[n] = v;
          ++n;
        }
      }
      break;
    }
    case WD_FEAT_IDENTITY: {
      const int64_t v = q.ints[(int64_t)s.src * q.batch + b];
      if (v != -1) {                   // -1 is the ignore_value of the dense -> sparse conversion
        if (EMIT) out[0] = (int32_t)((v >= 0 && v < s.num_buckets) ? v : 0);
        n = 1;
      }
      break;
    }
    case WD_FEAT_BUCKET: {
      n = 1;
      if (EMIT) out[0] = feat_bucketize(q.bounds + s.bound_off, s.nbound,
                                        feat_normalize(q.floats[(int64_t)s.src * q.batch + b], s.norm_kind, s.p0, s.p1));
      break;
    }
    default: {                         // WD_FEAT_CROSS: cartesian product, LAST key fastest, FingerprintCat64 in key order
      int32_t cnt[WD_MAX_CROSS_KEYS];
      int64_t total = 1;
      for (int k = 0; k < s.nkeys; ++k) {
        cnt[k] = feat_key_count(s.keys[k], q, b);
        total *= cnt[k];
      }
      n = (int32_t)total;
      if (EMIT) {
        const uint64_t m = s.num_buckets > 0 ? (uint64_t)s.num_buckets : (uint64_t)INT64_MAX;
        for (int64_t j = 0; j < total; ++j) {
          int32_t idx[WD_MAX_CROSS_KEYS];
          int64_t r = j;
          for (int k = s.nkeys - 1; k >= 0; --k) {
            idx[k] = (int32_t)(r % cnt[k]);
            r /= cnt[k];
          }
          uint64_t h = s.hash_key;
          for (int k = 0; k < s.nkeys; ++k) h = fingerprint_cat64(h, feat_key_value(s.keys[k], q, b, idx[k]));
          out[j] = (int32_t)(h % m);
        }
      }
    }
  }
  if (!EMIT) lens[i] = n;
}

// index in vocabulary_list of every token in [t0, t0 + n) (byte-wise equality; -1: not in the list)
__global__ void __launch_bounds__(256)
k_feat_vocab(const uint8_t *__restrict__ bytes, const int32_t *__restrict__ tok_offs, int64_t t0, int64_t n,
             const uint8_t *__restrict__ vbytes, const int32_t *__restrict__ voffs, int32_t nv,
             int32_t *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int32_t o0 = tok_offs[t0 + t], len = tok_offs[t0 + t + 1] - o0;
  int32_t hit = -1;
  for (int32_t v = 0; v < nv && hit < 0; ++v) {
    const int32_t w0 = voffs[v];
    if (voffs[v + 1] - w0 != len) continue;
    bool eq = true;
    for (int32_t c = 0; c < len; ++c) eq = eq && bytes[o0 + c] == vbytes[w0 + c];
    if (eq) hit = v;
  }
  out[t0 + t] = hit;
}

__global__ void k_feat_first_zero(int32_t *offs) { offs[0] = 0; }

}  // namespace

static bool feat_batch_ok(const wd_feat_batch_t *q) {
  return q && q->batch >= 0 && q->S > 0 && q->S <= (1 << 16);
}

extern "C" int wd_feat_vocab_lookup(const uint8_t *bytes, const int32_t *tok_offs, int64_t tok_begin, int64_t n,
                                    const uint8_t *vocab_bytes, const int32_t *vocab_offs, int32_t nvocab, int32_t *tok_val,
                                    wd_stream_t stream) {
  if (n <= 0) return WD_OK;
  WD_REQUIRE(bytes && tok_offs && vocab_bytes && vocab_offs && tok_val && nvocab > 0, "null pointer");
  hipLaunchKernelGGL(k_feat_vocab, dim3((unsigned)wd::ceil_div(n, 256)), dim3(256), 0, wd::as_stream(stream), bytes, tok_offs,
                     tok_begin, n, vocab_bytes, vocab_offs, nvocab, tok_val);
  return wd::check_launch("wd_feat_vocab_lookup");
}

extern "C" int wd_feat_lens(const wd_feat_slot_t *slots_dev, const wd_feat_batch_t *batch, int32_t *lens, wd_stream_t stream) {
  WD_REQUIRE(slots_dev && feat_batch_ok(batch) && lens, "null pointer / bad batch descriptor");
  if (batch->batch == 0) return WD_OK;
  hipLaunchKernelGGL((k_feat<false>), dim3((unsigned)wd::ceil_div(batch->batch * batch->S, 256)), dim3(256), 0,
                     wd::as_stream(stream), slots_dev, *batch, lens, (const int32_t *)nullptr, (int32_t *)nullptr);
  return wd::check_launch("wd_feat_lens");
}

extern "C" int wd_feat_emit(const wd_feat_slot_t *slots_dev, const wd_feat_batch_t *batch, const int32_t *bag_offs,
                            int32_t *ids, wd_stream_t stream) {
  WD_REQUIRE(slots_dev && feat_batch_ok(batch) && bag_offs && ids, "null pointer / bad batch descriptor");
  if (batch->batch == 0) return WD_OK;
  hipLaunchKernelGGL((k_feat<true>), dim3((unsigned)wd::ceil_div(batch->batch * batch->S, 256)), dim3(256), 0,
                     wd::as_stream(stream), slots_dev, *batch, (int32_t *)nullptr, bag_offs, ids);
  return wd::check_launch("wd_feat_emit");
}

// bag CSR from the lengths: offs[0] = 0, offs[i + 1] = lens[0] + ... + lens[i]   (offs[n] = number of ids of the batch)
extern "C" int64_t wd_feat_offsets_workspace_bytes(int64_t n) {
  size_t bytes = 0;
  if (rocprim::inclusive_scan(nullptr, bytes, (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)(n > 0 ? n : 1),
                              rocprim::plus<int32_t>()) != hipSuccess)
    return -1;
  return (int64_t)(bytes ? bytes : 16);
}

extern "C" int wd_feat_offsets(const int32_t *lens, int64_t n, int32_t *offs, void *workspace, int64_t workspace_bytes,
                               wd_stream_t stream) {
  WD_REQUIRE(offs, "null pointer");
  hipStream_t st = wd::as_stream(stream);
  hipLaunchKernelGGL(k_feat_first_zero, dim3(1), dim3(1), 0, st, offs);
  if (n > 0) {
    WD_REQUIRE(lens && workspace, "null pointer");
    size_t bytes = (size_t)workspace_bytes;
    if (rocprim::inclusive_scan(workspace, bytes, lens, offs + 1, (size_t)n, rocprim::plus<int32_t>(), st) != hipSuccess) {
      wd::set_error("wd_feat_offsets: rocprim::inclusive_scan failed");
      return WD_ERR_LAUNCH;
    }
  }
  return wd::check_launch("wd_feat_offsets");
}

extern "C" int wd_fingerprint64(const uint8_t *bytes, const int32_t *tok_offs, int64_t ntok, uint64_t *out_fp,
                                wd_stream_t stream) {
  if (ntok <= 0) return WD_OK;
  WD_REQUIRE(bytes && tok_offs && out_fp, "null pointer");
  hipLaunchKernelGGL(k_fingerprint64, dim3((unsigned)wd::ceil_div(ntok, 256)), dim3(256), 0, wd::as_stream(stream),
                     bytes, tok_offs, ntok, out_fp);
  return wd::check_launch("wd_fingerprint64");
}

extern "C" int wd_hash_bucket(const uint8_t *bytes, const int32_t *tok_offs, int64_t ntok,
                              const int32_t *token_bag_offs, int64_t nbags, const wd_slot_t *slots, int32_t S,
                              int32_t *out_ids, wd_stream_t stream) {
  if (ntok <= 0) return WD_OK;
  WD_REQUIRE(bytes && tok_offs && slots && out_ids, "null pointer");
  WD_REQUIRE(S > 0, "S must be > 0");
  WD_REQUIRE(token_bag_offs || nbags == ntok, "one-token-per-bag mode needs nbags == ntok");
  hipLaunchKernelGGL(k_hash_bucket, dim3((unsigned)wd::ceil_div(ntok, 256)), dim3(256), 0, wd::as_stream(stream),
                     bytes, tok_offs, ntok, token_bag_offs, nbags, slots, S, out_ids, (int32_t *)nullptr);
  return wd::check_launch("wd_hash_bucket");
}

extern "C" int wd_hash_bucket_cols(const uint8_t *bytes, const int32_t *tok_offs, int64_t nbags, const wd_slot_t *slots,
                                   int32_t S, int32_t *out_ids, int32_t *out_ids_cols, wd_stream_t stream) {
  if (nbags <= 0) return WD_OK;
  WD_REQUIRE(bytes && tok_offs && slots && out_ids && out_ids_cols, "null pointer");
  WD_REQUIRE(S > 0 && nbags % S == 0, "nbags must be batch * S");
  hipLaunchKernelGGL(k_hash_bucket, dim3((unsigned)wd::ceil_div(nbags, 256)), dim3(256), 0, wd::as_stream(stream),
                     bytes, tok_offs, nbags, (const int32_t *)nullptr, nbags, slots, S, out_ids, out_ids_cols);
  return wd::check_launch("wd_hash_bucket_cols");
}

extern "C" int wd_emit_hash_slot(const uint64_t *fp, const int32_t *feat_offs, int64_t batch, uint64_t num_buckets,
                                 const int32_t *bag_offs, int32_t S, int32_t slot, int32_t *ids, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(feat_offs && bag_offs && ids, "null pointer");
  WD_REQUIRE(num_buckets > 0 && num_buckets < (1ull << 31), "num_buckets out of range");
  WD_REQUIRE(slot >= 0 && slot < S, "slot out of range");
  hipLaunchKernelGGL(k_emit_hash_slot, dim3((unsigned)wd::ceil_div(batch, 256)), dim3(256), 0, wd::as_stream(stream),
                     fp, feat_offs, batch, num_buckets, bag_offs, S, slot, ids);
  return wd::check_launch("wd_emit_hash_slot");
}

extern "C" int wd_emit_int_slot(const int32_t *vals, const int32_t *feat_offs, int64_t batch, const int32_t *bag_offs,
                                int32_t S, int32_t slot, int32_t *ids, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(feat_offs && bag_offs && ids, "null pointer");
  WD_REQUIRE(slot >= 0 && slot < S, "slot out of range");
  hipLaunchKernelGGL(k_emit_int_slot, dim3((unsigned)wd::ceil_div(batch, 256)), dim3(256), 0, wd::as_stream(stream),
                     vals, feat_offs, batch, bag_offs, S, slot, ids);
  return wd::check_launch("wd_emit_int_slot");
}

extern "C" int wd_cross_hash(const wd_cross_keys_t *keys_host, int64_t batch, uint64_t hash_key, uint64_t num_buckets,
                             const int32_t *bag_offs, int32_t S, int32_t slot, int32_t *ids, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(keys_host && bag_offs && ids, "null pointer");
  WD_REQUIRE(keys_host->nkeys >= 1 && keys_host->nkeys <= WD_MAX_CROSS_KEYS, "nkeys out of range");
  WD_REQUIRE(num_buckets < (1ull << 31), "num_buckets must fit int32 ids");
  WD_REQUIRE(slot >= 0 && slot < S, "slot out of range");
  CrossArgs a;
  a.nkeys = keys_host->nkeys;
  for (int k = 0; k < WD_MAX_CROSS_KEYS; ++k) {
    a.vals[k] = k < a.nkeys ? keys_host->vals[k] : nullptr;
    a.offs[k] = k < a.nkeys ? keys_host->offs[k] : nullptr;
  }
  hipLaunchKernelGGL(k_cross_hash, dim3((unsigned)wd::ceil_div(batch, 256)), dim3(256), 0, wd::as_stream(stream), a,
                     batch, hash_key, num_buckets, bag_offs, S, slot, ids);
  return wd::check_launch("wd_cross_hash");
}
