// wide_deep_amd/csrc/hash.hip -- bit-exact feature hashing on gfx950.
//
// Replaces the TF ops the reference selects at
//   python/lib/build_estimator.py:86-88   categorical_column_with_hash_bucket -> string_to_hash_bucket_fast
//   python/lib/build_estimator.py:138-155 crossed_column -> SparseCross (hashed)
// i.e. FarmHash farmhashna::Hash64 ("Fingerprint64") and TF FingerprintCat64.
//
// Integer, HBM/latency-bound work: one token per lane, tokens packed back to back so a wavefront
// touches one contiguous span of the byte buffer.  No LDS needed; 64-bit integer VALU only.
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
                                const int32_t *__restrict__ ntok_dev, uint64_t *__restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ntok_dev) ntok = (int64_t)*ntok_dev + 1;       // the batch's token count lives on the device: tokens [0, T] (T = the trailing '')
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
    return j < real ? q.fp[q.tok_base[k.src] + eo[b] + j] : q.fp[q.ntok_dev ? *q.ntok_dev : q.empty_index];
  }
  if (k.kind == WD_FEAT_KEY_IDENTITY) {
    const int64_t v = q.ints[(int64_t)k.src * q.batch + b];
    return (uint64_t)((v >= 0 && v < k.num_buckets) ? v : 0);      // default_value = 0 for out-of-range ids
  }
  return (uint64_t)feat_bucketize(q.bounds + k.bound_off, k.nbound, q.floats[(int64_t)k.src * q.batch + b]);
}

// ids per (example, column): thread per pair; per block of 256 pairs their sum and "some bag does not hold exactly one id"
// (block_stats[2 * block], [2 * block + 1]: what k_feat_offsets builds the bag CSR and the batch flags from -- no zeroed flag, no
// device-wide scan library in the featurizer's launch sequence)
__global__ void __launch_bounds__(256)
k_feat_lens(const wd_feat_slot_t *__restrict__ slots, wd_feat_batch_t q, int32_t *__restrict__ lens, int32_t *__restrict__ block_stats) {
  __shared__ int32_t wsum[4], wmulti[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < q.batch * q.S;
  const int64_t b = live ? i / q.S : 0;
  const wd_feat_slot_t &s = slots[live ? i - b * q.S : 0];
  int32_t n = 0;
  if (live) switch (s.kind) {
    case WD_FEAT_HASH: {
      const int32_t *eo = q.ex_offs + (int64_t)s.src * (q.batch + 1);
      n = eo[b + 1] - eo[b];
      break;
    }
    case WD_FEAT_VOCAB: {
      const int32_t *eo = q.ex_offs + (int64_t)s.src * (q.batch + 1);
      const int32_t *tv = q.tok_val + q.tok_base[s.src] + eo[b];
      const int32_t m = eo[b + 1] - eo[b];
      for (int32_t j = 0; j < m; ++j) n += tv[j] >= 0 ? 1 : 0;      // out-of-vocabulary tokens are dropped (default_value = -1)
      break;
    }
    case WD_FEAT_IDENTITY:
      n = q.ints[(int64_t)s.src * q.batch + b] != -1 ? 1 : 0;       // -1 is the ignore_value of the dense -> sparse conversion
      break;
    case WD_FEAT_BUCKET:
      n = 1;
      break;
    default: {                         // WD_FEAT_CROSS: the cartesian product of its keys
      int64_t total = 1;
      for (int k = 0; k < s.nkeys; ++k) total *= feat_key_count(s.keys[k], q, b);
      n = (int32_t)total;
    }
  }
  if (live) lens[i] = n;
  int32_t sum = n;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
  const unsigned long long multi = __ballot(live && n != 1);
  if ((threadIdx.x & 63) == 0) {
    wsum[threadIdx.x >> 6] = sum;
    wmulti[threadIdx.x >> 6] = multi != 0ull;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    block_stats[2 * blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    block_stats[2 * blockIdx.x + 1] = wmulti[0] | wmulti[1] | wmulti[2] | wmulti[3];
  }
}

// bag CSR from the lengths: offs[0] = 0, offs[i + 1] = lens[0] + ... + lens[i].  Block k owns pairs [256 k, 256 k + 256): it adds up
// the sums of the blocks before it (one round of loads: <= 4 k blocks for a batch of 8192 x 128 columns) and scans its own 256
// lengths; block 0 also leaves the batch flags: flags[0] = the id count exceeds ids_capacity, flags[1] = some bag is not one id.
__global__ void __launch_bounds__(256)
k_feat_offsets(const int32_t *__restrict__ lens, const int32_t *__restrict__ block_stats, int64_t n, int32_t nblocks,
               int32_t *__restrict__ offs, int64_t ids_capacity, int32_t *__restrict__ flags) {
  __shared__ int32_t red[4], redm[4], wtot[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int upto = blockIdx.x == 0 ? nblocks : (int)blockIdx.x;     // block 0 sums every block (the id count) for the flags
  int32_t pre = 0, multi = 0;
  for (int k = t; k < upto; k += 256) {
    pre += block_stats[2 * k];
    multi |= block_stats[2 * k + 1];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    pre += __shfl_xor(pre, m, 64);
    multi |= __shfl_xor(multi, m, 64);
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + t;
  const int32_t v = i < n ? lens[i] : 0;
  int32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t u = __shfl_up(incl, off, 64);
    if (lane >= off) incl += u;
  }
  if (lane == 0) { red[w] = pre; redm[w] = multi; }
  if (lane == 63) wtot[w] = incl;
  __syncthreads();
  const int32_t total_before = red[0] + red[1] + red[2] + red[3];
  if (blockIdx.x == 0) {
    if (t == 0) {
      offs[0] = 0;
      if (flags) {
        flags[0] = (int64_t)total_before > ids_capacity ? 1 : 0;
        flags[1] = redm[0] | redm[1] | redm[2] | redm[3];
      }
    }
  }
  int32_t base = blockIdx.x == 0 ? 0 : total_before;
  for (int k = 0; k < w; ++k) base += wtot[k];
  if (i < n) offs[i + 1] = base + incl;
}


// ---- emission, one LANE per output id (round 6; rounds 2-5: one lane per bag, which for a crossed column walked the whole
// product -- 25-125 ids, two 64-bit divisions and a dependent load per key per id -- 2.6 ms for the 2.3 M ids of a configs[3] batch).
// A workgroup takes FE_EX consecutive examples = one contiguous span of the bag CSR and of the id array:
//   prologue   the span's bag offsets and, per (example, column), what an id is made from -- per key: count (with the ''
//              padding of quirk C.16), real count, first token / the immediate value -- into LDS; per column the modulus and its
//              Barrett reciprocal (one 64-bit division per column and workgroup instead of one per id);
//   prefixes   the LAST key varies fastest, so the ids of a bag share the chained hash of the leading keys in runs of count(last):
//              one lane per COMBINATION of the leading keys (25 per example for a three-key cross over slots of mean 5, for 125
//              ids) chains FingerprintCat64 over them into an LDS table (combinations beyond FE_PREFIX_CAP: chained per id);
//   ids        lane p of the span: example by comparison, column by binary search over the example's offsets in LDS, position j
//              in the bag; HASH: fp % buckets; CROSS: j -> (combination, digit of the last key) by one 32-bit division, ONE
//              FingerprintCat64 on the combination's hash, Barrett reduction; stores are consecutive 4-byte words of one span.
#ifndef FE_EXP
#define FE_EXP 0                     // ablation builds (profiles/run_scripts/build_feat_exp.sh): 1 stop behind the tables, 2 stop behind the
#endif                               // prefixes, 4 ids without the hashing (position in the bag stored)
constexpr int FE_EX = 2;             // examples per workgroup (fewer when the model has more than FE_MAX_PAIRS / 2 columns): 2 and 4 take the
                                     // same time alone (33.1 / 33.3 us), 2 needs half the LDS -- the launch then fits beside the one-launch tower
constexpr int FE_MAX_PAIRS = 1024;   // (example, column) pairs of a workgroup's LDS tables
constexpr int FE_PREFIX_CAP = 512;   // chained hashes of the LEADING keys' combinations kept per workgroup (4 KB; configs[3]: ~60 per workgroup)

struct FeCol {                        // per column, in LDS
  int32_t kind, nkeys;
  uint64_t hash_key, m, magic;        // m: modulus (num_buckets, or INT64_MAX for 0 buckets); magic = floor((2^64 - 1) / m)
};

__device__ __forceinline__ uint64_t barrett_mod(uint64_t h, uint64_t m, uint64_t magic) {
  const uint64_t qd = __umul64hi(h, magic);         // floor(h / m) - 2 <= qd <= floor(h / m)
  uint64_t r = h - qd * m;
  while (r >= m) r -= m;
  return r;
}

__global__ void __launch_bounds__(256)
k_feat_emit_par(const wd_feat_slot_t *__restrict__ slots, wd_feat_batch_t q, const int32_t *__restrict__ bag_offs,
                int32_t *__restrict__ ids, int32_t ex_per_wg, int64_t ids_cap) {
  extern __shared__ __attribute__((aligned(16))) int32_t fe_lds[];
  const int t = threadIdx.x;
  const int S = q.S;
  const int64_t b0 = (int64_t)blockIdx.x * ex_per_wg;
  const int E = (int)(q.batch - b0 < ex_per_wg ? q.batch - b0 : ex_per_wg);
  const int P = E * S;
  int32_t *offs = fe_lds;                                   // [P + 1]
  int32_t *kcnt = offs + (ex_per_wg * S + 1 + 3) / 4 * 4;   // [P][4] values a key contributes (tf_dense: incl. the '' padding)
  int32_t *kreal = kcnt + ex_per_wg * S * WD_MAX_CROSS_KEYS;   // [P][4] real tokens of a STRING key; -1: kbase IS the value
  int32_t *kbase = kreal + ex_per_wg * S * WD_MAX_CROSS_KEYS;  // [P][4] first token in fp / tok_val
  FeCol *col = reinterpret_cast<FeCol *>(kbase + ex_per_wg * S * WD_MAX_CROSS_KEYS);   // [S]
  int32_t *pbase = reinterpret_cast<int32_t *>(col + S);     // [P + 1] first prefix of a pair's crossed column (exclusive scan)
  __shared__ uint64_t hp[FE_PREFIX_CAP];
  __shared__ int32_t pscan[256];
  for (int i = t; i <= P; i += 256) offs[i] = bag_offs[b0 * S + i];
  for (int i = t; i < S; i += 256) {
    const wd_feat_slot_t &s = slots[i];
    FeCol c;
    c.kind = s.kind;
    c.nkeys = s.nkeys;
    c.hash_key = s.hash_key;
    c.m = s.num_buckets > 0 ? (uint64_t)s.num_buckets : (uint64_t)INT64_MAX;
    c.magic = ~0ull / c.m;
    col[i] = c;
  }
  for (int i = t; i < P; i += 256) {
    const int e = i / S, si = i - e * S;
    const int64_t b = b0 + e;
    const wd_feat_slot_t &s = slots[si];
    int32_t *c = kcnt + i * WD_MAX_CROSS_KEYS, *r = kreal + i * WD_MAX_CROSS_KEYS, *v = kbase + i * WD_MAX_CROSS_KEYS;
    switch (s.kind) {
      case WD_FEAT_HASH:
      case WD_FEAT_VOCAB: {
        const int32_t *eo = q.ex_offs + (int64_t)s.src * (q.batch + 1);
        const int32_t e0 = eo[b];
        v[0] = q.tok_base[s.src] + e0;
        r[0] = eo[b + 1] - e0;
        break;
      }
      case WD_FEAT_IDENTITY: {
        const int64_t x = q.ints[(int64_t)s.src * q.batch + b];
        v[0] = (int32_t)((x >= 0 && x < s.num_buckets) ? x : 0);
        break;
      }
      case WD_FEAT_BUCKET:
        v[0] = feat_bucketize(q.bounds + s.bound_off, s.nbound,
                              feat_normalize(q.floats[(int64_t)s.src * q.batch + b], s.norm_kind, s.p0, s.p1));
        break;
      default:
        for (int k = 0; k < s.nkeys; ++k) {
          const wd_feat_key_t &key = s.keys[k];
          if (key.kind == WD_FEAT_KEY_STRING) {
            const int32_t *eo = q.ex_offs + (int64_t)key.src * (q.batch + 1);
            const int32_t e0 = eo[b];
            r[k] = eo[b + 1] - e0;
            c[k] = q.lmax ? q.lmax[key.src] : r[k];              // tf_dense: padded_batch + '' padding (quirk C.16)
            v[k] = q.tok_base[key.src] + e0;
          } else {
            c[k] = feat_key_count(key, q, b);
            r[k] = -1;
            v[k] = (int32_t)feat_key_value(key, q, b, 0);
          }
        }
    }
  }
  const int32_t empty_index = q.ntok_dev ? *q.ntok_dev : q.empty_index;     // (uniform: a scalar load)
  __syncthreads();
  if (FE_EXP & 1) return;
  // ---- prefixes: combinations of the leading keys per crossed (example, column), scanned over the pairs ----
  const int PP = (P + 255) / 256;                       // pairs per thread (<= 4)
  int32_t mine = 0;
  for (int u = 0; u < PP; ++u) {
    const int i = t * PP + u;
    int32_t np = 0;
    if (i < P) {
      const FeCol &c = col[i % S];
      if (c.kind == WD_FEAT_CROSS) {
        np = 1;
        for (int k = 0; k + 1 < c.nkeys; ++k) np *= kcnt[i * WD_MAX_CROSS_KEYS + k];
        if (offs[i + 1] == offs[i]) np = 0;             // an empty bag needs none
      }
      pbase[i] = mine;                                  // exclusive within the thread's run
    }
    mine += np;
  }
  pscan[t] = mine;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {             // Hillis-Steele over the 256 thread sums
    const int32_t v = t >= off ? pscan[t - off] : 0;
    __syncthreads();
    pscan[t] += v;
    __syncthreads();
  }
  const int32_t texcl = pscan[t] - mine;
  for (int u = 0; u < PP; ++u) {
    const int i = t * PP + u;
    if (i < P) pbase[i] += texcl;
  }
  if (t == 255) pbase[P] = pscan[255];
  __syncthreads();
  const int32_t nprefix = pbase[P] < FE_PREFIX_CAP ? pbase[P] : FE_PREFIX_CAP;
  for (int32_t x = t; x < nprefix; x += 256) {
    int lo = 0, hi = P;                                 // last pair with pbase[pair] <= x (pairs without prefixes share an offset)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pbase[mid] <= x) lo = mid; else hi = mid;
    }
    const FeCol &c = col[lo % S];
    const int32_t *kc = kcnt + lo * WD_MAX_CROSS_KEYS, *kr = kreal + lo * WD_MAX_CROSS_KEYS, *kb = kbase + lo * WD_MAX_CROSS_KEYS;
    uint32_t r = (uint32_t)(x - pbase[lo]);
    uint32_t idx[WD_MAX_CROSS_KEYS];
#pragma unroll
    for (int k = WD_MAX_CROSS_KEYS - 2; k >= 0; --k) {
      if (k + 1 < c.nkeys) {
        const uint32_t n = (uint32_t)kc[k];
        const uint32_t d = r / n;
        idx[k] = r - d * n;
        r = d;
      }
    }
    uint64_t h = c.hash_key;
#pragma unroll
    for (int k = 0; k < WD_MAX_CROSS_KEYS - 1; ++k) {
      if (k + 1 < c.nkeys) {
        const int32_t real = kr[k];
        const uint64_t v = real < 0 ? (uint64_t)(int64_t)kb[k] : q.fp[(int32_t)idx[k] < real ? kb[k] + (int32_t)idx[k] : empty_index];
        h = fingerprint_cat64(h, v);
      }
    }
    hp[x] = h;
  }
  __syncthreads();
  if (FE_EXP & 2) return;
  const int64_t p0 = offs[0], p1 = offs[P];
  // (an id array that is too small: ids beyond it are dropped; wd_feat_offsets raised flags[0], the caller checks it)
  for (int64_t p = p0 + t; p < p1; p += 256) {
    if (p >= ids_cap) break;
    int e = 0;
    while (e + 1 < E && offs[(e + 1) * S] <= (int32_t)p) ++e;
    const int32_t *eo = offs + e * S;
    int lo = 0, hi = S;                            // last column with eo[column] <= p  (empty bags share an offset)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (eo[mid] <= (int32_t)p) lo = mid; else hi = mid;
    }
    const int pi = e * S + lo;
    const uint32_t j = (uint32_t)((int32_t)p - eo[lo]);
    const FeCol c = col[lo];
    const int32_t *kc = kcnt + pi * WD_MAX_CROSS_KEYS, *kr = kreal + pi * WD_MAX_CROSS_KEYS, *kb = kbase + pi * WD_MAX_CROSS_KEYS;
    int32_t out;
    if (FE_EXP & 4) {
      out = (int32_t)j + kc[0] + kr[0] + kb[0];
    } else if (c.kind == WD_FEAT_HASH || c.kind == WD_FEAT_CROSS) {
      // one fingerprint load and one Barrett reduction per id for BOTH kinds (a wavefront's 64 consecutive ids usually span hash
      // bags and a crossed bag: with a branch per kind it would run both reductions)
      const bool cross = c.kind == WD_FEAT_CROSS;
      int32_t fi = kb[0] + (int32_t)j;                  // HASH: the j-th token of the example's feature
      uint64_t h = 0;
      bool imm = false;
      int kl = 0;
      if (cross) {
        kl = c.nkeys - 1;                               // the last key: the fastest digit of j
        const uint32_t nl = (uint32_t)kc[kl];
        const uint32_t y = j / nl, il = j - y * nl;     // combination of the leading keys, digit of the last key
        const int32_t real = kr[kl];
        imm = real < 0;
        fi = imm ? empty_index : ((int32_t)il < real ? kb[kl] + (int32_t)il : empty_index);
        const int32_t px = pbase[pi] + (int32_t)y;
        if (px < FE_PREFIX_CAP) {
          h = hp[px];
        } else {                                        // beyond the table: chain the leading keys here
          uint32_t r = y, idx[WD_MAX_CROSS_KEYS];
#pragma unroll
          for (int k = WD_MAX_CROSS_KEYS - 2; k >= 0; --k) {
            if (k < kl) {
              const uint32_t n = (uint32_t)kc[k];
              const uint32_t d = r / n;
              idx[k] = r - d * n;
              r = d;
            }
          }
          h = c.hash_key;
#pragma unroll
          for (int k = 0; k < WD_MAX_CROSS_KEYS - 1; ++k) {
            if (k < kl) {
              const int32_t rk = kr[k];
              const uint64_t v = rk < 0 ? (uint64_t)(int64_t)kb[k] : q.fp[(int32_t)idx[k] < rk ? kb[k] + (int32_t)idx[k] : empty_index];
              h = fingerprint_cat64(h, v);
            }
          }
        }
      }
      uint64_t v = q.fp[fi];
      if (cross) v = fingerprint_cat64(h, imm ? (uint64_t)(int64_t)kb[kl] : v);
      out = (int32_t)barrett_mod(v, c.m, c.magic);
    } else if (c.kind == WD_FEAT_VOCAB) {
      const int32_t *tv = q.tok_val + kb[0];
      int32_t seen = -1;
      out = 0;
      for (int32_t i = 0; i < kr[0]; ++i) {          // the j-th token that is in the vocabulary
        const int32_t v = tv[i];
        if (v >= 0 && ++seen == (int32_t)j) { out = v; break; }
      }
    } else {
      out = kb[0];                                   // IDENTITY / BUCKET: one id, computed in the prologue
    }
    ids[p] = out;
  }
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

// ... of every token of the batch in ONE launch: the token's feature by its range [tok_base[f], tok_base[f] + tok_n[f]) (F <= a few
// dozen: a linear walk), the feature's vocabulary from a device table; features without a vocabulary_list column are skipped.
// ntok_dev (optional): the token count on the device (a captured featurizer).
__global__ void __launch_bounds__(256)
k_feat_vocab_all(const uint8_t *__restrict__ bytes, const int32_t *__restrict__ tok_offs, int64_t ntok,
                 const int32_t *__restrict__ ntok_dev, const int32_t *__restrict__ tok_base, const int32_t *__restrict__ tok_n,
                 int32_t F, const wd_feat_vocab_t *__restrict__ vt, int32_t *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (ntok_dev) ntok = *ntok_dev;
  if (t >= ntok) return;
  int f = -1;
  for (int g = 0; g < F; ++g) {
    const int32_t b0 = tok_base[g];
    if (t >= b0 && t < b0 + tok_n[g]) f = g;
  }
  if (f < 0 || vt[f].nvocab <= 0) return;
  const uint8_t *__restrict__ vbytes = vt[f].bytes;
  const int32_t *__restrict__ voffs = vt[f].offs;
  const int32_t nv = vt[f].nvocab;
  const int32_t o0 = tok_offs[t], len = tok_offs[t + 1] - o0;
  int32_t hit = -1;
  for (int32_t v = 0; v < nv && hit < 0; ++v) {
    const int32_t w0 = voffs[v];
    if (voffs[v + 1] - w0 != len) continue;
    bool eq = true;
    for (int32_t c = 0; c < len; ++c) eq = eq && bytes[o0 + c] == vbytes[w0 + c];
    if (eq) hit = v;
  }
  out[t] = hit;
}

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

extern "C" int wd_feat_vocab_lookup_all(const uint8_t *bytes, const int32_t *tok_offs, int64_t ntok, const int32_t *ntok_dev,
                                        const int32_t *tok_base, const int32_t *tok_n, int32_t nfeat,
                                        const wd_feat_vocab_t *vocab_table_dev, int32_t *tok_val, wd_stream_t stream) {
  if (ntok <= 0 || nfeat <= 0) return WD_OK;
  WD_REQUIRE(bytes && tok_offs && tok_base && tok_n && vocab_table_dev && tok_val, "null pointer");
  hipLaunchKernelGGL(k_feat_vocab_all, dim3((unsigned)wd::ceil_div(ntok, 256)), dim3(256), 0, wd::as_stream(stream), bytes, tok_offs,
                     ntok, ntok_dev, tok_base, tok_n, nfeat, vocab_table_dev, tok_val);
  return wd::check_launch("wd_feat_vocab_lookup_all");
}

extern "C" int wd_feat_lens(const wd_feat_slot_t *slots_dev, const wd_feat_batch_t *batch, int32_t *lens, int32_t *block_stats,
                            wd_stream_t stream) {
  WD_REQUIRE(slots_dev && feat_batch_ok(batch) && lens && block_stats, "null pointer / bad batch descriptor");
  if (batch->batch == 0) return WD_OK;
  hipLaunchKernelGGL(k_feat_lens, dim3((unsigned)wd::ceil_div(batch->batch * batch->S, 256)), dim3(256), 0,
                     wd::as_stream(stream), slots_dev, *batch, lens, block_stats);
  return wd::check_launch("wd_feat_lens");
}

static int feat_emit_ex_per_wg(int32_t S) {
  static const int env = getenv("WD_FEAT_EX") ? atoi(getenv("WD_FEAT_EX")) : 0;      // A/B runs
  const int want = env > 0 ? env : FE_EX;
  return S * want <= FE_MAX_PAIRS ? want : (FE_MAX_PAIRS / S > 0 ? FE_MAX_PAIRS / S : 1);
}

extern "C" int wd_feat_emit(const wd_feat_slot_t *slots_dev, const wd_feat_batch_t *batch, const int32_t *bag_offs,
                            int32_t *ids, int64_t ids_capacity, wd_stream_t stream) {
  WD_REQUIRE(slots_dev && feat_batch_ok(batch) && bag_offs && ids, "null pointer / bad batch descriptor");
  WD_REQUIRE(batch->S <= FE_MAX_PAIRS, "more than 1024 categorical columns");
  WD_REQUIRE(ids_capacity >= 0, "ids_capacity < 0");
  if (batch->batch == 0) return WD_OK;
  const int ex = feat_emit_ex_per_wg(batch->S);
  const int pairs = ex * batch->S;
  const size_t lds = ((size_t)(pairs + 1 + 3) / 4 * 4 + (size_t)3 * pairs * WD_MAX_CROSS_KEYS + pairs + 1) * 4 +
                     (size_t)batch->S * sizeof(FeCol);
  static size_t lds_allowed = 64 * 1024;       // (beyond ~600 columns the tables pass what a launch gets without the attribute)
  if (lds > lds_allowed) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_feat_emit_par), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds);
    WD_REQUIRE(e == hipSuccess, "hipFuncSetAttribute(k_feat_emit_par, dynamic LDS) failed");
    lds_allowed = lds;
  }
  hipLaunchKernelGGL(k_feat_emit_par, dim3((unsigned)wd::ceil_div(batch->batch, (int64_t)ex)), dim3(256), lds,
                     wd::as_stream(stream), slots_dev, *batch, bag_offs, ids, ex, ids_capacity);
  return wd::check_launch("wd_feat_emit");
}

// bag CSR from the lengths (+ the per-block sums wd_feat_lens left): offs[n] = number of ids of the batch
extern "C" int64_t wd_feat_offsets_workspace_bytes(int64_t n) { return 8 * wd::ceil_div(n > 0 ? n : 1, (int64_t)256); }

extern "C" int wd_feat_offsets(const int32_t *lens, const int32_t *block_stats, int64_t n, int32_t *offs, int64_t ids_capacity,
                               int32_t *flags, wd_stream_t stream) {
  WD_REQUIRE(offs && n >= 0, "null pointer");
  WD_REQUIRE(n == 0 || (lens && block_stats), "null pointer");
  const int64_t nblocks = n > 0 ? wd::ceil_div(n, (int64_t)256) : 0;
  hipLaunchKernelGGL(k_feat_offsets, dim3((unsigned)(nblocks > 0 ? nblocks : 1)), dim3(256), 0, wd::as_stream(stream), lens,
                     block_stats, n, (int32_t)nblocks, offs, ids_capacity, flags);
  return wd::check_launch("wd_feat_offsets");
}

extern "C" int wd_fingerprint64(const uint8_t *bytes, const int32_t *tok_offs, int64_t ntok, uint64_t *out_fp,
                                wd_stream_t stream) {
  if (ntok <= 0) return WD_OK;
  WD_REQUIRE(bytes && tok_offs && out_fp, "null pointer");
  hipLaunchKernelGGL(k_fingerprint64, dim3((unsigned)wd::ceil_div(ntok, 256)), dim3(256), 0, wd::as_stream(stream),
                     bytes, tok_offs, ntok, (const int32_t *)nullptr, out_fp);
  return wd::check_launch("wd_fingerprint64");
}

extern "C" int wd_fingerprint64_dyn(const uint8_t *bytes, const int32_t *tok_offs, int64_t ntok_capacity, const int32_t *ntok_dev,
                                    uint64_t *out_fp, wd_stream_t stream) {
  if (ntok_capacity <= 0) return WD_OK;
  WD_REQUIRE(bytes && tok_offs && out_fp && ntok_dev, "null pointer");
  hipLaunchKernelGGL(k_fingerprint64, dim3((unsigned)wd::ceil_div(ntok_capacity, 256)), dim3(256), 0, wd::as_stream(stream),
                     bytes, tok_offs, ntok_capacity, ntok_dev, out_fp);
  return wd::check_launch("wd_fingerprint64_dyn");
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
