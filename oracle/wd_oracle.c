/*
 * oracle/wd_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the TensorFlow 1.x semantics that the reference
 * (Lapis-Hong/wide_deep) selects for its train-step hot path.  Nothing under
 * wide_deep_amd/ may import, link or call this file; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * The arithmetic lives in a third-party dependency that is NOT vendored under
 * /root/reference: TensorFlow (requirements.txt:1, "tensorflow >=1.4",
 * unpinned).  The reference only wires TF ops together:
 *   - string hashing      python/lib/build_estimator.py:86-88   (categorical_column_with_hash_bucket
 *                                                                 -> string_to_hash_bucket_fast -> FarmHash Fingerprint64)
 *   - feature crossing    python/lib/build_estimator.py:138-155 (crossed_column -> SparseCross hashed, FingerprintCat64)
 *   - wide logits         python/lib/linear.py:29-36            (linear_model, sparse_combiner='sum')
 *   - embedding bag       python/lib/dnn.py:88-90               (input_layer -> safe_embedding_lookup_sparse, combiner='mean')
 *   - head                python/lib/joint.py:264-269           (sigmoid CE, SUM over batch, weight column)
 *   - optimizers          python/lib/utils/model_util.py:84-90  (Adagrad / Ftrl, TF defaults), conf/model.yaml:14,47-48
 *
 * PARITY PINNING: the reference's own tests pin no numbers at this boundary
 * (python/lib/wide_deep_test.py:80-85 only checks loss-down/auc-up).  The
 * integer functions are pinned against upstream-TF known answers
 * (tests/golden/kat_hash.json; Fingerprint64 of strings <= 16 bytes,
 * hash-bucket ids, hashed crosses).  Fingerprint64 for strings > 16 bytes is
 * pinned by published known answers of the same function outside TF (Guava's
 * FarmHashFingerprint64Test: 32- and 256-byte strings and the 3200-message
 * chain over lengths 0..3199; tests/helpers.py) and, up to 32 bytes, by a
 * compiled CityHash64 (tests/golden/kat_city_le32.json); a second independent
 * transcription (oracle/farmhash_py.py) is cross-checked as well.  The fp32
 * semantics stay "parity unpinned" against TF itself.
 *
 * Build: oracle/build.sh  (gcc -O2 -fopenmp -shared -fPIC) -> oracle/_build/libwd_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* FarmHash farmhashna::Hash64 == TF Fingerprint64 (little-endian fetches)    */
/* ------------------------------------------------------------------------- */
#define K0 0xc3a5c85c97cb3127ULL
#define K1 0xb492b66fbe98f273ULL
#define K2 0x9ae16a3b2f90404fULL

static inline uint64_t fetch64(const uint8_t *p) { uint64_t r; memcpy(&r, p, 8); return r; }
static inline uint64_t fetch32(const uint8_t *p) { uint32_t r; memcpy(&r, p, 4); return (uint64_t)r; }
static inline uint64_t rot64(uint64_t v, int s) { return s == 0 ? v : ((v >> s) | (v << (64 - s))); }
static inline uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

static inline uint64_t hash_len16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul;
  a ^= (a >> 47);
  uint64_t b = (v ^ a) * mul;
  b ^= (b >> 47);
  b *= mul;
  return b;
}

static uint64_t hash_len_0_16(const uint8_t *s, size_t len) {
  if (len >= 8) {
    uint64_t mul = K2 + len * 2;
    uint64_t a = fetch64(s) + K2;
    uint64_t b = fetch64(s + len - 8);
    uint64_t c = rot64(b, 37) * mul + a;
    uint64_t d = (rot64(a, 25) + b) * mul;
    return hash_len16(c, d, mul);
  }
  if (len >= 4) {
    uint64_t mul = K2 + len * 2;
    uint64_t a = fetch32(s);
    return hash_len16(len + (a << 3), fetch32(s + len - 4), mul);
  }
  if (len > 0) {
    uint8_t a = s[0];
    uint8_t b = s[len >> 1];
    uint8_t c = s[len - 1];
    uint32_t y = (uint32_t)a + ((uint32_t)b << 8);
    uint32_t z = (uint32_t)len + ((uint32_t)c << 2);
    return shift_mix(y * K2 ^ z * K0) * K2;
  }
  return K2;
}

static uint64_t hash_len_17_32(const uint8_t *s, size_t len) {
  uint64_t mul = K2 + len * 2;
  uint64_t a = fetch64(s) * K1;
  uint64_t b = fetch64(s + 8);
  uint64_t c = fetch64(s + len - 8) * mul;
  uint64_t d = fetch64(s + len - 16) * K2;
  return hash_len16(rot64(a + b, 43) + rot64(c, 30) + d, a + rot64(b + K2, 18) + c, mul);
}

static uint64_t hash_len_33_64(const uint8_t *s, size_t len) {
  uint64_t mul = K2 + len * 2;
  uint64_t a = fetch64(s) * K2;
  uint64_t b = fetch64(s + 8);
  uint64_t c = fetch64(s + len - 8) * mul;
  uint64_t d = fetch64(s + len - 16) * K2;
  uint64_t y = rot64(a + b, 43) + rot64(c, 30) + d;
  uint64_t z = hash_len16(y, a + rot64(b + K2, 18) + c, mul);
  uint64_t e = fetch64(s + 16) * mul;
  uint64_t f = fetch64(s + 24);
  uint64_t g = (y + fetch64(s + len - 32)) * mul;
  uint64_t h = (z + fetch64(s + len - 24)) * mul;
  return hash_len16(rot64(e + f, 43) + rot64(g, 30) + h, e + rot64(f + a, 18) + g, mul);
}

typedef struct { uint64_t first, second; } u128;

static inline u128 weak_hash_32_seeds(const uint8_t *s, uint64_t a, uint64_t b) {
  uint64_t w = fetch64(s), x = fetch64(s + 8), y = fetch64(s + 16), z = fetch64(s + 24);
  a += w;
  b = rot64(b + a + z, 21);
  uint64_t c = a;
  a += x;
  a += y;
  b += rot64(a, 44);
  u128 r = { a + z, b + c };
  return r;
}

uint64_t wdo_fingerprint64(const uint8_t *s, size_t len) {
  if (len <= 32) return len <= 16 ? hash_len_0_16(s, len) : hash_len_17_32(s, len);
  if (len <= 64) return hash_len_33_64(s, len);
  const uint64_t seed = 81;
  uint64_t x = seed;
  uint64_t y = seed * K1 + 113;
  uint64_t z = shift_mix(y * K2 + 113) * K2;
  u128 v = { 0, 0 }, w = { 0, 0 };
  x = x * K2 + fetch64(s);
  const uint8_t *end = s + ((len - 1) / 64) * 64;
  const uint8_t *last64 = end + ((len - 1) & 63) - 63;
  do {
    x = rot64(x + y + v.first + fetch64(s + 8), 37) * K1;
    y = rot64(y + v.second + fetch64(s + 48), 42) * K1;
    x ^= w.second;
    y += v.first + fetch64(s + 40);
    z = rot64(z + w.first, 33) * K1;
    v = weak_hash_32_seeds(s, v.second * K1, x + w.first);
    w = weak_hash_32_seeds(s + 32, z + w.second, y + fetch64(s + 16));
    uint64_t t = z; z = x; x = t;
    s += 64;
  } while (s != end);
  uint64_t mul = K1 + ((z & 0xff) << 1);
  s = last64;
  w.first += ((len - 1) & 63);
  v.first += w.first;
  w.first += v.first;
  x = rot64(x + y + v.first + fetch64(s + 8), 37) * mul;
  y = rot64(y + v.second + fetch64(s + 48), 42) * mul;
  x ^= w.second * 9;
  y += v.first * 9 + fetch64(s + 40);
  z = rot64(z + w.first, 33) * mul;
  v = weak_hash_32_seeds(s, v.second * mul, x + w.first);
  w = weak_hash_32_seeds(s + 32, z + w.second, y + fetch64(s + 16));
  { uint64_t t = z; z = x; x = t; }
  return hash_len16(hash_len16(v.first, w.first, mul) + shift_mix(y) * K0 + z,
                    hash_len16(v.second, w.second, mul) + x, mul);
}

/* TF FingerprintCat64 (tensorflow/core/platform/fingerprint.h) */
uint64_t wdo_fingerprint_cat64(uint64_t fp1, uint64_t fp2) {
  const uint64_t kMul = 0xc6a4a7935bd1e995ULL;
  uint64_t r = fp1 ^ kMul;
  r ^= shift_mix(fp2 * kMul) * kMul;
  r *= kMul;
  r = shift_mix(r) * kMul;
  r = shift_mix(r);
  return r;
}

/* packed tokens: bytes + offs[n+1] (int64) -> fp[n] */
void wdo_fingerprint64_batch(const uint8_t *bytes, const int64_t *offs, int64_t n, uint64_t *out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) out[i] = wdo_fingerprint64(bytes + offs[i], (size_t)(offs[i + 1] - offs[i]));
}

/* string_to_hash_bucket_fast: id = Fingerprint64(token) % num_buckets */
void wdo_hash_bucket_batch(const uint8_t *bytes, const int64_t *offs, int64_t n, uint64_t num_buckets, int64_t *out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    out[i] = (int64_t)(wdo_fingerprint64(bytes + offs[i], (size_t)(offs[i + 1] - offs[i])) % num_buckets);
}

/* ------------------------------------------------------------------------- */
/* SparseCross (hashed).  Each key column k: vals[k] (uint64: Fingerprint64 of */
/* the string, or the int64 id itself) + CSR offs[k][batch+1].                */
/* Cartesian product per example, LAST key varying fastest.                   */
/* ------------------------------------------------------------------------- */
void wdo_cross_offsets(const int32_t *const *offs, int nkeys, int64_t batch, int32_t *out_offs) {
  out_offs[0] = 0;
  for (int64_t b = 0; b < batch; ++b) {
    int64_t prod = 1;
    for (int k = 0; k < nkeys; ++k) prod *= (int64_t)(offs[k][b + 1] - offs[k][b]);
    out_offs[b + 1] = out_offs[b] + (int32_t)prod;
  }
}

void wdo_cross_hash(const uint64_t *const *vals, const int32_t *const *offs, int nkeys, int64_t batch,
                    uint64_t hash_key, uint64_t num_buckets, const int32_t *out_offs, int64_t *out_ids) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < batch; ++b) {
    int64_t n = out_offs[b + 1] - out_offs[b];
    int32_t cnt[16];
    for (int k = 0; k < nkeys; ++k) cnt[k] = offs[k][b + 1] - offs[k][b];
    for (int64_t j = 0; j < n; ++j) {
      /* decode j as mixed radix, last key fastest */
      int32_t idx[16];
      int64_t r = j;
      for (int k = nkeys - 1; k >= 0; --k) { idx[k] = (int32_t)(r % cnt[k]); r /= cnt[k]; }
      uint64_t h = hash_key;
      for (int k = 0; k < nkeys; ++k) h = wdo_fingerprint_cat64(h, vals[k][offs[k][b] + idx[k]]);
      uint64_t m = num_buckets > 0 ? num_buckets : (uint64_t)INT64_MAX;
      out_ids[out_offs[b] + j] = (int64_t)(h % m);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* safe_embedding_lookup_sparse: combiner mean (embedding) / sum (linear).    */
/* One column: table [V, D]; CSR ids/offs over nbags examples.                */
/* ids < 0 are pruned (vocab OOV), empty rows -> zero vector.                 */
/* ------------------------------------------------------------------------- */
void wdo_embag_fwd(const float *table, int64_t D, const int64_t *ids, const int32_t *offs, int64_t nbags,
                   int mean, float *out, int64_t ld_out) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < nbags; ++b) {
    float *o = out + b * ld_out;
    for (int64_t d = 0; d < D; ++d) o[d] = 0.f;
    int32_t cnt = 0;
    for (int32_t j = offs[b]; j < offs[b + 1]; ++j) {
      if (ids[j] < 0) continue;
      const float *row = table + ids[j] * D;
      for (int64_t d = 0; d < D; ++d) o[d] += row[d];
      ++cnt;
    }
    if (mean && cnt > 1) {
      float c = (float)cnt;
      for (int64_t d = 0; d < D; ++d) o[d] = o[d] / c;
    }
  }
}

/* Row gradient of the lookup: IndexedSlices with duplicates summed
 * (unique -> gather -> sparse_segment_{mean,sum}); rows are visited in
 * ascending id, contributions summed in ascending bag order. */
typedef struct { int64_t id; int32_t bag; float scale; } occ_t;
static int occ_cmp(const void *a, const void *b) {
  const occ_t *x = (const occ_t *)a, *y = (const occ_t *)b;
  if (x->id != y->id) return x->id < y->id ? -1 : 1;
  return (x->bag > y->bag) - (x->bag < y->bag);
}

/* returns number of unique rows; caller passes uniq_ids[nnz], row_grad[nnz*D] */
int64_t wdo_embag_row_grads(int64_t D, const int64_t *ids, const int32_t *offs, int64_t nbags, int mean,
                            const float *grad_out, int64_t ld_grad, int64_t *uniq_ids, float *row_grad) {
  int64_t nnz = offs[nbags];
  occ_t *occ = (occ_t *)malloc(sizeof(occ_t) * (size_t)(nnz > 0 ? nnz : 1));
  int64_t n = 0;
  for (int64_t b = 0; b < nbags; ++b) {
    int32_t cnt = 0;
    for (int32_t j = offs[b]; j < offs[b + 1]; ++j) cnt += ids[j] >= 0;
    float scale = (mean && cnt > 1) ? 1.0f / (float)cnt : 1.0f;
    for (int32_t j = offs[b]; j < offs[b + 1]; ++j)
      if (ids[j] >= 0) { occ[n].id = ids[j]; occ[n].bag = (int32_t)b; occ[n].scale = scale; ++n; }
  }
  qsort(occ, (size_t)n, sizeof(occ_t), occ_cmp);
  int64_t u = -1;
  for (int64_t i = 0; i < n; ++i) {
    if (i == 0 || occ[i].id != occ[i - 1].id) {
      ++u;
      uniq_ids[u] = occ[i].id;
      for (int64_t d = 0; d < D; ++d) row_grad[u * D + d] = 0.f;
    }
    const float *g = grad_out + (int64_t)occ[i].bag * ld_grad;
    for (int64_t d = 0; d < D; ++d) row_grad[u * D + d] += g[d] * occ[i].scale;
  }
  free(occ);
  return u + 1;
}

/* tf.train.AdagradOptimizer sparse apply: acc += g*g; var -= lr * g / sqrt(acc) */
void wdo_adagrad_rows(float *table, float *accum, int64_t D, const int64_t *uniq_ids, int64_t nuniq,
                      const float *row_grad, float lr) {
#pragma omp parallel for schedule(static)
  for (int64_t u = 0; u < nuniq; ++u) {
    float *w = table + uniq_ids[u] * D, *a = accum + uniq_ids[u] * D;
    const float *g = row_grad + u * D;
    for (int64_t d = 0; d < D; ++d) {
      a[d] += g[d] * g[d];
      w[d] -= lr * g[d] / sqrtf(a[d]);
    }
  }
}

void wdo_adagrad_dense(float *w, float *accum, const float *g, int64_t n, float lr) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    accum[i] += g[i] * g[i];
    w[i] -= lr * g[i] / sqrtf(accum[i]);
  }
}

/* tf.train.FtrlOptimizer (lr_power = -0.5, l2_shrinkage = 0), TF training_ops.cc FtrlCompute */
static inline void ftrl_one(float *w, float *z, float *n, float g, float lr, float l1, float l2) {
  float n_new = *n + g * g;
  *z += g - (sqrtf(n_new) - sqrtf(*n)) / lr * (*w);
  float quad = sqrtf(n_new) / lr + 2.0f * l2;
  float sgn = (*z > 0.f) ? 1.f : ((*z < 0.f) ? -1.f : 0.f);
  float pre = (sgn * l1 - *z) / quad;
  *w = fabsf(*z) > l1 ? pre : 0.f;
  *n = n_new;
}

void wdo_ftrl_rows(float *w, float *z, float *n, int64_t D, const int64_t *uniq_ids, int64_t nuniq,
                   const float *row_grad, float lr, float l1, float l2) {
#pragma omp parallel for schedule(static)
  for (int64_t u = 0; u < nuniq; ++u)
    for (int64_t d = 0; d < D; ++d) {
      int64_t i = uniq_ids[u] * D + d;
      ftrl_one(w + i, z + i, n + i, row_grad[u * D + d], lr, l1, l2);
    }
}

void wdo_ftrl_dense(float *w, float *z, float *n, const float *g, int64_t cnt, float lr, float l1, float l2) {
  for (int64_t i = 0; i < cnt; ++i) ftrl_one(w + i, z + i, n + i, g[i], lr, l1, l2);
}

/* ------------------------------------------------------------------------- */
/* All sparse columns of one step in ONE call, columns side by side (what TF's */
/* inter-op pool does with the per-column subgraphs).  Every column runs the   */
/* per-column functions above unchanged, so results are bit-identical to the   */
/* per-column calls; used by bench.py's cpu_baseline leg.                      */
/* ------------------------------------------------------------------------- */
void wdo_embag_fwd_cols(int ncols, const float *const *tables, const int64_t *D, const int64_t *const *ids,
                        const int32_t *const *offs, int64_t nbags, int mean, float *const *outs,
                        const int64_t *ld_out) {
  const int64_t chunk = 256;
  const int64_t nchunk = (nbags + chunk - 1) / chunk;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t job = 0; job < (int64_t)ncols * nchunk; ++job) {
    const int c = (int)(job / nchunk);
    const int64_t b0 = (job % nchunk) * chunk, b1 = b0 + chunk < nbags ? b0 + chunk : nbags;
    const int64_t d = D[c];
    for (int64_t b = b0; b < b1; ++b) {
      float *o = outs[c] + b * ld_out[c];
      for (int64_t k = 0; k < d; ++k) o[k] = 0.f;
      int32_t cnt = 0;
      for (int32_t j = offs[c][b]; j < offs[c][b + 1]; ++j) {
        if (ids[c][j] < 0) continue;
        const float *row = tables[c] + ids[c][j] * d;
        for (int64_t k = 0; k < d; ++k) o[k] += row[k];
        ++cnt;
      }
      if (mean && cnt > 1) {
        float cf = (float)cnt;
        for (int64_t k = 0; k < d; ++k) o[k] = o[k] / cf;
      }
    }
  }
}

/* kind 0: Adagrad rows (slot_b = accumulator); kind 1: Ftrl rows (slot_a = z / linear, slot_b = n / accumulator) */
void wdo_sparse_apply_cols(int ncols, const int64_t *D, const int64_t *const *ids, const int32_t *const *offs,
                           int64_t nbags, int mean, const float *const *grads, const int64_t *ld_grad,
                           float *const *tables, float *const *slot_a, float *const *slot_b, int kind, float lr,
                           float l1, float l2) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < ncols; ++c) {
    const int64_t nnz = offs[c][nbags];
    int64_t *uniq = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
    float *rg = (float *)malloc(sizeof(float) * (size_t)((nnz > 0 ? nnz : 1) * D[c]));
    const int64_t nu = wdo_embag_row_grads(D[c], ids[c], offs[c], nbags, mean, grads[c], ld_grad[c], uniq, rg);
    if (kind == 0) wdo_adagrad_rows(tables[c], slot_b[c], D[c], uniq, nu, rg, lr);
    else wdo_ftrl_rows(tables[c], slot_a[c], slot_b[c], D[c], uniq, nu, rg, lr, l1, l2);
    free(uniq);
    free(rg);
  }
}

/* ------------------------------------------------------------------------- */
/* Head: sigmoid_cross_entropy_with_logits, loss SUM over batch, weights.     */
/* loss_b = max(x,0) - x*y + log1p(exp(-|x|)); dloss/dx = w*(sigmoid(x)-y)    */
/* ------------------------------------------------------------------------- */
double wdo_bce_sum(const float *logits, const float *labels, const float *weights, int64_t n, float *dlogits,
                   float *prob) {
  double loss = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    float x = logits[i], y = labels[i], w = weights ? weights[i] : 1.0f;
    float l = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    float p = x >= 0.f ? 1.0f / (1.0f + expf(-x)) : expf(x) / (1.0f + expf(x));
    loss += (double)(w * l);
    if (dlogits) dlogits[i] = w * (p - y);
    if (prob) prob[i] = p;
  }
  return loss;
}

int wdo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void wdo_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
