// wide_deep_amd/csrc/sparse_fused.hip -- sparse backward of the train step in FOUR short launches.
//
// Same contract as sparse_update.hip (python/lib/joint.py:224-262 with tf.train.AdagradOptimizer on the
// embedding rows and tf.train.FtrlOptimizer on the wide rows; duplicate rows of the IndexedSlices gradient
// are summed BEFORE the optimizer formula is applied once per unique row), but without a
// device-wide sort: a device radix/merge sort of the ~2e5 occurrences of a batch costs 15-17 dependent
// launches (~100-150 us on MI355X) for 1.7 MB of data.  Instead:
//
//   K1 k_bucket_hist     occurrence -> bucket = slot.bucket_base + (id >> slot.bucket_shift); each workgroup histograms
//                        its chunk of bags in LDS (device-scope returned atomics on ~1.6k counters measured 35 us for
//                        2e5 occurrences) and writes its row of the count matrix; ranks of one-row buckets are
//                        deterministic (wave ballots), the others are whatever the LDS atomics return
//   K1b k_bucket_colscan per bucket: exclusive prefix over the chunks (column scan) + bucket totals
//   K2 k_bucket_scatter  every block scans the totals in LDS (nb <= 8192) and scatters (key << 32 | bag) pairs to
//                        start[bucket] + chunk_prefix + rank: all occurrences of a row range are now contiguous;
//                        block 0 also lists the buckets largest-first (launch order of K3: Zipf head rows)
//   K3 k_bucket_update   one workgroup per bucket: sort of the 64-bit pairs in LDS (rank sort up to 512 pairs,
//                        bitonic up to 1024 pairs; larger buckets sort in place in HBM/L2; one-row buckets need no
//                        sort), then per unique row the gradient is reduced in ascending bag order and the optimizer
//                        (Adagrad / FTRL specialised; any of wd_opt_t in the GEN instantiation) is applied in the same
//                        kernel; the extra last workgroup does bias_weights.
//
// The arrival order inside a bucket (atomics) is arbitrary, but the full sort on (row, bag) makes the
// summation order -- and therefore every updated bit -- independent of it: the step stays deterministic.
// Rows hit more than 32 times (Zipf heads, "missing value" tokens) are reduced cooperatively by the whole
// workgroup (64 lane groups in parallel + a fixed-shape LDS tree) instead of serially by one lane group.
#include "common.h"

namespace {

constexpr int CAP_LDS = 1024;   // pairs sorted in LDS per workgroup (8 KB: keeps 8 workgroups resident per CU)
constexpr int LONG_SEG = 32;    // segments longer than this are reduced by the whole workgroup
constexpr int MAX_LONG = 128;   // long-segment list per bucket
constexpr int MAX_NB = 8192;    // buckets (LDS scan array of K2)

constexpr int RANK_MAX = 512;   // buckets up to this size are rank-sorted (O(m^2 / 256) per lane, 2 barriers)
constexpr int MAX_SLOTS_LDS = 96;  // slot descriptors cached in LDS by k_bucket_update (more slots: read from HBM)
constexpr int MAX_CHUNKS = 512; // workgroups of K1 / K2 (rows of the count matrix); 128 until round 5: configs[3]'s 1.06 M occurrences
                                // were 8.3 k per workgroup, one bag after the other per lane -- histogram 35 us, scatter 49 us alone

// chunk c = bags [c*bags_per_chunk, (c+1)*bags_per_chunk); rank = position of an occurrence among the chunk's
// occurrences of the same bucket.
//   * buckets that get sorted afterwards (bucket_shift > 0): LDS atomics, arrival order is irrelevant.
//   * one-row buckets (bucket_shift == 0) are NOT sorted by k_bucket_update, so their order must be reproducible:
//     occurrences are taken in rounds (256 bags x position-in-bag); inside a round the lanes of a wavefront that share
//     a bucket find each other with ballots over the bucket bits (a match-any), and the four wavefronts append in
//     turn -- a bucket's content ends up in ascending bag order whatever the hardware's atomics order is.
__global__ void __launch_bounds__(256)
k_bucket_hist(const wd_slot_t *__restrict__ slots, int32_t S, const int32_t *__restrict__ ids,
              const int32_t *__restrict__ bag_offs, int64_t nbags, int64_t bags_per_chunk, int32_t nb,
              int32_t *__restrict__ cntm, int32_t *__restrict__ rank) {
  WD_SIDE_PRIO();
  __shared__ int32_t hist[MAX_NB];
  __shared__ int32_t has_stable, any_left;
  const int t = threadIdx.x;
  if (t == 0) has_stable = 0;
  for (int i = t; i < nb; i += 256) hist[i] = 0;
  __syncthreads();
  for (int i = t; i < S; i += 256)
    if (slots[i].bucket_shift == 0 && !(slots[i].flags & WD_SLOT_F_SMALL)) has_stable = 1;
  __syncthreads();
  const int64_t b0 = (int64_t)blockIdx.x * bags_per_chunk;
  const int64_t b1 = b0 + bags_per_chunk < nbags ? b0 + bags_per_chunk : nbags;
  if (!has_stable) {
    // a lane walks its bag EIGHT ids at a time: the loads of a round in flight together, then the LDS atomics, then the ranks (one id
    // after the other was a chain of load -> atomic -> store per id: 40 us for the 1.06 M occurrences of a configs[3] batch)
    for (int64_t bag = b0 + t; bag < b1; bag += 256) {
      const int si = (int)((uint32_t)bag % (uint32_t)S);       // (nbags < 2^31: wd_sparse_bucketize)
      const int32_t flags = slots[si].flags;
      if (flags & WD_SLOT_F_SMALL) continue;      // csrc/small_tables.hip updates this column
      const int32_t bbase = slots[si].bucket_base, bshift = slots[si].bucket_shift;
      const int32_t j0 = bag_offs[bag], j1 = bag_offs[bag + 1];
      for (int32_t j = j0; j < j1; j += 8) {
        int32_t idv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) idv[u] = j + u < j1 ? ids[j + u] : -1;
        int32_t rk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)       // (id < 0: padding entry of a fixed-capacity exchange segment)
          if (idv[u] >= 0) rk[u] = atomicAdd(&hist[bbase + (idv[u] >> bshift)], 1);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (idv[u] >= 0) rank[j + u] = rk[u];
      }
    }
  } else {
    const int bits = 32 - __builtin_clz((unsigned)(nb > 1 ? nb - 1 : 1));
    const int lane = t & 63, wave = t >> 6;
    for (int64_t base = b0; base < b1; base += 256) {
      const int64_t bag = base + t;
      int32_t j0 = 0, j1 = 0, bshift = 0, bbase = 0;
      if (bag < b1) {
        const wd_slot_t sl = slots[bag % S];
        j0 = bag_offs[bag];
        j1 = (sl.flags & WD_SLOT_F_SMALL) ? j0 : bag_offs[bag + 1];      // (csrc/small_tables.hip updates that column)
        bshift = sl.bucket_shift;
        bbase = sl.bucket_base;
      }
      for (int32_t k = 0;; ++k) {
        if (t == 0) any_left = 0;
        __syncthreads();
        int32_t bk = -1;
        if (j0 + k < j1) {
          const int32_t id = ids[j0 + k];
          if (id >= 0) bk = bbase + (id >> bshift);
          any_left = 1;
        }
        __syncthreads();
        if (!any_left) break;
        const bool st = bk >= 0 && bshift == 0;
        if (bk >= 0 && !st) rank[j0 + k] = atomicAdd(&hist[bk], 1);
        unsigned long long same = __ballot(st);
        const int32_t kb = st ? bk : 0;
        for (int bit = 0; bit < bits; ++bit) {
          const bool one = (kb >> bit) & 1;
          const unsigned long long v = __ballot(one);
          same &= one ? v : ~v;
        }
        const int lower = __popcll(same & ((1ull << lane) - 1ull));
        const int cnt = __popcll(same);
        for (int w = 0; w < 4; ++w) {
          if (wave == w && st) {
            const int32_t r = hist[bk] + lower;   // every lane of the wavefront reads before its last lane writes
            rank[j0 + k] = r;
            if (lower == cnt - 1) hist[bk] = r + 1;
          }
          __syncthreads();
        }
      }
    }
  }
  __syncthreads();
  int32_t *row = cntm + (int64_t)blockIdx.x * nb;
  for (int i = t; i < nb; i += 256) row[i] = hist[i];
}

// cpre[c][b] = sum_{c' < c} cntm[c'][b];  total[b] = column sum.  A workgroup takes 64 buckets x G groups of 32 chunks
// (G = ceil(nchunks / 32) <= 16 wavefronts): every count is loaded in ONE round of 32 loads per lane, the groups' sums meet in
// LDS.  (Round 1: thread per bucket, in place -- one dependent L2 round trip per chunk, 30 us for 104 chunks; rounds 2-4: thread
// per bucket, 32 loads deep -- one round per 32 chunks, 13 us for 128 chunks.)
constexpr int SCAN_DEPTH = 32;
constexpr int SCAN_GROUPS = MAX_CHUNKS / SCAN_DEPTH;
__global__ void __launch_bounds__(64 * SCAN_GROUPS)
k_bucket_colscan(const int32_t *__restrict__ cntm, int32_t *__restrict__ cpre, int32_t nchunks, int32_t nb,
                 int32_t *__restrict__ total) {
  WD_SIDE_PRIO();
  __shared__ int32_t gsum[SCAN_GROUPS][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, G = blockDim.x >> 6;
  const int b = blockIdx.x * 64 + lane;
  const int c0 = g * SCAN_DEPTH;
  int32_t v[SCAN_DEPTH];
#pragma unroll
  for (int q = 0; q < SCAN_DEPTH; ++q) v[q] = (b < nb && c0 + q < nchunks) ? cntm[(int64_t)(c0 + q) * nb + b] : 0;
  int32_t run = 0;
#pragma unroll
  for (int q = 0; q < SCAN_DEPTH; ++q) {
    const int32_t x = v[q];
    v[q] = run;
    run += x;
  }
  gsum[g][lane] = run;
  __syncthreads();
  int32_t base = 0;
  for (int h = 0; h < g; ++h) base += gsum[h][lane];
  if (b >= nb) return;
#pragma unroll
  for (int q = 0; q < SCAN_DEPTH; ++q)
    if (c0 + q < nchunks) cpre[(int64_t)(c0 + q) * nb + b] = base + v[q];
  if (g == G - 1) total[b] = base + run;
}

__global__ void __launch_bounds__(256)
k_bucket_scatter(const wd_slot_t *__restrict__ slots, int32_t S, const int32_t *__restrict__ ids,
                 const int32_t *__restrict__ bag_offs, int64_t nbags, int64_t bags_per_chunk,
                 int32_t nb, const int32_t *__restrict__ total, const int32_t *__restrict__ cntm,
                 const int32_t *__restrict__ rank, int32_t *__restrict__ start, uint64_t *__restrict__ pairs) {
  WD_SIDE_PRIO();
  __shared__ int32_t sstart[MAX_NB];
  __shared__ int32_t wsum[4];
  // exclusive scan of total[0..nb): thread t owns E consecutive counters
  const int t = threadIdx.x;
  const int E = (nb + 255) / 256;
  int32_t local = 0;
  for (int e = 0; e < E; ++e) {
    const int i = t * E + e;
    const int32_t c = i < nb ? total[i] : 0;
    if (i < nb) sstart[i] = local;  // exclusive within the thread's run
    local += c;
  }
  int32_t incl = local;
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t v = __shfl_up(incl, off, 64);
    if ((t & 63) >= off) incl += v;
  }
  if ((t & 63) == 63) wsum[t >> 6] = incl;
  __syncthreads();
  int32_t wbase = 0;
  for (int w = 0; w < (t >> 6); ++w) wbase += wsum[w];
  const int32_t excl = wbase + incl - local;
  const int32_t *row = cntm + (int64_t)blockIdx.x * nb;
  for (int e = 0; e < E; ++e) {
    const int i = t * E + e;
    if (i < nb) {
      const int32_t st = sstart[i] + excl;
      if (blockIdx.x == 0) start[i] = st;
      sstart[i] = st + row[i];  // + what earlier chunks put into this bucket
    }
  }
  if (blockIdx.x == 0 && t == 255) start[nb] = excl + local;
  if (blockIdx.x == 0) {
    // Launch order of k_bucket_update, largest buckets first (start[nb+2 ..]): a bucket holding a Zipf head row takes
    // several times longer than the rest; started last it was the tail of the kernel (83 us instead of 44 at Zipf
    // 1.05).  Buckets are binned by floor(log2(size)); the order inside a bin is whatever the LDS atomics give --
    // it only schedules workgroups, no result depends on it.
    __shared__ int32_t cls_cnt[33], cls_pos[33];
    if (t < 33) cls_cnt[t] = 0;
    __syncthreads();
    for (int i = t; i < nb; i += 256) {
      const int32_t c = total[i];
      atomicAdd(&cls_cnt[c > 0 ? 32 - __builtin_clz((unsigned)c) : 0], 1);
    }
    __syncthreads();
    if (t == 0) {
      int32_t run = 0;
      for (int c = 32; c >= 0; --c) {
        cls_pos[c] = run;
        run += cls_cnt[c];
      }
    }
    __syncthreads();
    int32_t *order = start + nb + 2;
    for (int i = t; i < nb; i += 256) {
      const int32_t c = total[i];
      order[atomicAdd(&cls_pos[c > 0 ? 32 - __builtin_clz((unsigned)c) : 0], 1)] = i;
    }
  }
  __syncthreads();
  const int64_t b0 = (int64_t)blockIdx.x * bags_per_chunk;
  const int64_t b1 = b0 + bags_per_chunk < nbags ? b0 + bags_per_chunk : nbags;
  for (int64_t bag = b0 + t; bag < b1; bag += 256) {
    const int si = (int)((uint32_t)bag % (uint32_t)S);
    if (slots[si].flags & WD_SLOT_F_SMALL) continue;
    const int32_t bbase = slots[si].bucket_base, bshift = slots[si].bucket_shift;
    const uint32_t rbase = (uint32_t)slots[si].row_base;
    const int32_t j0 = bag_offs[bag], j1 = bag_offs[bag + 1];
    for (int32_t j = j0; j < j1; j += 8) {      // (eight ids and ranks in flight: k_bucket_hist)
      int32_t idv[8], rk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        idv[u] = j + u < j1 ? ids[j + u] : -1;
        rk[u] = j + u < j1 ? rank[j + u] : 0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (idv[u] >= 0)
          pairs[sstart[bbase + (idv[u] >> bshift)] + rk[u]] = ((uint64_t)(rbase + (uint32_t)idv[u]) << 32) | (uint32_t)bag;
    }
  }
}

__device__ __forceinline__ void ftrl_update(float &w, float &z, float &n, float g, float lr, float l1, float l2) {
  const float n_new = n + g * g;
  z += g - (sqrtf(n_new) - sqrtf(n)) / lr * w;
  const float quad = sqrtf(n_new) / lr + 2.0f * l2;
  const float sgn = z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f);
  const float pre = (sgn * l1 - z) / quad;
  w = fabsf(z) > l1 ? pre : 0.f;
  n = n_new;
}

__device__ __forceinline__ float4 adagrad4(float4 &a, float4 w, float4 g, float lr) {
  a.x += g.x * g.x; a.y += g.y * g.y; a.z += g.z * g.z; a.w += g.w * g.w;
  w.x -= lr * g.x / sqrtf(a.x);
  w.y -= lr * g.y / sqrtf(a.y);
  w.z -= lr * g.z / sqrtf(a.z);
  w.w -= lr * g.w / sqrtf(a.w);
  return w;
}

// Any tf.train optimizer the reference accepts (python/lib/utils/model_util.py:84-90), one parameter at a time.
// Slots: a / b of include/wd_hip.h (wd_opt_t): Adagrad b = accumulator; Ftrl a = linear (z), b = accumulator (n);
// RMSProp a = rms, b = momentum (centered: c = mean gradient); Adam a = m, b = v.
struct OptK {
  int32_t kind;
  float lr, p0, p1, p2, p3;
  float b1p, b2p;   // Adam: beta1^t, beta2^t of this step
};

__device__ __forceinline__ void ftrl_update(float &w, float &z, float &n, float g, float lr, float l1, float l2);

// FtrlOptimizer with learning_rate_power != -0.5 (TF training_ops.cc FtrlCompute, general branch): accum^(-lr_power) in
// place of sqrt(accum)
// l2_shrink != 0 (FtrlOptimizer(l2_shrinkage_regularization_strength=...)): the LINEAR slot accumulates the gradient of the
// shrinkage-penalised loss, g + 2 l2_shrink var, the ACCUMULATOR the plain g^2 (TF training_ops.cc FtrlCompute; the constants of
// ftrl_test.testFtrlWithL1_L2_L2Shrinkage pin exactly this form: tests/golden/kat_tf_fp32.json)
__device__ __forceinline__ void ftrl_update_pow(float &w, float &z, float &n, float g, float lr, float l1, float l2,
                                                float lr_power, float l2_shrink) {
  const float n_new = n + g * g;
  const float pn = lr_power == -0.5f ? sqrtf(n_new) : powf(n_new, -lr_power);
  const float po = lr_power == -0.5f ? sqrtf(n) : powf(n, -lr_power);
  z += (g + 2.0f * l2_shrink * w) - (pn - po) / lr * w;
  const float quad = pn / lr + 2.0f * l2;
  const float sgn = z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f);
  const float pre = (sgn * l1 - z) / quad;
  w = fabsf(z) > l1 ? pre : 0.f;
  n = n_new;
}

__device__ __forceinline__ void opt_step(const OptK &o, float &w, float &a, float &b, float &c, float g) {
  switch (o.kind) {
    case WD_OPT_SGD:       // GradientDescentOptimizer: var -= lr * g
      w -= o.lr * g;
      break;
    case WD_OPT_ADAGRAD:   // accum += g^2; var -= lr * g / sqrt(accum)
      b += g * g;
      w -= o.lr * g / sqrtf(b);
      break;
    case WD_OPT_FTRL:      // p2 = learning_rate_power (TF default -0.5; 0 = fixed learning rate)
      if (o.p2 == -0.5f && o.p3 == 0.0f) ftrl_update(w, a, b, g, o.lr, o.p0, o.p1);
      else ftrl_update_pow(w, a, b, g, o.lr, o.p0, o.p1, o.p2, o.p3);
      break;
    case WD_OPT_RMSPROP_CENTERED:   // ApplyCenteredRMSProp: + mg += (g - mg)(1 - decay); denominator sqrt(ms - mg^2 + eps)
      a += (g * g - a) * (1.0f - o.p0);
      c += (g - c) * (1.0f - o.p0);
      b = b * o.p1 + o.lr * g / sqrtf(a - c * c + o.p2);
      w -= b;
      break;
    case WD_OPT_RMSPROP:   // ms += (g^2 - ms)(1 - decay); mom = mom*momentum + lr*g/sqrt(ms + eps); var -= mom
      a += (g * g - a) * (1.0f - o.p0);
      b = b * o.p1 + o.lr * g / sqrtf(a + o.p2);
      w -= b;
      break;
    default: {             // Adam on a touched row: AdamOptimizer._apply_sparse_shared (m*b1 + (1-b1)g, ...)
      a = a * o.p0 + (1.0f - o.p0) * g;
      b = b * o.p1 + (1.0f - o.p1) * g * g;
      const float lr_t = o.lr * sqrtf(1.0f - o.b2p) / (1.0f - o.b1p);
      w -= lr_t * a / (sqrtf(b) + o.p2);
    }
  }
}

struct UpdArgs {
  float *emb, *accum, *wide, *bias;
  float *accum_a;            // second embedding slot table (generic optimizers; `accum` is slot b)
  float *accum_c;            // third one (centered RMSProp's mean gradient)
  uint32_t *touched;         // Adam: bit per fused row updated here (wd_adam_untouched does the others)
  OptK oe, ow;               // embedding (dnn scope) / wide + bias (linear scope) optimizer
  const float *pow_e, *pow_w;
  const wd_slot_t *slots;
  const int32_t *bag_offs;
  const float *dx;
  const float *dlogit;
  int64_t ldx, batch, ld_dlogit;
  int32_t S, nb;
  float lr_emb, lr_w, l1, l2;
  // row-record layout (wd_sparse_apply_rec): `emb` points at records of rec_stride floats indexed by the FUSED row (key),
  // the embedding row in front, {w, z, n, -} behind it at `wide` = emb + dim; the accumulator table keeps its flat layout.
  // 0: separate tables (emb rows of `dim` floats from sl.emb_off, wide lines of 4 floats).
  int32_t rec_stride;
  // pairs that do NOT arrive in ascending bag order inside a bucket (wd_bucket_onehot): one-row buckets are sorted as well
  int32_t unstable;
  // patch of the NEXT batch's prefetched input layer (wd_prefetch_onehot ran before this update): every row rewritten here
  // that the next batch reads too -- found in the next batch's bucket of the same row range -- is stored into the next x
  // tile / wide-weight list from the registers that hold its new value.  nstart == NULL: off.
  const int32_t *nstart;
  const uint64_t *npairs;
  float *nx, *nwv;
  int64_t nldx;
};


// all comparators ascending (mirror first step), so virtual +inf padding beyond m never moves.
// k and j are powers of two: indices by shifts/masks.
template <typename PtrT>
__device__ __forceinline__ void bitonic_sort(PtrT p, int m) {
  int lgP = 0;
  while ((1 << lgP) < m) ++lgP;
  const int half = (1 << lgP) >> 1;
  for (int lk = 1; lk <= lgP; ++lk) {
    const int k = 1 << lk, kh = k >> 1;
    for (int tq = threadIdx.x; tq < half; tq += 256) {
      const int blk = tq >> (lk - 1), off = tq & (kh - 1);
      const int i1 = (blk << lk) + off, i2 = (blk << lk) + k - 1 - off;
      if (i2 < m) {
        const uint64_t a = p[i1], b = p[i2];
        if (a > b) { p[i1] = b; p[i2] = a; }
      }
    }
    __syncthreads();
    for (int lj = lk - 2; lj >= 0; --lj) {
      const int j = 1 << lj;
      for (int tq = threadIdx.x; tq < half; tq += 256) {
        const int i1 = ((tq >> lj) << (lj + 1)) | (tq & (j - 1)), i2 = i1 + j;
        if (i2 < m) {
          const uint64_t a = p[i1], b = p[i2];
          if (a > b) { p[i1] = b; p[i2] = a; }
        }
      }
      __syncthreads();
    }
  }
}

template <bool GEN>   // GEN: optimizers other than Adagrad (embeddings) + Ftrl (wide); same control flow
__global__ void __launch_bounds__(256)
k_bucket_update(UpdArgs u, const int32_t *__restrict__ start, uint64_t *__restrict__ pairs) {
  if (GEN) {   // Adam: the beta powers live in HBM so that a captured graph sees them advance
    if (u.pow_e) { u.oe.b1p = u.pow_e[0]; u.oe.b2p = u.pow_e[1]; }
    if (u.pow_w) { u.ow.b1p = u.pow_w[0]; u.ow.b2p = u.pow_w[1]; }
  }
  // generic per-row updates: up to 4 consecutive parameters of an embedding row / one wide {w, a, b, -} line
  auto emb_apply = [&](int64_t off, int cnt, const float *g) {
    for (int k2 = 0; k2 < cnt; ++k2) {
      float w = u.emb[off + k2];
      float a = u.accum_a ? u.accum_a[off + k2] : 0.f, b = u.accum ? u.accum[off + k2] : 0.f;
      float c = u.accum_c ? u.accum_c[off + k2] : 0.f;
      opt_step(u.oe, w, a, b, c, g[k2]);
      u.emb[off + k2] = w;
      if (u.accum_a) u.accum_a[off + k2] = a;
      if (u.accum) u.accum[off + k2] = b;
      if (u.accum_c) u.accum_c[off + k2] = c;
    }
  };
  auto wide_apply = [&](float4 &r, float g) { opt_step(u.ow, r.x, r.y, r.z, r.w, g); };   // line = {w, a, b, c}
  auto touch = [&](uint32_t key) {
    if (GEN && u.touched) atomicOr(&u.touched[key >> 5], 1u << (key & 31));
  };
  const int64_t ws = u.rec_stride ? u.rec_stride : 4;        // floats between two wide lines
  // offset of a row in the accumulator table (always flat) -> offset of the same row in `emb`
  auto emb_at = [&](int64_t acc_off, uint32_t key) { return u.rec_stride ? (int64_t)key * u.rec_stride : acc_off; };
  __shared__ uint64_t lds_pairs[CAP_LDS];
  __shared__ float4 red[256];                               // long-segment tree; doubles as the rank-sort input
  uint64_t *lds_in = reinterpret_cast<uint64_t *>(red);     // RANK_MAX * 8 B == 256 * 16 B
  __shared__ float redw[256];
  __shared__ int long_i0[MAX_LONG], long_i1[MAX_LONG];
  __shared__ int nlong;
  __shared__ int32_t seg_ex[256];
  __shared__ float seg_scale[256];
  __shared__ wd_slot_t lds_slots[MAX_SLOTS_LDS];   // slot descriptors staged once per workgroup
  const int t = threadIdx.x;

  if ((int)blockIdx.x == u.nb) {  // bias_weights: g = sum_b dlogit[b] (fixed-shape tree), dense FTRL
    if (!u.bias) return;
    float acc = 0.f;
    for (int64_t i0 = t; i0 < u.batch; i0 += 8 * 256) {      // eight loads in flight per trip, same order of adds
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = i0 + 256 * k < u.batch ? u.dlogit[(i0 + 256 * k) * u.ld_dlogit] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + 256 * k < u.batch) acc += v[k];
    }
    redw[t] = acc;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
      if (t < st) redw[t] += redw[t + st];
      __syncthreads();
    }
    if (t == 0) {
      float w = u.bias[0], z = u.bias[1], n = u.bias[2], c3 = u.bias[3];
      if (GEN) {
        OptK ob = u.ow;
        if (ob.kind == WD_OPT_ADAM) ob.kind = WD_OPT_ADAM_DENSE;   // a dense [1] variable: ApplyAdam's form
        if (ob.kind == WD_OPT_ADAM_DENSE) {
          z += (redw[0] - z) * (1.0f - ob.p0);
          n += (redw[0] * redw[0] - n) * (1.0f - ob.p1);
          w -= ob.lr * sqrtf(1.0f - ob.b2p) / (1.0f - ob.b1p) * z / (sqrtf(n) + ob.p2);
        } else {
          opt_step(ob, w, z, n, c3, redw[0]);
        }
      } else {
        ftrl_update(w, z, n, redw[0], u.lr_w, u.l1, u.l2);
      }
      u.bias[0] = w; u.bias[1] = z; u.bias[2] = n;
      if (GEN) u.bias[3] = c3;
    }
    return;
  }

  const int bkt = start[u.nb + 2 + blockIdx.x];   // largest buckets first (order list written by k_bucket_scatter)
  const int32_t s0 = start[bkt];
  const int m = start[bkt + 1] - s0;
  int32_t nst0 = 0, nst1 = 0;
  if (!GEN && u.nstart) {      // (with the loads above: same dependency, one round trip)
    nst0 = u.nstart[bkt];
    nst1 = u.nstart[bkt + 1];
  }
  if (t == 0) nlong = 0;
  if (m == 0) return;
  // The same row range of the NEXT batch (its input layer was gathered before this update): a row rewritten here has to be
  // stored into the next x tile again iff the next batch holds it too -- 1.6 % of the rows with uniform ids.  Every lane takes
  // one of the next batch's pairs of this bucket (loaded here, next to the bucket's own: one more dependent round trip per
  // workgroup costs 30 % of this kernel) and, when all updates of the bucket are done, looks its row up in the sorted list.
  int32_t n0 = 0, m2 = 0;
  uint64_t np0 = 0;
  if (!GEN && u.nstart) {
    n0 = nst0;
    m2 = nst1 - nst0;
    if (t < m2) np0 = u.npairs[n0 + t];
  }
  const bool slots_in_lds = u.S <= MAX_SLOTS_LDS;
  if (slots_in_lds && t < u.S) lds_slots[t] = u.slots[t];   // overlaps with the pair loads below
  const uint64_t *sp;  // sorted pairs (flat pointer: LDS or global)
  // the scatter is stable (ascending bag order inside a bucket); a bucket of a slot with bucket_shift == 0 holds ONE row
  const bool one_row = u.slots[(int32_t)(uint32_t)pairs[s0] % u.S].bucket_shift == 0;
  const bool single_row = one_row && !u.unstable;
  if (single_row) {
    __syncthreads();
    sp = pairs + s0;           // already grouped and ordered: nothing to sort
  } else if (m <= RANK_MAX) {
    // rank sort: position of element i = number of elements ordered before it (ties by index); every lane reads
    // the same LDS word per step (broadcast, conflict-free), no barriers inside
    for (int i = t; i < m; i += 256) lds_in[i] = pairs[s0 + i];
    __syncthreads();
    for (int i = t; i < m; i += 256) {
      const uint64_t x = lds_in[i];
      int r = 0;
      for (int j = 0; j < m; ++j) {
        const uint64_t y = lds_in[j];
        r += (y < x || (y == x && j < i)) ? 1 : 0;
      }
      lds_pairs[r] = x;
    }
    __syncthreads();
    sp = lds_pairs;
  } else if (m <= CAP_LDS) {
    for (int i = t; i < m; i += 256) lds_pairs[i] = pairs[s0 + i];
    __syncthreads();
    bitonic_sort<uint64_t *>(lds_pairs, m);
    sp = lds_pairs;
  } else {
    __syncthreads();
    bitonic_sort<uint64_t *>(pairs + s0, m);
    sp = pairs + s0;
  }

  const int S = u.S;
  const int gidx = t >> 2, gl = t & 3;
  const bool whole_long = one_row && m > LONG_SEG;   // one row, many occurrences: the bucket IS one long segment
  if (whole_long && t == 0) {
    nlong = 1;
    long_i0[0] = 0;
    long_i1[0] = m;
  }
  // ---- short segments: one 4-lane group per unique row ------------------------------------------------
  for (int i = gidx; i < (whole_long ? 0 : m); i += 64) {
    const uint32_t key = (uint32_t)(sp[i] >> 32);
    if (i > 0 && (uint32_t)(sp[i - 1] >> 32) == key) continue;  // not a segment head
    int e = i + 1;
    while (e < m && e - i <= LONG_SEG && (uint32_t)(sp[e] >> 32) == key) ++e;
    bool is_long = (e - i > LONG_SEG);
    if (is_long) {
      int slot_l = -1;
      if (gl == 0) slot_l = atomicAdd(&nlong, 1);
      slot_l = __shfl(slot_l, (t & 63) & ~3, 64);
      if (slot_l < MAX_LONG) {
        if (gl == 0) {
          // upper bound of key in sp[e..m)
          int lo = e, hi = m;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((uint32_t)(sp[mid] >> 32) == key) lo = mid + 1; else hi = mid;
          }
          long_i0[slot_l] = i;
          long_i1[slot_l] = lo;
        }
        continue;
      }
      // list full: fall through and reduce serially (correct, just slower)
      while (e < m && (uint32_t)(sp[e] >> 32) == key) ++e;
    }
    const int32_t bag0 = (int32_t)(uint32_t)sp[i];
    const int32_t sidx = bag0 % S;
    const wd_slot_t sl = slots_in_lds ? lds_slots[sidx] : u.slots[sidx];
    const bool do_emb = u.emb && sl.kind == WD_SLOT_EMBEDDING && (int64_t)key - sl.row_base < sl.num_buckets;
    const bool do_wide = u.wide && sl.wide;
    const int D = sl.dim;
    if (e - i == 1 && (D & 3) == 0 && D <= 16) {
      // the common case (a row hit once, dim <= 16): ONE round of independent loads -- bag length, dx, accumulator,
      // row, dlogit and the wide {w,z,n} line are all addressed from (key, bag) alone
      const int64_t b = bag0 / S;
      const int32_t o0 = u.bag_offs[bag0], o1 = u.bag_offs[bag0 + 1];
      const bool lane_emb = do_emb && gl < (D >> 2);
      const int64_t off = sl.emb_off + ((int64_t)key - sl.row_base) * D + 4 * gl;
      const int64_t eoff = GEN ? off : emb_at(off - 4 * gl, key) + 4 * gl;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f), a = d, w = d, r = d;
      float dl = 0.f;
      if (lane_emb) {
        d = *reinterpret_cast<const float4 *>(u.dx + b * u.ldx + sl.out_col + 4 * gl);
        if (!GEN) {
          a = *reinterpret_cast<float4 *>(u.accum + off);
          w = *reinterpret_cast<float4 *>(u.emb + eoff);
        }
      }
      if (do_wide && gl == 0) {
        dl = u.dlogit[b * u.ld_dlogit];
        r = *reinterpret_cast<float4 *>(u.wide + (int64_t)key * ws);
      }
      if (lane_emb) {
        const int32_t len = o1 - o0;
        const float scale = len > 1 ? 1.0f / (float)len : 1.0f;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);   // 0 + d*scale: same rounding as the general loop
        g.x += d.x * scale; g.y += d.y * scale; g.z += d.z * scale; g.w += d.w * scale;
        if (GEN) {
          const float gg[4] = {g.x, g.y, g.z, g.w};
          emb_apply(off, 4, gg);
        } else {
          const float4 wn = adagrad4(a, w, g, u.lr_emb);
          *reinterpret_cast<float4 *>(u.accum + off) = a;
          *reinterpret_cast<float4 *>(u.emb + eoff) = wn;
        }
      }
      if (do_wide && gl == 0) {
        float g = 0.f;
        g += dl;
        if (GEN) wide_apply(r, g);
        else ftrl_update(r.x, r.y, r.z, g, u.lr_w, u.l1, u.l2);
        *reinterpret_cast<float4 *>(u.wide + (int64_t)key * ws) = r;
      }
      if (gl == 0) touch(key);
      continue;
    }
    if (do_emb && (D & 3) == 0) {
      const int64_t off = sl.emb_off + ((int64_t)key - sl.row_base) * D;
      const int64_t eoff = GEN ? off : emb_at(off, key);
      for (int c = gl; c < (D >> 2); c += 4) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        // four occurrences per round: their (bag length, gradient row) loads are independent and issued together, the adds
        // keep the order of the plain loop (bit-identical sums).  Rows hit 8-32 times (skewed ids, multi-hot bags) used to be a
        // chain of that many exposed round trips: Zipf(1.05) 89 -> ~70 us for the update of a C2 batch.
        int j = i;
        for (; !GEN && j + 4 <= e; j += 4) {      // (the generic-optimizer instantiation keeps the plain loop: registers)
          int32_t bag[4], len[4];
          float4 d[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) bag[q] = (int32_t)(uint32_t)sp[j + q];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            len[q] = u.bag_offs[bag[q] + 1] - u.bag_offs[bag[q]];
            d[q] = *reinterpret_cast<const float4 *>(u.dx + (int64_t)(bag[q] / S) * u.ldx + sl.out_col + 4 * c);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float scale = len[q] > 1 ? 1.0f / (float)len[q] : 1.0f;
            g.x += d[q].x * scale; g.y += d[q].y * scale; g.z += d[q].z * scale; g.w += d[q].w * scale;
          }
        }
        for (; j < e; ++j) {
          const int32_t bag = (int32_t)(uint32_t)sp[j];
          const int32_t len = u.bag_offs[bag + 1] - u.bag_offs[bag];
          const float scale = len > 1 ? 1.0f / (float)len : 1.0f;
          const float4 d = *reinterpret_cast<const float4 *>(u.dx + (int64_t)(bag / S) * u.ldx + sl.out_col + 4 * c);
          g.x += d.x * scale; g.y += d.y * scale; g.z += d.z * scale; g.w += d.w * scale;
        }
        if (GEN) {
          const float gg[4] = {g.x, g.y, g.z, g.w};
          emb_apply(off + 4 * c, 4, gg);
        } else {
          float4 a = *reinterpret_cast<float4 *>(u.accum + off + 4 * c);
          const float4 w = *reinterpret_cast<float4 *>(u.emb + eoff + 4 * c);
          const float4 wn = adagrad4(a, w, g, u.lr_emb);
          *reinterpret_cast<float4 *>(u.accum + off + 4 * c) = a;
          *reinterpret_cast<float4 *>(u.emb + eoff + 4 * c) = wn;
        }
      }
    } else if (do_emb) {  // dims that are not a multiple of 4 (opt-in override only)
      const int64_t off = sl.emb_off + ((int64_t)key - sl.row_base) * D;
      const int64_t eoff = GEN ? off : emb_at(off, key);
      for (int d0 = gl; d0 < D; d0 += 4) {
        float g = 0.f;
        for (int j = i; j < e; ++j) {
          const int32_t bag = (int32_t)(uint32_t)sp[j];
          const int32_t len = u.bag_offs[bag + 1] - u.bag_offs[bag];
          const float scale = len > 1 ? 1.0f / (float)len : 1.0f;
          g += u.dx[(int64_t)(bag / S) * u.ldx + sl.out_col + d0] * scale;
        }
        if (GEN) {
          emb_apply(off + d0, 1, &g);
        } else {
          const float a = u.accum[off + d0] + g * g;
          u.accum[off + d0] = a;
          u.emb[eoff + d0] -= u.lr_emb * g / sqrtf(a);
        }
      }
    }
    if (do_wide && gl == 0) {
      float g = 0.f;
      int j = i;
      for (; j + 4 <= e; j += 4) {      // (loads of four occurrences together, adds in order)
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = u.dlogit[(int64_t)((int32_t)(uint32_t)sp[j + q] / S) * u.ld_dlogit];
#pragma unroll
        for (int q = 0; q < 4; ++q) g += v[q];
      }
      for (; j < e; ++j) g += u.dlogit[(int64_t)((int32_t)(uint32_t)sp[j] / S) * u.ld_dlogit];
      float4 r = *reinterpret_cast<float4 *>(u.wide + (int64_t)key * ws);  // {w, z, n, -}
      if (GEN) wide_apply(r, g);
      else ftrl_update(r.x, r.y, r.z, g, u.lr_w, u.l1, u.l2);
      *reinterpret_cast<float4 *>(u.wide + (int64_t)key * ws) = r;
    }
    if (gl == 0) touch(key);
  }
  __syncthreads();
  // ---- long segments: the whole workgroup per row, 64 lane groups in parallel + fixed-shape tree -------
  const int nl = nlong < MAX_LONG ? nlong : MAX_LONG;
  for (int q = 0; q < nl; ++q) {
    const int i = long_i0[q], e = long_i1[q];
    const uint32_t key = (uint32_t)(sp[i] >> 32);
    if (t == 0) touch(key);
    const int32_t sidx = (int32_t)(uint32_t)sp[i] % S;
    const wd_slot_t sl = slots_in_lds ? lds_slots[sidx] : u.slots[sidx];
    const bool do_emb = u.emb && sl.kind == WD_SLOT_EMBEDDING && (int64_t)key - sl.row_base < sl.num_buckets;
    const bool do_wide = u.wide && sl.wide;
    const int D = sl.dim;
    if (do_emb) {
      const int64_t off = sl.emb_off + ((int64_t)key - sl.row_base) * D;
      const int64_t eoff = GEN ? off : emb_at(off, key);
      const int nchunk = (D + 3) >> 2;
      for (int c0 = 0; c0 < nchunk; c0 += 4) {
        const int c = c0 + gl;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        // tiles of 256 occurrences: (example, 1/len) staged by all lanes in one round of loads, then every 4-lane
        // group issues its four gradient-row loads back to back (same summation order as a j += 64 walk)
        for (int j0 = i; j0 < e; j0 += 256) {
          const int n = e - j0 < 256 ? e - j0 : 256;
          if (t < n) {
            const int32_t bag = (int32_t)(uint32_t)sp[j0 + t];
            const int32_t len = u.bag_offs[bag + 1] - u.bag_offs[bag];
            seg_ex[t] = bag / S;
            seg_scale[t] = len > 1 ? 1.0f / (float)len : 1.0f;
          }
          __syncthreads();
          if (c < nchunk) {
            float4 d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int idx = gidx + 64 * q;
              d[q] = make_float4(0.f, 0.f, 0.f, 0.f);
              if (idx < n) {
                const float *dp = u.dx + (int64_t)seg_ex[idx] * u.ldx + sl.out_col + 4 * c;
                if ((D & 3) == 0) {
                  d[q] = *reinterpret_cast<const float4 *>(dp);
                } else {
                  d[q].x = dp[0];
                  if (4 * c + 1 < D) d[q].y = dp[1];
                  if (4 * c + 2 < D) d[q].z = dp[2];
                  if (4 * c + 3 < D) d[q].w = dp[3];
                }
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int idx = gidx + 64 * q;
              if (idx < n) {
                const float scale = seg_scale[idx];
                g.x += d[q].x * scale; g.y += d[q].y * scale; g.z += d[q].z * scale; g.w += d[q].w * scale;
              }
            }
          }
          __syncthreads();
        }
        red[t] = g;
        __syncthreads();
        for (int st = 32; st >= 1; st >>= 1) {
          if (gidx < st) {
            float4 a = red[t], b = red[t + 4 * st];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            red[t] = a;
          }
          __syncthreads();
        }
        if (gidx == 0 && c < nchunk) {
          g = red[t];
          float gg[4] = {g.x, g.y, g.z, g.w};
          if (GEN) {
            emb_apply(off + 4 * c, D - 4 * c < 4 ? D - 4 * c : 4, gg);
          } else {
            for (int k2 = 0; k2 < 4 && 4 * c + k2 < D; ++k2) {
              const int64_t o = off + 4 * c + k2;
              const float a = u.accum[o] + gg[k2] * gg[k2];
              u.accum[o] = a;
              u.emb[eoff + 4 * c + k2] -= u.lr_emb * gg[k2] / sqrtf(a);
            }
          }
        }
        __syncthreads();
      }
    }
    if (do_wide) {
      float g = 0.f;
      for (int j = i + t; j < e; j += 1024) {   // four independent (pair -> dlogit) chains per lane and round
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int jj = j + 256 * q;
          v[q] = jj < e ? u.dlogit[(int64_t)((int32_t)(uint32_t)sp[jj] / S) * u.ld_dlogit] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) g += v[q];
      }
      redw[t] = g;
      __syncthreads();
      for (int st = 128; st >= 1; st >>= 1) {
        if (t < st) redw[t] += redw[t + st];
        __syncthreads();
      }
      if (t == 0) {
        float4 r = *reinterpret_cast<float4 *>(u.wide + (int64_t)key * ws);
        if (GEN) wide_apply(r, redw[0]);
        else ftrl_update(r.x, r.y, r.z, redw[0], u.lr_w, u.l1, u.l2);
        *reinterpret_cast<float4 *>(u.wide + (int64_t)key * ws) = r;
      }
      __syncthreads();
    }
  }
  // ---- patch of the next batch's prefetched input layer: every pair (row, bag') of the next batch in this row range whose row
  // was rewritten above -- found by binary search in the sorted list -- is stored again from the table (device-scope loads:
  // the rows were written by other wavefronts of this workgroup, complete behind the barrier, but not through this CU's L1)
  if (GEN || m2 == 0) return;
  __syncthreads();
  for (int i = t; i < m2; i += 256) {
    const uint64_t pr = i == t ? np0 : u.npairs[n0 + i];
    const uint32_t key = (uint32_t)(pr >> 32);
    int lo = 0, hi = m;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((uint32_t)(sp[mid] >> 32) < key) lo = mid + 1; else hi = mid;
    }
    if (lo >= m || (uint32_t)(sp[lo] >> 32) != key) continue;
    const int32_t bag2 = (int32_t)(uint32_t)pr;
    const int32_t sidx = bag2 % S;
    const wd_slot_t sl = slots_in_lds ? lds_slots[sidx] : u.slots[sidx];
    const float *src = u.emb + (int64_t)key * u.rec_stride;
    float *dst = u.nx + (int64_t)(bag2 / S) * u.nldx + sl.out_col;
    for (int d = 0; d < sl.dim; ++d) dst[d] = __hip_atomic_load(src + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sl.wide)
      u.nwv[bag2] = __hip_atomic_load(u.wide + (int64_t)key * ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace

extern "C" int32_t wd_bucket_max(void) { return MAX_NB; }
extern "C" int32_t wd_bucket_chunks(void) { return MAX_CHUNKS; }

// Phase 1 (needs only the ids): bucket the occurrences.  Independent of the forward / tower work of the step, so the
// engine runs it on a side stream while the tower computes.
extern "C" int wd_sparse_bucketize(const wd_slot_t *slots, int32_t S, const int32_t *ids, const int32_t *bag_offs,
                                   int64_t batch, int64_t nnz, int32_t *bucket_cnt, int32_t *bucket_start, int32_t *rank,
                                   uint64_t *pairs, int32_t nbuckets, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(slots && ids && bag_offs && bucket_cnt && bucket_start && rank && pairs, "null pointer");
  WD_REQUIRE(S > 0 && nbuckets > 0 && nbuckets <= MAX_NB, "bad bucket geometry");
  WD_REQUIRE(batch * S < ((int64_t)1 << 31), "batch * S must fit 31 bits");
  hipStream_t st = wd::as_stream(stream);
  const int64_t nbags = batch * S;
  // bucket_cnt layout: [MAX_CHUNKS][nbuckets] counts, [MAX_CHUNKS][nbuckets] chunk prefixes, total[nbuckets]
  // ~2 k occurrences per chunk, 128 to MAX_CHUNKS chunks: a chunk's fixed cost (its LDS counters, its row of the count matrix,
  // the scan of the totals in K2) is paid per chunk and bucket -- one id per bag (the sharded owner's request list, 213 k
  // occurrences) stays at 128 chunks, configs[3]'s 1.06 M occurrences take all 512.  WD_BUCKET_CHUNKS overrides (A/B runs)
  static const int chunk_env = getenv("WD_BUCKET_CHUNKS") ? atoi(getenv("WD_BUCKET_CHUNKS")) : 0;
  int64_t want = chunk_env > 0 ? chunk_env : wd::ceil_div(nnz, (int64_t)2048);
  want = want < 128 ? 128 : (want > MAX_CHUNKS ? MAX_CHUNKS : want);
  int64_t bags_per_chunk = wd::ceil_div(wd::ceil_div(nbags, want), 256) * 256;
  const int nchunks = nnz > 0 ? (int)wd::ceil_div(nbags, bags_per_chunk) : 0;
  int32_t *cpre = bucket_cnt + (int64_t)MAX_CHUNKS * nbuckets;
  int32_t *total = cpre + (int64_t)MAX_CHUNKS * nbuckets;
  if (nchunks > 0) {
    hipLaunchKernelGGL(k_bucket_hist, dim3(nchunks), dim3(256), 0, st, slots, S, ids, bag_offs, nbags, bags_per_chunk,
                       nbuckets, bucket_cnt, rank);
  }
  const int groups = nchunks > 0 ? (nchunks + SCAN_DEPTH - 1) / SCAN_DEPTH : 1;
  hipLaunchKernelGGL(k_bucket_colscan, dim3((unsigned)wd::ceil_div(nbuckets, 64)), dim3(64 * groups), 0, st, bucket_cnt,
                     cpre, nchunks, nbuckets, total);
  hipLaunchKernelGGL(k_bucket_scatter, dim3(nchunks > 0 ? nchunks : 1), dim3(256), 0, st, slots, S, ids, bag_offs,
                     nchunks > 0 ? nbags : (int64_t)0, bags_per_chunk, nbuckets, total, cpre, rank, bucket_start, pairs);
  return wd::check_launch("wd_sparse_bucketize");
}

// Phase 2 (needs dx / dlogit): per bucket sort + duplicate reduction + Adagrad / FTRL, and the bias update.
extern "C" int wd_sparse_apply(float *emb, float *emb_accum, float *wide, float *bias_wzn, const wd_slot_t *slots,
                               int32_t S, const int32_t *bag_offs, int64_t batch, const float *dx, int64_t ldx,
                               const float *dlogit, int64_t ld_dlogit, float lr_emb, float lr_wide, float l1, float l2,
                               const int32_t *bucket_start, uint64_t *pairs, int32_t nbuckets, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(slots && bag_offs && bucket_start && pairs, "null pointer");
  WD_REQUIRE(S > 0 && nbuckets > 0 && nbuckets <= MAX_NB, "bad bucket geometry");
  WD_REQUIRE(!emb || (emb_accum && dx), "embedding update needs accum and dx");
  WD_REQUIRE(!(wide || bias_wzn) || dlogit, "wide / bias update needs dlogit");
  UpdArgs u;
  u.emb = emb; u.accum = emb_accum; u.wide = wide; u.bias = bias_wzn; u.slots = slots; u.bag_offs = bag_offs;
  u.dx = dx; u.dlogit = dlogit; u.ldx = ldx; u.batch = batch; u.S = S; u.nb = nbuckets;
  u.ld_dlogit = ld_dlogit > 0 ? ld_dlogit : 1;
  u.lr_emb = lr_emb; u.lr_w = lr_wide; u.l1 = l1; u.l2 = l2;
  u.accum_a = u.accum_c = nullptr; u.touched = nullptr; u.oe = OptK{}; u.ow = OptK{}; u.pow_e = u.pow_w = nullptr;
  u.rec_stride = 0;
  u.unstable = 0; u.nstart = nullptr; u.npairs = nullptr; u.nx = u.nwv = nullptr; u.nldx = 0;
  hipLaunchKernelGGL(k_bucket_update<false>, dim3((unsigned)nbuckets + 1), dim3(256), 0, wd::as_stream(stream), u,
                     bucket_start, pairs);
  return wd::check_launch("wd_sparse_apply");
}

// wd_sparse_apply on the row-record layout: ONE table of `rec_stride`-float records indexed by the fused row, each holding
// the embedding row ([0, dim)) and the wide line {w, z, n, -} ([dim, dim + 4)) of that row -- an update touches two random
// lines per row (record + accumulator) instead of three (profiles/r2z_layouts.txt: 35 -> 29 us for one C2 batch).
extern "C" int wd_sparse_apply_rec(float *rec, int32_t rec_stride, int32_t dim, float *emb_accum, float *bias_wzn,
                                   const wd_slot_t *slots, int32_t S, const int32_t *bag_offs, int64_t batch,
                                   const float *dx, int64_t ldx, const float *dlogit, int64_t ld_dlogit, float lr_emb,
                                   float lr_wide, float l1, float l2, const int32_t *bucket_start, uint64_t *pairs,
                                   int32_t nbuckets, const wd_apply_next_t *next, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(rec && emb_accum && dx && dlogit && slots && bag_offs && bucket_start && pairs, "null pointer");
  WD_REQUIRE(S > 0 && nbuckets > 0 && nbuckets <= MAX_NB, "bad bucket geometry");
  WD_REQUIRE(dim > 0 && dim % 4 == 0 && rec_stride % 4 == 0 && rec_stride >= dim + 4, "record = [dim | w z n -], 16-byte aligned");
  UpdArgs u;
  u.emb = rec; u.accum = emb_accum; u.wide = rec + dim; u.bias = bias_wzn; u.slots = slots; u.bag_offs = bag_offs;
  u.dx = dx; u.dlogit = dlogit; u.ldx = ldx; u.batch = batch; u.S = S; u.nb = nbuckets;
  u.ld_dlogit = ld_dlogit > 0 ? ld_dlogit : 1;
  u.lr_emb = lr_emb; u.lr_w = lr_wide; u.l1 = l1; u.l2 = l2;
  u.accum_a = u.accum_c = nullptr; u.touched = nullptr; u.oe = OptK{}; u.ow = OptK{}; u.pow_e = u.pow_w = nullptr;
  u.rec_stride = rec_stride;
  u.unstable = 0; u.nstart = nullptr; u.npairs = nullptr; u.nx = u.nwv = nullptr; u.nldx = 0;
  if (next) {
    u.unstable = next->unsorted_buckets;
    if (next->bucket_start) {
      WD_REQUIRE(next->pairs && next->x && next->wide_vals && next->ldx > 0 && dim <= 16,
                 "wd_apply_next_t: the next batch's pairs, x tile and wide-weight list (dim <= 16)");
      u.nstart = next->bucket_start; u.npairs = next->pairs; u.nx = next->x; u.nwv = next->wide_vals; u.nldx = next->ldx;
    }
  }
  hipLaunchKernelGGL(k_bucket_update<false>, dim3((unsigned)nbuckets + 1), dim3(256), 0, wd::as_stream(stream), u,
                     bucket_start, pairs);
  return wd::check_launch("wd_sparse_apply_rec");
}

static OptK to_optk(const wd_opt_t *o) {
  OptK k{};
  if (o) {
    k.kind = o->kind; k.lr = o->lr; k.p0 = o->p0; k.p1 = o->p1; k.p2 = o->p2; k.p3 = o->p3;
    k.b1p = o->p0; k.b2p = o->p1;   // Adam without a power buffer: first step
  }
  return k;
}

static bool opt_ok(const wd_opt_t *o) {
  return o && ((o->kind >= WD_OPT_SGD && o->kind <= WD_OPT_ADAM) || o->kind == WD_OPT_RMSPROP_CENTERED);
}

// wd_sparse_apply with the optimizer of each scope chosen by the caller (model_util.py:84-90).
extern "C" int wd_sparse_apply_opt(float *emb, float *emb_a, float *emb_b, float *wide, float *bias, const wd_slot_t *slots,
                                   int32_t S, const int32_t *bag_offs, int64_t batch, const float *dx, int64_t ldx,
                                   const float *dlogit, int64_t ld_dlogit, const wd_opt_t *emb_opt,
                                   const wd_opt_t *wide_opt, const int32_t *bucket_start, uint64_t *pairs,
                                   int32_t nbuckets, uint32_t *touched, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(slots && bag_offs && bucket_start && pairs, "null pointer");
  WD_REQUIRE(S > 0 && nbuckets > 0 && nbuckets <= MAX_NB, "bad bucket geometry");
  WD_REQUIRE(!emb || (dx && opt_ok(emb_opt)), "embedding update needs dx and a valid optimizer");
  WD_REQUIRE(!(wide || bias) || (dlogit && opt_ok(wide_opt)), "wide / bias update needs dlogit and a valid optimizer");
  if (emb) {
    const int k = emb_opt->kind;
    WD_REQUIRE(k == WD_OPT_SGD || emb_b, "this optimizer needs slot table b");
    WD_REQUIRE(k == WD_OPT_SGD || k == WD_OPT_ADAGRAD || emb_a, "this optimizer needs slot table a");
    WD_REQUIRE(k != WD_OPT_RMSPROP_CENTERED || emb_opt->slot_c, "centered RMSProp needs wd_opt_t.slot_c (mean-gradient table)");
  }
  const bool adam = (emb && emb_opt->kind == WD_OPT_ADAM) || ((wide || bias) && wide_opt->kind == WD_OPT_ADAM);
  WD_REQUIRE(!adam || touched, "Adam needs the touched-row bitmap (wd_adam_untouched updates the other rows)");
  UpdArgs u;
  u.accum_c = (emb && emb_opt->kind == WD_OPT_RMSPROP_CENTERED) ? emb_opt->slot_c : nullptr;
  u.emb = emb; u.accum = emb_b; u.accum_a = emb_a; u.wide = wide; u.bias = bias; u.slots = slots; u.bag_offs = bag_offs;
  u.dx = dx; u.dlogit = dlogit; u.ldx = ldx; u.batch = batch; u.S = S; u.nb = nbuckets;
  u.ld_dlogit = ld_dlogit > 0 ? ld_dlogit : 1;
  u.lr_emb = u.lr_w = u.l1 = u.l2 = 0.f;
  u.rec_stride = 0;
  u.unstable = 0; u.nstart = nullptr; u.npairs = nullptr; u.nx = u.nwv = nullptr; u.nldx = 0;
  u.touched = adam ? touched : nullptr;
  u.oe = to_optk(emb ? emb_opt : nullptr); u.ow = to_optk((wide || bias) ? wide_opt : nullptr);
  u.pow_e = (emb && emb_opt->kind == WD_OPT_ADAM) ? emb_opt->pow : nullptr;
  u.pow_w = ((wide || bias) && wide_opt->kind == WD_OPT_ADAM) ? wide_opt->pow : nullptr;
  hipLaunchKernelGGL(k_bucket_update<true>, dim3((unsigned)nbuckets + 1), dim3(256), 0, wd::as_stream(stream), u,
                     bucket_start, pairs);
  return wd::check_launch("wd_sparse_apply_opt");
}

namespace {

// dense variables of a scope (tower kernels / biases / BN affines): one elementwise launch, tf.train kernels' forms
__global__ void __launch_bounds__(256)
k_opt_dense(float *__restrict__ w, float *__restrict__ sa, float *__restrict__ sb, float *__restrict__ sc,
            const float *__restrict__ g, int64_t n, OptK o, const float *__restrict__ pw) {
  if (pw) { o.b1p = pw[0]; o.b2p = pw[1]; }
  const float lr_t = o.kind == WD_OPT_ADAM ? o.lr * sqrtf(1.0f - o.b2p) / (1.0f - o.b1p) : 0.f;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float wv = w[i], a = sa ? sa[i] : 0.f, b = sb ? sb[i] : 0.f, c = sc ? sc[i] : 0.f;
    const float gi = g[i];
    if (o.kind == WD_OPT_ADAM) {   // ApplyAdam: m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); var -= lr_t m / (sqrt(v) + eps)
      a += (gi - a) * (1.0f - o.p0);
      b += (gi * gi - b) * (1.0f - o.p1);
      wv -= lr_t * a / (sqrtf(b) + o.p2);
    } else {
      opt_step(o, wv, a, b, c, gi);
    }
    w[i] = wv;
    if (sa) sa[i] = a;
    if (sb) sb[i] = b;
    if (sc) sc[i] = c;
  }
}

// Adam moves EVERY row of a sparsely updated variable (AdamOptimizer._apply_sparse_shared assigns m*beta1, v*beta2 and
// the var update over the whole variable): rows without a gradient this step, marked 0 in `touched`, get g = 0 here.
// One thread per (row, 4-float chunk) of a slot / per wide line; the bitmap is cleared by wd_adam_untouched's memset.
__global__ void __launch_bounds__(256)
k_adam_untouched_emb(float *__restrict__ emb, float *__restrict__ m, float *__restrict__ v,
                     const wd_slot_t *__restrict__ slots, const uint32_t *__restrict__ touched, OptK o,
                     const float *__restrict__ pw) {
  const wd_slot_t sl = slots[blockIdx.y];
  if (sl.emb_off < 0 || sl.kind != WD_SLOT_EMBEDDING) return;
  if (pw) { o.b1p = pw[0]; o.b2p = pw[1]; }
  const float lr_t = o.lr * sqrtf(1.0f - o.b2p) / (1.0f - o.b1p);
  const int64_t n = (int64_t)sl.num_buckets * sl.dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t key = (uint32_t)(sl.row_base + i / sl.dim);
    if (touched[key >> 5] & (1u << (key & 31))) continue;
    const int64_t e = sl.emb_off + i;
    const float a = m[e] * o.p0, b = v[e] * o.p1;
    m[e] = a;
    v[e] = b;
    emb[e] -= lr_t * a / (sqrtf(b) + o.p2);
  }
}

__global__ void __launch_bounds__(256)
k_adam_untouched_wide(float *__restrict__ wide, const wd_slot_t *__restrict__ slots, const uint32_t *__restrict__ touched,
                      OptK o, const float *__restrict__ pw) {
  const wd_slot_t sl = slots[blockIdx.y];
  if (!sl.wide) return;
  if (pw) { o.b1p = pw[0]; o.b2p = pw[1]; }
  const float lr_t = o.lr * sqrtf(1.0f - o.b2p) / (1.0f - o.b1p);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sl.num_buckets; i += (int64_t)gridDim.x * 256) {
    const uint32_t key = (uint32_t)(sl.row_base + i);
    if (touched[key >> 5] & (1u << (key & 31))) continue;
    float4 r = *reinterpret_cast<float4 *>(wide + (int64_t)key * 4);
    r.y *= o.p0;
    r.z *= o.p1;
    r.x -= lr_t * r.y / (sqrtf(r.z) + o.p2);
    *reinterpret_cast<float4 *>(wide + (int64_t)key * 4) = r;
  }
}

__global__ void k_adam_tick(float *pw, float b1, float b2) {
  pw[0] *= b1;
  pw[1] *= b2;
}

}  // namespace

extern "C" int wd_opt_dense(float *w, float *slot_a, float *slot_b, const float *g, int64_t n, const wd_opt_t *opt,
                            wd_stream_t stream) {
  if (n <= 0) return WD_OK;
  WD_REQUIRE(w && g && opt_ok(opt), "null pointer / bad optimizer");
  WD_REQUIRE(opt->kind == WD_OPT_SGD || slot_b, "this optimizer needs slot b");
  WD_REQUIRE(opt->kind == WD_OPT_SGD || opt->kind == WD_OPT_ADAGRAD || slot_a, "this optimizer needs slot a");
  WD_REQUIRE(opt->kind != WD_OPT_RMSPROP_CENTERED || opt->slot_c, "centered RMSProp needs wd_opt_t.slot_c");
  const int blocks = (int)std::min<int64_t>(wd::ceil_div(n, 256), 2048);
  hipLaunchKernelGGL(k_opt_dense, dim3(blocks), dim3(256), 0, wd::as_stream(stream), w, slot_a, slot_b,
                     opt->kind == WD_OPT_RMSPROP_CENTERED ? opt->slot_c : nullptr, g, n, to_optk(opt),
                     opt->kind == WD_OPT_ADAM ? opt->pow : nullptr);
  return wd::check_launch("wd_opt_dense");
}

extern "C" int wd_adam_untouched(float *emb, float *emb_m, float *emb_v, float *wide, const wd_slot_t *slots, int32_t S,
                                 int64_t max_rows, int64_t total_rows, uint32_t *touched, const wd_opt_t *emb_opt,
                                 const wd_opt_t *wide_opt, wd_stream_t stream) {
  WD_REQUIRE(slots && touched && S > 0 && total_rows > 0, "null pointer");
  hipStream_t st = wd::as_stream(stream);
  if (emb && emb_opt && emb_opt->kind == WD_OPT_ADAM) {
    WD_REQUIRE(emb_m && emb_v, "Adam needs both slot tables");
    const unsigned gx = (unsigned)std::min<int64_t>(wd::ceil_div(max_rows * 64, 256), 4096);
    hipLaunchKernelGGL(k_adam_untouched_emb, dim3(gx, (unsigned)S), dim3(256), 0, st, emb, emb_m, emb_v, slots, touched,
                       to_optk(emb_opt), emb_opt->pow);
  }
  if (wide && wide_opt && wide_opt->kind == WD_OPT_ADAM) {
    const unsigned gx = (unsigned)std::min<int64_t>(wd::ceil_div(max_rows, 256), 4096);
    hipLaunchKernelGGL(k_adam_untouched_wide, dim3(gx, (unsigned)S), dim3(256), 0, st, wide, slots, touched,
                       to_optk(wide_opt), wide_opt->pow);
  }
  if (hipMemsetAsync(touched, 0, (size_t)((total_rows + 31) / 32) * 4, st) != hipSuccess) {
    wd::set_error("wd_adam_untouched: hipMemsetAsync failed");
    return WD_ERR_LAUNCH;
  }
  return wd::check_launch("wd_adam_untouched");
}

extern "C" int wd_adam_tick(float *pow, float beta1, float beta2, wd_stream_t stream) {
  WD_REQUIRE(pow, "null pointer");
  hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, wd::as_stream(stream), pow, beta1, beta2);
  return wd::check_launch("wd_adam_tick");
}

extern "C" int wd_sparse_bwd_fused(float *emb, float *emb_accum, float *wide, float *bias_wzn, const wd_slot_t *slots,
                                   int32_t S, const int32_t *ids, const int32_t *bag_offs, int64_t batch, int64_t nnz,
                                   const float *dx, int64_t ldx, const float *dlogit, int64_t ld_dlogit, float lr_emb,
                                   float lr_wide, float l1, float l2, int32_t *bucket_cnt, int32_t *bucket_start,
                                   int32_t *rank, uint64_t *pairs, int32_t nbuckets, wd_stream_t stream) {
  int rc = wd_sparse_bucketize(slots, S, ids, bag_offs, batch, nnz, bucket_cnt, bucket_start, rank, pairs, nbuckets,
                               stream);
  if (rc != WD_OK) return rc;
  return wd_sparse_apply(emb, emb_accum, wide, bias_wzn, slots, S, bag_offs, batch, dx, ldx, dlogit, ld_dlogit, lr_emb,
                         lr_wide, l1, l2, bucket_start, pairs, nbuckets, stream);
}
