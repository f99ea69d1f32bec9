// wide_deep_amd/csrc/small_tables.hip -- categorical columns whose whole table fits in LDS and whose bags are LONG: the crossed
// columns of python/lib/build_estimator.py:138-155 (tf.feature_column.crossed_column over multi-valued keys: a bag holds the
// PRODUCT of its keys' counts -- ~25 / ~125 ids per example for BASELINE configs[3]'s crosses over 2 / 3 slots of mean 5 -- hashed
// into a few hundred buckets), their embedding_column (combiner='mean', dnn.py:83-90 input_layer) and linear_model weight
// (linear.py:29-36), with the reference's optimizers on them (joint.py:224-262: Adagrad on the dnn scope, Ftrl on linear; sparse
// IndexedSlices semantics: a row moves once per step, by the sum of its occurrences' gradients, only if the batch holds it).
//
// The general path (sparse_fused.hip) sorts occurrences into row-range buckets; a 200-row table that a batch hits 1.2 M times is
// its worst case -- every row a "long segment" of ~5 k occurrences, and the order-stable histogram of one-row buckets walks the
// longest bag of every 256 in lock-step: 1.9 of the 2.0 ms of a configs[3] step (profiles/r5_c4_crosses_kernel_stats_before.md).
// Here the table never leaves the CU:
//   k_small_fwd    a workgroup takes 8 examples; per small slot it loads the table (+ wide weights) into LDS, then one
//                  wavefront per bag: lanes stride over the ids, add rows from LDS, meet in a fixed shuffle tree; mean -> x,
//                  wide sum -> added to the wide logit once per example (slot order, no atomics).
//   k_small_bwd    (slice of the batch, slot) per workgroup, bags in ascending order: the bag's ids are COUNTED into an LDS
//                  histogram (integer atomics: exact, order-free), then the owner thread of every row adds
//                  count x (dx / len | dlogit) to its partial sums in LDS -- a fixed order of float adds, no sort, no float atomic;
//                  one barrier per bag (histograms double-buffered); the slice's bag bounds and gradients are staged in one
//                  round of loads, the ids of four bags ride in a register ring.  Partials + hit counts go to HBM per slice.
//   k_small_apply  per slot: partials summed in a fixed tree over the slices, Adagrad on the embedding rows / Ftrl on {w, z, n} of the rows
//                  the batch touched.
#include "common.h"

namespace {

constexpr int SM_MAX_DIM = 16;
constexpr int SM_SLICE = WD_SMALL_BAGS_PER_SLICE;
constexpr int SM_EX_PER_WG = 8;       // forward: examples per workgroup (2 per wavefront: a bag is a chain of two round trips --
                                      // offsets, ids -- that 4 k wavefronts hide better than 1 k with eight bags each)

struct SmallArgs {
  const wd_slot_t *slots;
  const int32_t *small_idx;   // [nsmall] slot numbers
  int32_t nsmall, S;
  const int32_t *ids, *bag_offs;
  int64_t batch;
  // forward
  const float *emb, *wide;
  float *x;
  int64_t ldx;
  float *wide_logit;
  // backward
  const float *dx, *dlogit;
  float *part;                // [nsmall][nslice][part_rows][D_max + 2]  (g_0 .. g_{D-1}, g_wide, count)
  int32_t nslice, part_rows, part_w;
  int32_t cnt_ints;           // LDS words of the backward's [bag][row] count matrix
  int64_t bags_per_slice;
  // apply
  float *emb_w, *emb_acc, *wide_w;
  float lr_emb, lr_w, l1, l2;
  // table layout: separate tables (es == 0: row r of a slot at emb + emb_off + r * D; wide line at wide + (row_base + r) * 4) or
  // row records (engine.py: es = ws = the record stride; row at emb + (row_base + r) * es, wide line at wide + (row_base + r) * ws,
  // `wide` pointing at the {w, z, n, -} part of record 0); the Adagrad accumulator is flat in both
  int64_t es, ws;
};

__device__ __forceinline__ int64_t sm_row(const SmallArgs &a, const wd_slot_t &sl, int r, int D) {
  return a.es ? (sl.row_base + r) * a.es : sl.emb_off + (int64_t)r * D;
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float wave_sum(float v) {     // fixed-shape tree over the 64 lanes
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// LDS row of the forward: LGP float4 lanes per id (1, 2 or 4 for D <= 4, 8, 16), the row padded by one float4 so that rows of
// neighbouring ids start 4 * (LGP + 1) banks apart (rounds 5-6: rows of D floats read one float per lane -- for D = 16 every lane of a
// wavefront on one of TWO banks, a 32-way conflict on each of the 16 reads of an id: 46 us for the 1.2 M ids of a configs[3] batch)
// (one float4 per id -- D <= 4 -- needs no padding: random ids spread over the eight groups of four banks)
__host__ __device__ __forceinline__ int sm_lgs(int D) { return D <= 4 ? 0 : (D <= 8 ? 1 : 2); }        // log2(lanes per id)
__host__ __device__ __forceinline__ int sm_fwd_stride(int D) { return D <= 0 ? 0 : (D <= 4 ? 4 : (4 << sm_lgs(D)) + 4); }

__global__ void __launch_bounds__(256) k_small_fwd(SmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];     // table [R][DS] then wide weights [R]
  constexpr int EPW = SM_EX_PER_WG / 4;      // examples per wavefront
  constexpr int NI = 8;                      // ids per lane in flight
  const int t = threadIdx.x, lane = t & 63, wave = uni(t >> 6);
  const int64_t b0 = (int64_t)blockIdx.x * SM_EX_PER_WG + wave * EPW;
  float wsum[EPW];
#pragma unroll
  for (int e = 0; e < EPW; ++e) wsum[e] = 0.f;
  bool any_wide = false;
  for (int k = 0; k < a.nsmall; ++k) {
    const int s = uni(a.small_idx[k]);
    const wd_slot_t sl = a.slots[s];
    const int R = uni(sl.num_buckets), D = uni(sl.dim);
    const int LGS = sm_lgs(D), LGP = 1 << LGS, DS = sm_fwd_stride(D), IPI = 64 >> LGS;      // ids per wavefront and round
    const int i = lane >> LGS, q = lane & (LGP - 1);
    float *T = lds, *W = lds + (int64_t)R * DS;
    // this wavefront's bags of the column: bounds, then the first NI ids of every lane -- in flight while the table lands
    int32_t j0[EPW], j1[EPW], idv[EPW][NI];
#pragma unroll
    for (int e = 0; e < EPW; ++e) {
      j0[e] = 0; j1[e] = 0;
      if (b0 + e < a.batch) {
        j0[e] = uni(a.bag_offs[(b0 + e) * a.S + s]);
        j1[e] = uni(a.bag_offs[(b0 + e) * a.S + s + 1]);
      }
#pragma unroll
      for (int u = 0; u < NI; ++u) {
        const int32_t j = j0[e] + i + u * IPI;
        idv[e][u] = j < j1[e] ? a.ids[j] : -1;
      }
    }
    __syncthreads();                         // the previous slot's table is no longer read
    if (D > 0) {
      const int DPS = LGS + 2, DP = 1 << DPS;
      for (int x = t; x < R * DP; x += 256) {
        const int r = x >> DPS, d = x & (DP - 1);
        T[r * DS + d] = d < D ? a.emb[sm_row(a, sl, r, D) + d] : 0.f;
      }
    }
    if (sl.wide)
      for (int x = t; x < R; x += 256) W[x] = a.wide[(sl.row_base + x) * a.ws];
    __syncthreads();
    any_wide = any_wide || sl.wide;
#pragma unroll
    for (int e = 0; e < EPW; ++e) {
      const int64_t b = b0 + e;
      if (b >= a.batch) break;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      float ws = 0.f;
      // combiner='mean' divides by the bag's length, as wd_embag_fwd and k_small_bwd do (a negative id -- none reaches a bag
      // of the featurizer -- contributes a zero row but still counts)
      const float cnt = (float)(j1[e] - j0[e]);
      auto one = [&](int32_t id) {
        if (id < 0) return;
        if (sl.wide && q == 0) ws += W[id];
        if (D > 0) {
          const float4 v = *reinterpret_cast<const float4 *>(T + id * DS + 4 * q);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      };
#pragma unroll
      for (int u = 0; u < NI; ++u) one(idv[e][u]);
      for (int32_t j = j0[e] + i + NI * IPI; j < j1[e]; j += IPI) one(a.ids[j]);
      for (int m = LGP; m < 64; m <<= 1) {         // fixed tree over the lanes of one quarter
        acc.x += __shfl_xor(acc.x, m, 64); acc.y += __shfl_xor(acc.y, m, 64);
        acc.z += __shfl_xor(acc.z, m, 64); acc.w += __shfl_xor(acc.w, m, 64);
        ws += __shfl_xor(ws, m, 64);
      }
      if (sl.wide) wsum[e] += ws;
      if (D > 0 && sl.out_col >= 0 && lane < LGP) {
        if (cnt > 1.f) { acc.x /= cnt; acc.y /= cnt; acc.z /= cnt; acc.w /= cnt; }     // (an empty bag stays the zero vector)
        float *xp = a.x + b * a.ldx + sl.out_col + 4 * q;
        const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (4 * q + c < D) xp[c] = v[c];
      }
    }
  }
  if (any_wide && a.wide_logit && lane == 0) {
#pragma unroll
    for (int e = 0; e < EPW; ++e)
      if (b0 + e < a.batch) a.wide_logit[b0 + e] += wsum[e];      // behind wd_wide_fwd (bias + the other columns) in stream order
  }
}

// Backward partials of one (slice of the batch, slot).  Round 5 walked the slice's bags one after the other -- count a bag into an
// LDS histogram, barrier, every row's owner adds count x gradient: 64 barriers in a row on 256 workgroups, 55 us at configs[3] for
// 10 MB of ids.  Now the bags of a pass (as many as fit: 64 at 200 rows) are counted TOGETHER into a [bag][row] matrix of integer
// LDS atomics (exact, order-free; four wavefronts, a bag each, the ids of eight bags in flight), one barrier, and the owner thread
// of a row walks the pass's bags in ascending order adding count x (dx / len | dlogit) -- the same adds in the same order as
// before (bit-identical partials), on staged products, without a barrier between them.
constexpr int SM_GS = 20;                    // staged gradient of a bag: [dx * (1 / len)  (16) | dlogit | - - -]
constexpr int SM_CNT_INTS = 12800;           // count matrix of a pass, at most (64 bags x 200 rows)

__global__ void __launch_bounds__(256) k_small_bwd(SmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = uni(t >> 6);
  const int k = blockIdx.y, c = blockIdx.x;
  const int s = uni(a.small_idx[k]);
  const wd_slot_t sl = a.slots[s];
  const int R = uni(sl.num_buckets), D = uni(sl.dim);
  const int PW = D + 2;                                   // partial record: g_0 .. g_{D-1}, g_wide, count
  float *part = lds;                                      // [R][PW]
  int32_t *cnt = reinterpret_cast<int32_t *>(lds + (int64_t)a.part_rows * a.part_w);      // [PB][R]
  int32_t *lo = cnt + a.cnt_ints, *hi = lo + SM_SLICE;       // [SM_SLICE] each: ids [lo, hi) of the slice's q-th bag
  float *gs = reinterpret_cast<float *>(hi + SM_SLICE);      // [SM_SLICE][SM_GS]
  const int64_t e0 = (int64_t)c * a.bags_per_slice;
  const int64_t e1 = e0 + a.bags_per_slice < a.batch ? e0 + a.bags_per_slice : a.batch;
  const int nbag = (int)(e1 - e0);
  int PB = a.cnt_ints / R;                                   // bags per pass (cnt_ints >= the longest table: at least one)
  PB = PB < (int)a.bags_per_slice ? PB : (int)a.bags_per_slice;
  for (int i = t; i < R * PW; i += 256) part[i] = 0.f;
  // the slice's bag bounds and gradients in ONE round of loads
  for (int i = t; i < nbag; i += 256) {
    lo[i] = a.bag_offs[(e0 + i) * a.S + s];
    hi[i] = a.bag_offs[(e0 + i) * a.S + s + 1];
  }
  for (int i = t; i < nbag * (D + 1); i += 256) {
    const int q = i / (D + 1), d = i - q * (D + 1);
    float v;
    if (d < D) {
      v = (a.dx && sl.out_col >= 0) ? a.dx[(e0 + q) * a.ldx + sl.out_col + d] : 0.f;
      const int32_t len = a.bag_offs[(e0 + q) * a.S + s + 1] - a.bag_offs[(e0 + q) * a.S + s];
      const float scale = len > 1 ? 1.0f / (float)len : 1.0f;      // mean combiner: every occurrence carries dx / len
      gs[q * SM_GS + d] = v * scale;
    } else {
      gs[q * SM_GS + SM_MAX_DIM] = (sl.wide && a.dlogit) ? a.dlogit[e0 + q] : 0.f;
    }
  }
  for (int q0 = 0; q0 < nbag; q0 += PB) {
    const int nb = nbag - q0 < PB ? nbag - q0 : PB;
    __syncthreads();                                      // staging done / the previous pass's counts are consumed
    for (int i = t; i < nb * R; i += 256) cnt[i] = 0;
    __syncthreads();
    // count: wavefront w takes bags w, w + 4, ...; the ids of eight of them (128 per bag) in flight at a time
    for (int qb = wave; qb < nb; qb += 32) {
      int32_t idr[8][2];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = qb + 4 * u;
        idr[u][0] = -1; idr[u][1] = -1;
        if (q < nb) {
          const int32_t j0 = lo[q0 + q], j1 = hi[q0 + q];
          if (j0 + lane < j1) idr[u][0] = a.ids[j0 + lane];
          if (j0 + lane + 64 < j1) idr[u][1] = a.ids[j0 + lane + 64];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = qb + 4 * u;
        if (q >= nb) break;
        int32_t *cp = cnt + q * R;
        if (idr[u][0] >= 0) atomicAdd(&cp[idr[u][0]], 1);
        if (idr[u][1] >= 0) atomicAdd(&cp[idr[u][1]], 1);
        const int32_t j1 = hi[q0 + q];
        for (int32_t j = lo[q0 + q] + 128 + lane; j < j1; j += 64) {
          const int32_t id = a.ids[j];
          if (id >= 0) atomicAdd(&cp[id], 1);
        }
      }
    }
    __syncthreads();
    for (int rr = t; rr < R; rr += 256) {
      float *pr = part + (int64_t)rr * PW;
      float acc[SM_MAX_DIM + 2];
#pragma unroll
      for (int d = 0; d < SM_MAX_DIM; ++d) acc[d] = d < D ? pr[d] : 0.f;
      acc[SM_MAX_DIM] = pr[D];
      acc[SM_MAX_DIM + 1] = pr[D + 1];
      for (int q = 0; q < nb; ++q) {
        const int32_t n = cnt[q * R + rr];
        if (n == 0) continue;
        const float fn = (float)n;
        const float *gp = gs + (q0 + q) * SM_GS;
#pragma unroll
        for (int d4 = 0; d4 < SM_MAX_DIM / 4; ++d4) {
          if (4 * d4 < D) {
            const float4 g = *reinterpret_cast<const float4 *>(gp + 4 * d4);
            acc[4 * d4 + 0] += fn * g.x; acc[4 * d4 + 1] += fn * g.y; acc[4 * d4 + 2] += fn * g.z; acc[4 * d4 + 3] += fn * g.w;
          }
        }
        acc[SM_MAX_DIM] += fn * gp[SM_MAX_DIM];
        acc[SM_MAX_DIM + 1] += fn;
      }
#pragma unroll
      for (int d = 0; d < SM_MAX_DIM; ++d)
        if (d < D) pr[d] = acc[d];
      pr[D] = acc[SM_MAX_DIM];
      pr[D + 1] = acc[SM_MAX_DIM + 1];
    }
  }
  __syncthreads();
  float *out = a.part + ((int64_t)k * a.nslice + c) * a.part_rows * a.part_w;
  for (int i = t; i < R * PW; i += 256) {
    const int r = i / PW, d = i - r * PW;
    out[(int64_t)r * a.part_w + d] = part[i];
  }
}

__device__ __forceinline__ void ftrl1(float &w, float &z, float &n, float g, float lr, float l1, float l2) {
  const float n_new = n + g * g;
  z += g - (sqrtf(n_new) - sqrtf(n)) / lr * w;
  const float quad = sqrtf(n_new) / lr + 2.0f * l2;
  const float sgn = z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f);
  const float pre = (sgn * l1 - z) / quad;
  w = fabsf(z) > l1 ? pre : 0.f;
  n = n_new;
}

// SM_AP lanes per (row, element): element d < D = embedding column d (Adagrad), d == D = the wide weight (Ftrl).  Lane l adds the
// partials of slices l * per .. (l + 1) * per - 1 in slice order, the SM_AP lane sums meet in a fixed shuffle tree -- one round of
// loads per lane instead of a chain of nslice / 32 rounds in ONE thread per element (round 5: 8 workgroups on the whole chip, 15 us
// alone and 50-70 us beside the row update's traffic, on the critical stream of a configs[3] step)
constexpr int SM_AP = 32;

__device__ __forceinline__ void small_slice_sum(const SmallArgs &a, int k, int r, int d, int D, int lane, float &g, float &hits) {
  const float *p = a.part + (int64_t)k * a.nslice * a.part_rows * a.part_w + (int64_t)r * a.part_w;
  const int64_t st = (int64_t)a.part_rows * a.part_w;
  const int per = (a.nslice + SM_AP - 1) / SM_AP;
  g = 0.f; hits = 0.f;
  constexpr int RND = 8;
  for (int c0 = lane * per; c0 < (lane + 1) * per; c0 += RND) {
    float v[RND], h[RND];
#pragma unroll
    for (int u = 0; u < RND; ++u) {
      const bool live = c0 + u < (lane + 1) * per && c0 + u < a.nslice;
      v[u] = live ? p[(c0 + u) * st + d] : 0.f;
      h[u] = live ? p[(c0 + u) * st + D + 1] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < RND; ++u) { g += v[u]; hits += h[u]; }
  }
#pragma unroll
  for (int m = SM_AP / 2; m >= 1; m >>= 1) {
    g += __shfl_xor(g, m, SM_AP);
    hits += __shfl_xor(hits, m, SM_AP);
  }
}

__global__ void __launch_bounds__(256) k_small_apply(SmallArgs a) {
  const int k = blockIdx.y;
  const int s = uni(a.small_idx[k]);
  const wd_slot_t sl = a.slots[s];
  const int R = sl.num_buckets, D = sl.dim;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int i = gid / SM_AP, lane = gid % SM_AP;
  if (i >= R * (D + 1)) return;          // (whole lane groups leave together: SM_AP divides the workgroup)
  const int r = i / (D + 1), d = i - r * (D + 1);
  float g, hits;
  small_slice_sum(a, k, r, d, D, lane, g, hits);
  if (lane != 0 || hits == 0.f) return;             // the batch does not hold this row: it does not move
  if (d < D) {
    if (!a.emb_w || sl.out_col < 0) return;
    const int64_t o = sl.emb_off + (int64_t)r * D + d;
    const float acc = a.emb_acc[o] + g * g;
    a.emb_acc[o] = acc;
    a.emb_w[sm_row(a, sl, r, D) + d] -= a.lr_emb * g / sqrtf(acc);
  } else if (sl.wide && a.wide_w) {
    float4 q = *reinterpret_cast<float4 *>(a.wide_w + (sl.row_base + r) * a.ws);
    ftrl1(q.x, q.y, q.z, g, a.lr_w, a.l1, a.l2);
    *reinterpret_cast<float4 *>(a.wide_w + (sl.row_base + r) * a.ws) = q;
  }
}

// ---- the same update in two halves, for tables that are REPLICATED on every rank of a row-sharded model (dist.py: a 200-row
// crossed column is not worth an all-to-all): k_small_reduce leaves this rank's gradient sums + hit counts per row
// ([nsmall][part_rows][part_w], partials added in the same fixed tree), the ranks all-reduce that buffer, k_small_apply_sum applies it.
__global__ void __launch_bounds__(256) k_small_reduce(SmallArgs a, float *__restrict__ gsum) {
  const int k = blockIdx.y;
  const int s = uni(a.small_idx[k]);
  const wd_slot_t sl = a.slots[s];
  const int R = sl.num_buckets, D = sl.dim, PW = D + 2;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int i = gid / SM_AP, lane = gid % SM_AP;
  if (i >= a.part_rows * a.part_w) return;
  const int r = i / a.part_w, d = i - r * a.part_w;
  float g = 0.f, hits = 0.f;
  if (r < R && d < PW) small_slice_sum(a, k, r, d, D, lane, g, hits);      // (d == D + 1: the hit count itself)
  if (lane == 0) gsum[(int64_t)k * a.part_rows * a.part_w + i] = g;       // (rows / columns a narrower table does not have: zeros)
}

__global__ void __launch_bounds__(256) k_small_apply_sum(SmallArgs a, const float *__restrict__ gsum) {
  const int k = blockIdx.y;
  const int s = uni(a.small_idx[k]);
  const wd_slot_t sl = a.slots[s];
  const int R = sl.num_buckets, D = sl.dim;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R * (D + 1)) return;
  const int r = i / (D + 1), d = i - r * (D + 1);
  const float *p = gsum + (int64_t)k * a.part_rows * a.part_w + (int64_t)r * a.part_w;
  const float g = p[d], hits = p[D + 1];
  if (hits == 0.f) return;                          // no rank's batch holds this row: it does not move
  if (d < D) {
    if (!a.emb_w || sl.out_col < 0) return;
    const int64_t o = sl.emb_off + (int64_t)r * D + d;
    const float acc = a.emb_acc[o] + g * g;
    a.emb_acc[o] = acc;
    a.emb_w[sm_row(a, sl, r, D) + d] -= a.lr_emb * g / sqrtf(acc);
  } else if (sl.wide && a.wide_w) {
    float4 q = *reinterpret_cast<float4 *>(a.wide_w + (sl.row_base + r) * a.ws);
    ftrl1(q.x, q.y, q.z, g, a.lr_w, a.l1, a.l2);
    *reinterpret_cast<float4 *>(a.wide_w + (sl.row_base + r) * a.ws) = q;
  }
}

}  // namespace

// bags per slice of the backward: a slice is ONE workgroup walking its bags one after the other (a barrier each), so a small batch
// -- the reference's own 64-512 -- is cut finer: 16 bags per slice up to 2048 examples (batch 512 of the shipped conf: 57 -> ~20 us)
static int64_t small_bags_per_slice(int64_t batch) { return batch <= 2048 ? 16 : WD_SMALL_BAGS_PER_SLICE; }

extern "C" int64_t wd_small_tables_ws_floats(int32_t nsmall, int32_t max_rows, int32_t max_dim, int64_t max_batch) {
  if (nsmall <= 0) return 0;
  // (the finer slicing of small batches never needs more slices than a full batch of the coarse one: 2048 / 16 = 128 <= ...)
  int64_t nslice = wd::ceil_div(max_batch, small_bags_per_slice(max_batch));
  if (max_batch > 2048 && nslice < 128) nslice = 128;      // an engine of capacity > 2048 may be handed a batch <= 2048
  return (int64_t)nsmall * nslice * max_rows * (max_dim + 2);
}

static int small_check(const wd_slot_t *slots, const int32_t *small_idx, int32_t nsmall, int32_t max_rows, int32_t max_dim) {
  WD_REQUIRE(slots && small_idx && nsmall > 0, "null pointer / no slots");
  WD_REQUIRE(max_rows > 0 && max_dim >= 0 && max_dim <= SM_MAX_DIM, "max_rows > 0, 0 <= max_dim <= 16");
  WD_REQUIRE((int64_t)max_rows * (max_dim + 2) <= WD_SMALL_MAX_FLOATS, "rows x (dim + 2) must fit WD_SMALL_MAX_FLOATS");
  return WD_OK;
}

extern "C" int wd_small_tables_fwd(const float *emb, const float *wide, const wd_slot_t *slots, int32_t S,
                                   const int32_t *small_idx, int32_t nsmall, int32_t max_rows, int32_t max_dim,
                                   const int32_t *ids, const int32_t *bag_offs, int64_t batch, float *x, int64_t ldx,
                                   float *wide_logit, int32_t rec_stride, wd_stream_t stream) {
  if (batch <= 0 || nsmall <= 0) return WD_OK;
  WD_REQUIRE(rec_stride >= 0 && rec_stride % 4 == 0, "rec_stride: 0 (separate tables) or the record stride in floats");
  const int rc = small_check(slots, small_idx, nsmall, max_rows, max_dim);
  if (rc != WD_OK) return rc;
  WD_REQUIRE(ids && bag_offs, "null pointer");
  WD_REQUIRE(!(max_dim > 0) || (emb && x), "embedded columns need the table and x");
  SmallArgs a{};
  a.slots = slots; a.small_idx = small_idx; a.nsmall = nsmall; a.S = S; a.ids = ids; a.bag_offs = bag_offs; a.batch = batch;
  a.emb = emb; a.wide = wide; a.x = x; a.ldx = ldx; a.wide_logit = wide_logit;
  a.es = rec_stride; a.ws = rec_stride ? rec_stride : 4;
  const size_t lds = (size_t)max_rows * (sm_fwd_stride(max_dim) + 1) * 4;      // (< 64 KB for every shape small_check admits)
  hipLaunchKernelGGL(k_small_fwd, dim3((unsigned)wd::ceil_div(batch, (int64_t)SM_EX_PER_WG)), dim3(256), lds,
                     wd::as_stream(stream), a);
  return wd::check_launch("wd_small_tables_fwd");
}

static int small_bwd_args(SmallArgs &a, size_t &lds, float *emb, float *emb_accum, float *wide_wzn, const wd_slot_t *slots, int32_t S,
                          const int32_t *small_idx, int32_t nsmall, int32_t max_rows, int32_t max_dim, const int32_t *ids,
                          const int32_t *bag_offs, int64_t batch, const float *dx, int64_t ldx, const float *dlogit, float lr_emb,
                          float lr_wide, float l1, float l2, float *ws, int64_t ws_floats, int32_t rec_stride = 0) {
  const int rc = small_check(slots, small_idx, nsmall, max_rows, max_dim);
  if (rc != WD_OK) return rc;
  WD_REQUIRE(ids && bag_offs && ws, "null pointer");
  WD_REQUIRE(!emb || (emb_accum && dx), "embedding update needs accum and dx");
  WD_REQUIRE(!wide_wzn || dlogit, "wide update needs dlogit");
  a = SmallArgs{};
  a.slots = slots; a.small_idx = small_idx; a.nsmall = nsmall; a.S = S; a.ids = ids; a.bag_offs = bag_offs; a.batch = batch;
  a.dx = emb ? dx : nullptr; a.ldx = ldx; a.dlogit = wide_wzn ? dlogit : nullptr;
  a.bags_per_slice = small_bags_per_slice(batch);
  a.nslice = (int32_t)wd::ceil_div(batch, a.bags_per_slice);
  a.part = ws; a.part_rows = max_rows; a.part_w = max_dim + 2;
  WD_REQUIRE((int64_t)nsmall * a.nslice * max_rows * (max_dim + 2) <= ws_floats, "workspace too small (wd_small_tables_ws_floats)");
  a.emb_w = emb; a.emb_acc = emb_accum; a.wide_w = wide_wzn;
  a.lr_emb = lr_emb; a.lr_w = lr_wide; a.l1 = l1; a.l2 = l2;
  a.es = rec_stride; a.ws = rec_stride ? rec_stride : 4;
  {
    const int64_t want = a.bags_per_slice * max_rows;      // every bag of a slice in one pass, if that fits
    a.cnt_ints = (int32_t)(want < SM_CNT_INTS ? want : (max_rows > SM_CNT_INTS ? max_rows : SM_CNT_INTS));
  }
  lds = (size_t)max_rows * (max_dim + 2) * 4 + (size_t)a.cnt_ints * 4 + (size_t)2 * SM_SLICE * 4 + (size_t)SM_SLICE * SM_GS * 4;
  // configs[3] (200 rows x 16 + 2, 64 bags per pass) asks for 14 + 50 + 5.5 KB, a wide-only table of ~4 k rows for 32 + 50 + 5.5:
  // beyond the 64 KB a launch gets without the attribute (WD_SMALL_MAX_FLOATS bounds it at 32 + 50 + 5.5 KB)
  static size_t lds_allowed = 64 * 1024;
  if (lds > lds_allowed) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_small_bwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds);
    WD_REQUIRE(e == hipSuccess, "hipFuncSetAttribute(k_small_bwd, dynamic LDS) failed");
    lds_allowed = lds;
  }
  return WD_OK;
}

extern "C" int wd_small_tables_bwd(float *emb, float *emb_accum, float *wide_wzn, const wd_slot_t *slots, int32_t S,
                                   const int32_t *small_idx, int32_t nsmall, int32_t max_rows, int32_t max_dim,
                                   const int32_t *ids, const int32_t *bag_offs, int64_t batch, const float *dx, int64_t ldx,
                                   const float *dlogit, float lr_emb, float lr_wide, float l1, float l2, float *ws,
                                   int64_t ws_floats, int32_t rec_stride, wd_stream_t stream) {
  if (batch <= 0 || nsmall <= 0) return WD_OK;
  WD_REQUIRE(rec_stride >= 0 && rec_stride % 4 == 0, "rec_stride: 0 (separate tables) or the record stride in floats");
  SmallArgs a;
  size_t lds;
  const int rc = small_bwd_args(a, lds, emb, emb_accum, wide_wzn, slots, S, small_idx, nsmall, max_rows, max_dim, ids, bag_offs, batch,
                                dx, ldx, dlogit, lr_emb, lr_wide, l1, l2, ws, ws_floats, rec_stride);
  if (rc != WD_OK) return rc;
  hipStream_t st = wd::as_stream(stream);
  hipLaunchKernelGGL(k_small_bwd, dim3((unsigned)a.nslice, (unsigned)nsmall), dim3(256), lds, st, a);
  hipLaunchKernelGGL(k_small_apply, dim3((unsigned)wd::ceil_div((int64_t)max_rows * (max_dim + 1) * SM_AP, (int64_t)256), (unsigned)nsmall),
                     dim3(256), 0, st, a);
  return wd::check_launch("wd_small_tables_bwd");
}

extern "C" int wd_small_tables_grad(const wd_slot_t *slots, int32_t S, const int32_t *small_idx, int32_t nsmall, int32_t max_rows,
                                    int32_t max_dim, const int32_t *ids, const int32_t *bag_offs, int64_t batch, const float *dx,
                                    int64_t ldx, const float *dlogit, float *ws, int64_t ws_floats, float *gsum, wd_stream_t stream) {
  WD_REQUIRE(gsum && nsmall > 0 && batch >= 0, "null pointer / no slots");
  SmallArgs a;
  size_t lds;
  float dummy = 0.f;       // (the two halves carry no tables: non-null markers make the kernel read dx / dlogit)
  const int rc = small_bwd_args(a, lds, dx ? &dummy : nullptr, dx ? &dummy : nullptr, dlogit ? &dummy : nullptr, slots, S, small_idx,
                                nsmall, max_rows, max_dim, ids, bag_offs, batch > 0 ? batch : 1, dx, ldx, dlogit, 0.f, 0.f, 0.f, 0.f,
                                ws, ws_floats);
  if (rc != WD_OK) return rc;
  a.emb_w = a.emb_acc = a.wide_w = nullptr;
  hipStream_t st = wd::as_stream(stream);
  if (batch <= 0) {        // a rank without examples still takes part in the all-reduce: zeros
    if (hipMemsetAsync(gsum, 0, sizeof(float) * (size_t)nsmall * max_rows * (max_dim + 2), st) != hipSuccess) {
      wd::set_error("wd_small_tables_grad: hipMemsetAsync failed");
      return WD_ERR_LAUNCH;
    }
    return wd::check_launch("wd_small_tables_grad");
  }
  hipLaunchKernelGGL(k_small_bwd, dim3((unsigned)a.nslice, (unsigned)nsmall), dim3(256), lds, st, a);
  hipLaunchKernelGGL(k_small_reduce, dim3((unsigned)wd::ceil_div((int64_t)max_rows * (max_dim + 2) * SM_AP, (int64_t)256), (unsigned)nsmall),
                     dim3(256), 0, st, a, gsum);
  return wd::check_launch("wd_small_tables_grad");
}

extern "C" int wd_small_tables_apply(float *emb, float *emb_accum, float *wide_wzn, const wd_slot_t *slots, int32_t S,
                                     const int32_t *small_idx, int32_t nsmall, int32_t max_rows, int32_t max_dim, const float *gsum,
                                     float lr_emb, float lr_wide, float l1, float l2, int32_t rec_stride, wd_stream_t stream) {
  if (nsmall <= 0) return WD_OK;
  const int rc = small_check(slots, small_idx, nsmall, max_rows, max_dim);
  if (rc != WD_OK) return rc;
  WD_REQUIRE(gsum && (!emb || emb_accum), "null pointer");
  SmallArgs a{};
  a.slots = slots; a.small_idx = small_idx; a.nsmall = nsmall; a.S = S;
  a.part_rows = max_rows; a.part_w = max_dim + 2;
  a.emb_w = emb; a.emb_acc = emb_accum; a.wide_w = wide_wzn;
  a.lr_emb = lr_emb; a.lr_w = lr_wide; a.l1 = l1; a.l2 = l2;
  a.es = rec_stride; a.ws = rec_stride ? rec_stride : 4;
  hipLaunchKernelGGL(k_small_apply_sum, dim3((unsigned)wd::ceil_div((int64_t)max_rows * (max_dim + 1), (int64_t)256), (unsigned)nsmall),
                     dim3(256), 0, wd::as_stream(stream), a, gsum);
  return wd::check_launch("wd_small_tables_apply");
}
