// wide_deep_amd/csrc/embag.hip -- forward sparse side of the train step on gfx950.
//
//   wd_embag_fwd      tf.feature_column.input_layer / embedding_column(combiner='mean')   python/lib/dnn.py:83-90
//   wd_indicator_fwd  indicator_column multi-hot counts                                   build_estimator.py:108,118
//   wd_dense_fwd      numeric_column(normalizer_fn)                                       build_estimator.py:61-68,121-136
//   wd_wide_fwd       tf.feature_column.linear_model(sparse_combiner='sum')               python/lib/linear.py:29-36
//   wd_bce_sum_fwd_bwd  binary head, sigmoid CE, SUM reduction                            python/lib/joint.py:216-222,264-269
//
// All of it is HBM-bound gather work.  Layout decisions:
//   * bags are example-major (bag = b*S + s) so a wavefront's pooled rows land in ONE contiguous
//     span of x (16 bags x 64 B for D=16) and its id / offset reads are coalesced;
//   * a row of D floats is read by D/4 lanes as float4 (16 B per lane, 1 KiB per wave instruction);
//   * each lane group keeps UNROLL independent row reads in flight to cover HBM latency.
#include "common.h"

namespace {

// WIDE (round 6, row records): the bag's wide sum rides along -- lane 0 adds the {w, ..} word behind every row it fetches (same
// 128-byte line) and leaves wide_vals[bag]; wd_wide_sum adds a row's bags up.  The separate wide pass re-read every line: 30 us of
// the 120 us input layer of a configs[3] step.
template <int LANES, bool WIDE = false>  // LANES = D/4 lanes cooperate on one bag
__global__ void __launch_bounds__(256)
k_embag_fwd(const float *__restrict__ emb, const wd_slot_t *__restrict__ slots, int32_t S,
            const int32_t *__restrict__ group_slots, int32_t ngroup, const int32_t *__restrict__ ids,
            const int32_t *__restrict__ bag_offs, int64_t nwork, float *__restrict__ x, int64_t ldx, int64_t RS4,
            float *__restrict__ wide_vals = nullptr) {
  WD_SIDE_PRIO();
  // RS4: row stride in float4 units (LANES for a dense table; larger for rows received through the exchange)
  constexpr int D = LANES * 4;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t work = tid / LANES;  // (b, g) pair
  const int lane = (int)(tid % LANES);
  if (work >= nwork) return;
  const int64_t b = work / ngroup;
  const int32_t s = group_slots[work - b * ngroup];
  // only the two fields needed: a by-value copy of the whole wd_slot_t went through 32 B/lane of scratch memory
  const int64_t emb_off = slots[s].emb_off;
  const int32_t out_col = slots[s].out_col;
  const int64_t bag = b * S + s;
  const int32_t j0 = bag_offs[bag], j1 = bag_offs[bag + 1];
  const float4 *__restrict__ tab = reinterpret_cast<const float4 *>(emb + emb_off);
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 *__restrict__ tv = reinterpret_cast<const f4 *>(tab);
  f4 acc = (f4)(0.f);
  float wacc = 0.f;
  const float *__restrict__ wp = emb + emb_off + D;      // WIDE: the {w, z, n, -} words sit behind the D floats of a row
  int32_t j = j0;
  for (; j + 4 <= j1; j += 4) {
    int32_t i0 = ids[j], i1 = ids[j + 1], i2 = ids[j + 2], i3 = ids[j + 3];
    f4 r0 = i0 >= 0 ? tv[(int64_t)i0 * RS4 + lane] : (f4)(0.f);
    f4 r1 = i1 >= 0 ? tv[(int64_t)i1 * RS4 + lane] : (f4)(0.f);
    f4 r2 = i2 >= 0 ? tv[(int64_t)i2 * RS4 + lane] : (f4)(0.f);
    f4 r3 = i3 >= 0 ? tv[(int64_t)i3 * RS4 + lane] : (f4)(0.f);
    if (WIDE && lane == 0) {
      const float w0 = i0 >= 0 ? wp[(int64_t)i0 * RS4 * 4] : 0.f, w1 = i1 >= 0 ? wp[(int64_t)i1 * RS4 * 4] : 0.f;
      const float w2 = i2 >= 0 ? wp[(int64_t)i2 * RS4 * 4] : 0.f, w3 = i3 >= 0 ? wp[(int64_t)i3 * RS4 * 4] : 0.f;
      wacc += w0; wacc += w1; wacc += w2; wacc += w3;
    }
    acc += r0; acc += r1; acc += r2; acc += r3;
  }
  for (; j < j1; ++j) {
    const int32_t i0 = ids[j];
    acc += i0 >= 0 ? tv[(int64_t)i0 * RS4 + lane] : (f4)(0.f);
    if (WIDE && lane == 0 && i0 >= 0) wacc += wp[(int64_t)i0 * RS4 * 4];
  }
  if (WIDE && lane == 0) wide_vals[bag] = wacc;
  const int32_t n = j1 - j0;
  if (n > 1) {  // combiner='mean': sum / count (duplicates counted), SURVEY App. A.6
    const float c = (float)n;
    acc.x /= c; acc.y /= c; acc.z /= c; acc.w /= c;
  }
  float *o = x + b * ldx + out_col + lane * 4;
  if ((((uintptr_t)o) & 15) == 0) {
    *reinterpret_cast<f4 *>(o) = acc;
  } else {
    o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
  }
  (void)D;
}

// Contiguous-slot variant (the engine's layout: the slots of one embedding dim are a contiguous slot range
// [slot0, slot0 + ngroup)): no group_slots indirection, per-slot metadata staged once per workgroup in LDS while
// the CSR offsets are already in flight, BPG bags per lane group so that every lane has BPG independent
// offset -> id -> row chains outstanding.  Dependent global round trips per bag: offsets, ids, row (was 5).
template <int LANES, int BPG, bool ONEHOT>
__global__ void __launch_bounds__(256)
k_embag_fwd_range(const float *__restrict__ emb, const wd_slot_t *__restrict__ slots, int32_t S, int32_t slot0,
                  int32_t ngroup, const int32_t *__restrict__ ids, const int32_t *__restrict__ bag_offs,
                  int64_t nwork, float *__restrict__ x, int64_t ldx) {
  constexpr int MAXG = 128;
  __shared__ int64_t s_emb_off[MAXG];
  __shared__ int32_t s_out_col[MAXG];
  const int t = threadIdx.x;
  const int lane = t % LANES;
  const int64_t grp = ((int64_t)blockIdx.x * 256 + t) / LANES;   // lane group index
  int64_t w[BPG], b[BPG];
  int32_t g[BPG], j0[BPG], j1[BPG];
#pragma unroll
  for (int q = 0; q < BPG; ++q) {
    w[q] = grp * BPG + q;
    const int64_t wc = w[q] < nwork ? w[q] : nwork - 1;
    b[q] = wc / ngroup;
    g[q] = (int32_t)(wc - b[q] * ngroup);
    const int64_t bag = b[q] * S + slot0 + g[q];
    if (ONEHOT) {
      j0[q] = (int32_t)bag;
      j1[q] = (int32_t)bag + 1;
    } else {
      j0[q] = bag_offs[bag];
      j1[q] = bag_offs[bag + 1];
    }
  }
  for (int i = t; i < ngroup; i += 256) {
    const wd_slot_t sl = slots[slot0 + i];
    s_emb_off[i] = sl.emb_off;
    s_out_col[i] = sl.out_col;
  }
  int32_t id0[BPG];
#pragma unroll
  for (int q = 0; q < BPG; ++q) id0[q] = j1[q] > j0[q] ? ids[j0[q]] : 0;
  __syncthreads();
  float4 acc[BPG];
#pragma unroll
  for (int q = 0; q < BPG; ++q) {
    const float4 *__restrict__ tab = reinterpret_cast<const float4 *>(emb + s_emb_off[g[q]]);
    acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j1[q] > j0[q]) {   // rows are touched once per step: nontemporal, do not pollute L2
      typedef float floatx4 __attribute__((ext_vector_type(4)));
      const floatx4 r = __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(&tab[(int64_t)id0[q] * LANES + lane]));
      acc[q] = make_float4(r.x, r.y, r.z, r.w);
    }
  }
#pragma unroll
  for (int q = 0; q < BPG; ++q) {
    const float4 *__restrict__ tab = reinterpret_cast<const float4 *>(emb + s_emb_off[g[q]]);
    // multi-hot tail: the ids of up to eight more occurrences in ONE round of loads, then their rows in one round, added in
    // bag order (groups of four + a one-by-one remainder made a bag of 7 a chain of 9 dependent round trips)
    for (int32_t j = j0[q] + 1; j < j1[q]; j += 8) {
      int32_t iv[8];
      float4 rv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) iv[u] = j + u < j1[q] ? ids[j + u] : 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) rv[u] = j + u < j1[q] ? tab[(int64_t)iv[u] * LANES + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j + u < j1[q]) { acc[q].x += rv[u].x; acc[q].y += rv[u].y; acc[q].z += rv[u].z; acc[q].w += rv[u].w; }
    }
    const int32_t n = j1[q] - j0[q];
    if (n > 1) {  // combiner='mean'
      const float c = (float)n;
      acc[q].x /= c; acc[q].y /= c; acc[q].z /= c; acc[q].w /= c;
    }
    if (w[q] < nwork) {
      float *o = x + b[q] * ldx + s_out_col[g[q]] + lane * 4;
      if ((((uintptr_t)o) & 15) == 0) {
        *reinterpret_cast<float4 *>(o) = acc[q];
      } else {
        o[0] = acc[q].x; o[1] = acc[q].y; o[2] = acc[q].z; o[3] = acc[q].w;
      }
    }
  }
}

// The same gather as a device function for the fused input-layer launch below.  Deliberately NOT shared with the kernel
// above: routing k_embag_fwd_range through this inlined body gives the same instruction mix but a kernel that measures
// 14.4 us instead of 11.7 us on MI355X (re-measured in round 2, profiles/r2z_bench_c2_uniform.json roofline_gather_kernel: code placement, MI355X_MICROARCH.md item 8).
constexpr int MAXG = 128;   // slots per dim group staged in LDS

template <int LANES, int BPG, bool ONEHOT>
__device__ __forceinline__ void embag_range_body(const float *__restrict__ emb, const wd_slot_t *__restrict__ slots,
                                                 int32_t S, int32_t slot0, int32_t ngroup,
                                                 const int32_t *__restrict__ ids, const int32_t *__restrict__ bag_offs,
                                                 int64_t nwork, float *__restrict__ x, int64_t ldx, int64_t block) {
  __shared__ int64_t s_emb_off[MAXG];   // declared here (not passed in) so that the accesses stay ds_* instructions
  __shared__ int32_t s_out_col[MAXG];
  const int t = threadIdx.x;
  const int lane = t % LANES;
  const int64_t grp = (block * 256 + t) / LANES;   // lane group index
  int64_t w[BPG], b[BPG];
  int32_t g[BPG], j0[BPG], j1[BPG];
#pragma unroll
  for (int q = 0; q < BPG; ++q) {
    w[q] = grp * BPG + q;
    const int64_t wc = w[q] < nwork ? w[q] : nwork - 1;
    b[q] = wc / ngroup;
    g[q] = (int32_t)(wc - b[q] * ngroup);
    const int64_t bag = b[q] * S + slot0 + g[q];
    if (ONEHOT) {
      j0[q] = (int32_t)bag;
      j1[q] = (int32_t)bag + 1;
    } else {
      j0[q] = bag_offs[bag];
      j1[q] = bag_offs[bag + 1];
    }
  }
  for (int i = t; i < ngroup; i += 256) {
    const wd_slot_t sl = slots[slot0 + i];
    s_emb_off[i] = sl.emb_off;
    s_out_col[i] = sl.out_col;
  }
  int32_t id0[BPG];
#pragma unroll
  for (int q = 0; q < BPG; ++q) id0[q] = j1[q] > j0[q] ? ids[j0[q]] : 0;
  __syncthreads();
  float4 acc[BPG];
#pragma unroll
  for (int q = 0; q < BPG; ++q) {
    const float4 *__restrict__ tab = reinterpret_cast<const float4 *>(emb + s_emb_off[g[q]]);
    acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j1[q] > j0[q]) {   // rows are touched once per step: nontemporal, do not pollute L2
      typedef float floatx4 __attribute__((ext_vector_type(4)));
      const floatx4 r = __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(&tab[(int64_t)id0[q] * LANES + lane]));
      acc[q] = make_float4(r.x, r.y, r.z, r.w);
    }
  }
#pragma unroll
  for (int q = 0; q < BPG; ++q) {
    const float4 *__restrict__ tab = reinterpret_cast<const float4 *>(emb + s_emb_off[g[q]]);
    // multi-hot tail: the ids of up to eight more occurrences in ONE round of loads, then their rows in one round, added in
    // bag order (groups of four + a one-by-one remainder made a bag of 7 a chain of 9 dependent round trips)
    for (int32_t j = j0[q] + 1; j < j1[q]; j += 8) {
      int32_t iv[8];
      float4 rv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) iv[u] = j + u < j1[q] ? ids[j + u] : 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) rv[u] = j + u < j1[q] ? tab[(int64_t)iv[u] * LANES + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j + u < j1[q]) { acc[q].x += rv[u].x; acc[q].y += rv[u].y; acc[q].z += rv[u].z; acc[q].w += rv[u].w; }
    }
    const int32_t n = j1[q] - j0[q];
    if (n > 1) {  // combiner='mean'
      const float c = (float)n;
      acc[q].x /= c; acc[q].y /= c; acc[q].z /= c; acc[q].w /= c;
    }
    if (w[q] < nwork) {
      float *o = x + b[q] * ldx + s_out_col[g[q]] + lane * 4;
      if ((((uintptr_t)o) & 15) == 0) {
        *reinterpret_cast<float4 *>(o) = acc[q];
      } else {
        o[0] = acc[q].x; o[1] = acc[q].y; o[2] = acc[q].z; o[3] = acc[q].w;
      }
    }
  }
}

// dims that are not a multiple of 4 (never produced by the reference's embedding_dim, kept for the
// opt-in embedding_dim override): one lane per (bag, element).
__global__ void k_embag_fwd_generic(const float *__restrict__ emb, const wd_slot_t *__restrict__ slots, int32_t S,
                                    const int32_t *__restrict__ group_slots, int32_t ngroup, int32_t D,
                                    const int32_t *__restrict__ ids, const int32_t *__restrict__ bag_offs,
                                    int64_t nwork, float *__restrict__ x, int64_t ldx) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t work = tid / D;
  const int d = (int)(tid % D);
  if (work >= nwork) return;
  const int64_t b = work / ngroup;
  const int32_t s = group_slots[work - b * ngroup];
  const wd_slot_t sl = slots[s];
  const int64_t bag = b * S + s;
  const int32_t j0 = bag_offs[bag], j1 = bag_offs[bag + 1];
  const float *tab = emb + sl.emb_off;
  float acc = 0.f;
  for (int32_t j = j0; j < j1; ++j) acc += tab[(int64_t)ids[j] * D + d];
  if (j1 - j0 > 1) acc /= (float)(j1 - j0);
  x[b * ldx + sl.out_col + d] = acc;
}

// indicator_column: multi-hot COUNTS of the bag's ids.  16 lanes per (example, slot): lane c owns the
// columns c, c + 16, ... and counts the ids equal to each -- no zero-fill pass, no read-modify-write, every column of the
// slot written exactly once.
__global__ void __launch_bounds__(256)
k_indicator_fwd(const wd_slot_t *__restrict__ slots, int32_t S, const int32_t *__restrict__ group_slots, int32_t ngroup,
                const int32_t *__restrict__ ids, const int32_t *__restrict__ bag_offs, int64_t nwork,
                float *__restrict__ x, int64_t ldx) {
  const int64_t work = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int lane = threadIdx.x & 15;
  if (work >= nwork) return;
  const int64_t b = work / ngroup;
  const int32_t s = group_slots[work - b * ngroup];
  const int32_t nb = slots[s].num_buckets, out_col = slots[s].out_col;
  float *o = x + b * ldx + out_col;
  const int64_t bag = b * S + s;
  const int32_t j0 = bag_offs[bag], j1 = bag_offs[bag + 1];
  for (int32_t c = lane; c < nb; c += 16) {
    float cnt = 0.f;
    for (int32_t j = j0; j < j1; ++j) cnt += ids[j] == c ? 1.0f : 0.0f;
    o[c] = cnt;
  }
}

__device__ __forceinline__ void dense_body(const float *__restrict__ dense, int64_t ld_dense,
                                           const wd_dense_col_t *__restrict__ cols, int32_t ncols, int64_t batch,
                                           float *__restrict__ x, int64_t ldx, int64_t block) {
  const int64_t i = block * 256 + threadIdx.x;
  if (i >= batch * ncols) return;
  const int64_t b = i / ncols;
  const int32_t j = (int32_t)(i - b * ncols);
  const wd_dense_col_t c = cols[j];
  float v = dense[b * ld_dense + j];
  if (c.kind == 1) v = (v - c.p0) / (c.p1 - c.p0);
  else if (c.kind == 2) v = (v - c.p0) / c.p1;
  else if (c.kind == 3) v = logf(v);
  x[b * ldx + c.out_col] = v;
}

__global__ void k_dense_fwd(const float *__restrict__ dense, int64_t ld_dense, const wd_dense_col_t *__restrict__ cols,
                            int32_t ncols, int64_t batch, float *__restrict__ x, int64_t ldx) {
  dense_body(dense, ld_dense, cols, ncols, batch, x, ldx, blockIdx.x);
}

// 16 lanes per example: lanes stride over the example's bags, then a shuffle reduction inside the
// 16-lane group (4 examples per wavefront).
__device__ __forceinline__ void wide_body(const float *__restrict__ wide, const float *__restrict__ bias,
                                          const wd_slot_t *__restrict__ slots, int32_t S,
                                          const int32_t *__restrict__ ids, const int32_t *__restrict__ bag_offs,
                                          int64_t batch, int32_t stride, float *__restrict__ out, int64_t block) {
  const int64_t tid = block * 256 + threadIdx.x;
  const int64_t b = tid >> 4;
  const int lane = (int)(tid & 15);
  float acc = 0.f;
  if (b < batch) {
    for (int32_t s = lane; s < S; s += 16) {
      const wd_slot_t sl = slots[s];
      if (!sl.wide || (sl.flags & WD_SLOT_F_SMALL)) continue;     // (small tables: wd_small_tables_fwd adds their share)
      const int64_t bag = b * S + s;
      const int32_t j0 = bag_offs[bag], j1 = bag_offs[bag + 1];
      for (int32_t j = j0; j < j1; j += 8) {      // eight ids, then their eight weights, each in one round of loads; adds in bag order
        int32_t iv[8];
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) iv[u] = j + u < j1 ? ids[j + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = iv[u] >= 0 ? wide[(sl.row_base + iv[u]) * stride] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (iv[u] >= 0) acc += wv[u];   // id < 0: dropped exchange entry
      }
    }
  }
  acc += __shfl_xor(acc, 8, 16);
  acc += __shfl_xor(acc, 4, 16);
  acc += __shfl_xor(acc, 2, 16);
  acc += __shfl_xor(acc, 1, 16);
  if (b < batch && lane == 0) out[b] = acc + bias[0];
}

// out[b] = bias + the wide sums of example b's bags (wide_vals[b * S + s], left by k_embag_fwd<.., true>): 16 lanes per example,
// lane l takes the slots l, l + 16, ..; small-table columns (wd_small_tables_fwd adds their share) and non-wide columns skipped
__global__ void __launch_bounds__(256)
k_wide_sum(const float *__restrict__ wide_vals, const float *__restrict__ bias, const wd_slot_t *__restrict__ slots, int32_t S,
           int64_t batch, float *__restrict__ out) {
  WD_SIDE_PRIO();
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t b = tid >> 4;
  const int lane = (int)(tid & 15);
  float acc = 0.f;
  if (b < batch)
    for (int32_t s = lane; s < S; s += 16) {
      const wd_slot_t sl = slots[s];
      if (!sl.wide || (sl.flags & WD_SLOT_F_SMALL)) continue;
      acc += wide_vals[b * S + s];
    }
  acc += __shfl_xor(acc, 8, 16);
  acc += __shfl_xor(acc, 4, 16);
  acc += __shfl_xor(acc, 2, 16);
  acc += __shfl_xor(acc, 1, 16);
  if (b < batch && lane == 0) out[b] = acc + bias[0];
}

__global__ void __launch_bounds__(256)
k_wide_fwd(const float *__restrict__ wide, const float *__restrict__ bias, const wd_slot_t *__restrict__ slots,
           int32_t S, const int32_t *__restrict__ ids, const int32_t *__restrict__ bag_offs, int64_t batch,
           int32_t stride, float *__restrict__ out) {
  WD_SIDE_PRIO();
  wide_body(wide, bias, slots, S, ids, bag_offs, batch, stride, out, blockIdx.x);
}

// The whole sparse forward of a step in ONE launch (three dependent-free pieces share a grid instead of paying three
// kernel boundaries): blocks [0, nb_emb) = embedding-bag gather, [nb_emb, nb_emb + nb_wide) = wide sum,
// the rest = numeric columns.
struct InputLayerArgs {
  const float *emb; const wd_slot_t *slots; const int32_t *ids; const int32_t *bag_offs;
  float *x; int64_t ldx; int64_t batch; int64_t nwork;
  int32_t S, slot0, ngroup;
  const float *wide; const float *bias; float *wide_out; int32_t wide_stride;
  const float *dense; int64_t ld_dense; const wd_dense_col_t *cols; int32_t ncols;
  int32_t nb_emb, nb_wide;
};

template <int LANES, int BPG, bool ONEHOT>
__global__ void __launch_bounds__(256) k_input_layer(InputLayerArgs a) {
  const int bid = blockIdx.x;
  if (bid < a.nb_emb) {
    embag_range_body<LANES, BPG, ONEHOT>(a.emb, a.slots, a.S, a.slot0, a.ngroup, a.ids, a.bag_offs, a.nwork, a.x, a.ldx,
                                         bid);
  } else if (bid < a.nb_emb + a.nb_wide) {
    wide_body(a.wide, a.bias, a.slots, a.S, a.ids, a.bag_offs, a.batch, a.wide_stride, a.wide_out, bid - a.nb_emb);
  } else {
    dense_body(a.dense, a.ld_dense, a.cols, a.ncols, a.batch, a.x, a.ldx, bid - a.nb_emb - a.nb_wide);
  }
}

__global__ void k_bce(const float *__restrict__ dnn_logit, const float *__restrict__ wide_logit,
                      const float *__restrict__ labels, const float *__restrict__ weights, int64_t batch,
                      float *__restrict__ logit, float *__restrict__ prob, float *__restrict__ dlogit,
                      float *__restrict__ loss_sum) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float l = 0.f;
  if (i < batch) {
    float x = 0.f;
    if (dnn_logit) x += dnn_logit[i];
    if (wide_logit) x += wide_logit[i];
    const float y = labels[i];
    const float w = weights ? weights[i] : 1.0f;
    const float e = expf(-fabsf(x));
    // sigmoid_cross_entropy_with_logits: max(x,0) - x*y + log(1+exp(-|x|))
    l = w * (fmaxf(x, 0.f) - x * y + log1pf(e));
    const float p = x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
    if (logit) logit[i] = x;
    if (prob) prob[i] = p;
    if (dlogit) dlogit[i] = w * (p - y);
  }
  // wave reduction, then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) l += __shfl_down(l, off, 64);
  if ((threadIdx.x & 63) == 0 && loss_sum) atomicAdd(loss_sum, l);
}

// loss_sum[0] = sum_b w_b * CE(logit_b, y_b), REPRODUCIBLE: one workgroup, thread t adds examples t, t + 1024, ... in
// that order, then a fixed-shape tree over the 1024 partials (the head kernels above add their per-wave partials with a
// float atomic, i.e. in arrival order).
__global__ void __launch_bounds__(1024)
k_bce_loss_sum(const float *__restrict__ logit, const float *__restrict__ labels, const float *__restrict__ weights,
               int64_t batch, float *__restrict__ loss_sum) {
  __shared__ float part[1024];
  float l = 0.f;
  // eight examples per lane and trip, their loads in flight together (one example per trip was a chain of batch / 1024
  // dependent round trips in a one-workgroup launch that sits on the step's critical path); the adds keep their order
  for (int64_t i0 = threadIdx.x; i0 < batch; i0 += 8 * 1024) {
    float x[8], y[8], w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t i = i0 + 1024 * k;
      const bool live = i < batch;
      x[k] = live ? logit[i] : 0.f;
      y[k] = live ? labels[i] : 0.f;
      w[k] = (live && weights) ? weights[i] : 1.0f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + 1024 * k < batch) l += w[k] * (fmaxf(x[k], 0.f) - x[k] * y[k] + log1pf(expf(-fabsf(x[k]))));
  }
  part[threadIdx.x] = l;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_sum[0] = part[0];
}

}  // namespace

extern "C" int wd_bce_loss_sum(const float *logit, const float *labels, const float *weights, int64_t batch,
                               float *loss_sum, wd_stream_t stream) {
  WD_REQUIRE(loss_sum, "null pointer");
  WD_REQUIRE(batch <= 0 || (logit && labels), "null pointer");
  hipLaunchKernelGGL(k_bce_loss_sum, dim3(1), dim3(1024), 0, wd::as_stream(stream), logit, labels, weights,
                     batch > 0 ? batch : 0, loss_sum);
  return wd::check_launch("wd_bce_loss_sum");
}

static int embag_fwd_impl(const float *emb, int64_t row_stride, const wd_slot_t *slots, int32_t S,
                          const int32_t *group_slots, int32_t ngroup, int32_t dim, const int32_t *ids,
                          const int32_t *bag_offs, int64_t batch, float *x, int64_t ldx, wd_stream_t stream,
                          float *wide_vals = nullptr) {
  if (batch <= 0 || ngroup <= 0) return WD_OK;
  WD_REQUIRE(emb && slots && group_slots && ids && bag_offs && x, "null pointer");
  WD_REQUIRE(dim > 0, "dim must be > 0");
  WD_REQUIRE(row_stride == dim || (row_stride > dim && row_stride % 4 == 0 && dim % 4 == 0 && dim <= 128),
             "row_stride must be dim, or a multiple of 4 above dim with dim in {4,...,128}");
  const int64_t RS4 = row_stride / 4;
  const int64_t nwork = batch * ngroup;
  hipStream_t st = wd::as_stream(stream);
#define WD_LAUNCH_EMBAG(L)                                                                                         \
  hipLaunchKernelGGL(k_embag_fwd<L>, dim3((unsigned)wd::ceil_div(nwork * L, 256)), dim3(256), 0, st, emb, slots, S, \
                     group_slots, ngroup, ids, bag_offs, nwork, x, ldx, RS4, (float *)nullptr)
#define WD_LAUNCH_EMBAG_W(L)                                                                                            \
  hipLaunchKernelGGL((k_embag_fwd<L, true>), dim3((unsigned)wd::ceil_div(nwork * L, 256)), dim3(256), 0, st, emb, slots, \
                     S, group_slots, ngroup, ids, bag_offs, nwork, x, ldx, RS4, wide_vals)
  if (wide_vals) {
    WD_REQUIRE((dim == 4 || dim == 8 || dim == 16) && row_stride >= dim + 4, "wide_vals: row records [dim | w z n -], dim in {4, 8, 16}");
    if (dim == 4) WD_LAUNCH_EMBAG_W(1);
    else if (dim == 8) WD_LAUNCH_EMBAG_W(2);
    else WD_LAUNCH_EMBAG_W(4);
    return wd::check_launch("wd_embag_fwd_wide");
  }
  switch (dim) {
    case 4: WD_LAUNCH_EMBAG(1); break;
    case 8: WD_LAUNCH_EMBAG(2); break;
    case 16: WD_LAUNCH_EMBAG(4); break;
    case 32: WD_LAUNCH_EMBAG(8); break;
    case 64: WD_LAUNCH_EMBAG(16); break;
    case 128: WD_LAUNCH_EMBAG(32); break;
    default:
      hipLaunchKernelGGL(k_embag_fwd_generic, dim3((unsigned)wd::ceil_div(nwork * dim, 256)), dim3(256), 0, st, emb,
                         slots, S, group_slots, ngroup, dim, ids, bag_offs, nwork, x, ldx);
  }
#undef WD_LAUNCH_EMBAG
#undef WD_LAUNCH_EMBAG_W
  return wd::check_launch("wd_embag_fwd");
}

extern "C" int wd_embag_fwd(const float *emb, const wd_slot_t *slots, int32_t S, const int32_t *group_slots,
                            int32_t ngroup, int32_t dim, const int32_t *ids, const int32_t *bag_offs, int64_t batch,
                            float *x, int64_t ldx, wd_stream_t stream) {
  return embag_fwd_impl(emb, dim, slots, S, group_slots, ngroup, dim, ids, bag_offs, batch, x, ldx, stream);
}

extern "C" int wd_embag_fwd_strided(const float *emb, int64_t row_stride, const wd_slot_t *slots, int32_t S,
                                    const int32_t *group_slots, int32_t ngroup, int32_t dim, const int32_t *ids,
                                    const int32_t *bag_offs, int64_t batch, float *x, int64_t ldx, wd_stream_t stream) {
  return embag_fwd_impl(emb, row_stride, slots, S, group_slots, ngroup, dim, ids, bag_offs, batch, x, ldx, stream);
}

extern "C" int wd_indicator_fwd(const wd_slot_t *slots, int32_t S, const int32_t *group_slots, int32_t ngroup,
                                const int32_t *ids, const int32_t *bag_offs, int64_t batch, float *x, int64_t ldx,
                                wd_stream_t stream) {
  if (batch <= 0 || ngroup <= 0) return WD_OK;
  WD_REQUIRE(slots && group_slots && ids && bag_offs && x, "null pointer");
  const int64_t nwork = batch * ngroup;
  hipLaunchKernelGGL(k_indicator_fwd, dim3((unsigned)wd::ceil_div(nwork * 16, 256)), dim3(256), 0, wd::as_stream(stream),
                     slots, S, group_slots, ngroup, ids, bag_offs, nwork, x, ldx);
  return wd::check_launch("wd_indicator_fwd");
}

extern "C" int wd_dense_fwd(const float *dense, int64_t ld_dense, const wd_dense_col_t *cols, int32_t ncols,
                            int64_t batch, float *x, int64_t ldx, wd_stream_t stream) {
  if (batch <= 0 || ncols <= 0) return WD_OK;
  WD_REQUIRE(dense && cols && x, "null pointer");
  hipLaunchKernelGGL(k_dense_fwd, dim3((unsigned)wd::ceil_div(batch * ncols, 256)), dim3(256), 0,
                     wd::as_stream(stream), dense, ld_dense, cols, ncols, batch, x, ldx);
  return wd::check_launch("wd_dense_fwd");
}

extern "C" int wd_embag_fwd_wide(const float *rec, int64_t rec_stride, const wd_slot_t *rec_slots, int32_t S,
                                 const int32_t *group_slots, int32_t ngroup, int32_t dim, const int32_t *ids,
                                 const int32_t *bag_offs, int64_t batch, float *x, int64_t ldx, float *wide_vals,
                                 wd_stream_t stream) {
  WD_REQUIRE(wide_vals, "null pointer");
  return embag_fwd_impl(rec, rec_stride, rec_slots, S, group_slots, ngroup, dim, ids, bag_offs, batch, x, ldx, stream, wide_vals);
}

extern "C" int wd_wide_sum(const float *wide_vals, const float *bias, const wd_slot_t *slots, int32_t S, int64_t batch, float *out,
                           wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(wide_vals && bias && slots && out && S > 0, "null pointer");
  hipLaunchKernelGGL(k_wide_sum, dim3((unsigned)wd::ceil_div(batch * 16, 256)), dim3(256), 0, wd::as_stream(stream), wide_vals, bias,
                     slots, S, batch, out);
  return wd::check_launch("wd_wide_sum");
}

extern "C" int wd_wide_fwd(const float *wide, int32_t wide_stride, const float *bias, const wd_slot_t *slots, int32_t S,
                           const int32_t *ids, const int32_t *bag_offs, int64_t batch, float *out, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(wide && bias && slots && ids && bag_offs && out, "null pointer");
  hipLaunchKernelGGL(k_wide_fwd, dim3((unsigned)wd::ceil_div(batch * 16, 256)), dim3(256), 0, wd::as_stream(stream),
                     wide, bias, slots, S, ids, bag_offs, batch, wide_stride, out);
  return wd::check_launch("wd_wide_fwd");
}

extern "C" int wd_bce_sum_fwd_bwd(const float *dnn_logit, const float *wide_logit, const float *labels,
                                  const float *weights, int64_t batch, float *logit, float *prob, float *dlogit,
                                  float *loss_sum, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(labels, "labels required");
  WD_REQUIRE(dnn_logit || wide_logit, "at least one logit source required");
  hipLaunchKernelGGL(k_bce, dim3((unsigned)wd::ceil_div(batch, 256)), dim3(256), 0, wd::as_stream(stream), dnn_logit,
                     wide_logit, labels, weights, batch, logit, prob, dlogit, loss_sum);
  return wd::check_launch("wd_bce_sum_fwd_bwd");
}

extern "C" int wd_embag_fwd_range(const float *emb, const wd_slot_t *slots, int32_t S, int32_t slot0, int32_t ngroup,
                                  int32_t dim, const int32_t *ids, const int32_t *bag_offs, int64_t batch, float *x,
                                  int64_t ldx, wd_stream_t stream) {
  if (batch <= 0 || ngroup <= 0) return WD_OK;
  WD_REQUIRE(emb && slots && ids && x, "null pointer");
  WD_REQUIRE(slot0 >= 0 && slot0 + ngroup <= S, "slot range out of bounds");
  WD_REQUIRE(ngroup <= 128, "at most 128 slots per dim group");
  const int64_t nwork = batch * ngroup;
  hipStream_t st = wd::as_stream(stream);
  constexpr int BPG = 2;
#define WD_LAUNCH_RANGE(L)                                                                                           \
  do {                                                                                                               \
    const dim3 grid((unsigned)wd::ceil_div(wd::ceil_div(nwork, BPG) * L, 256));                                      \
    if (bag_offs)                                                                                                    \
      hipLaunchKernelGGL((k_embag_fwd_range<L, BPG, false>), grid, dim3(256), 0, st, emb, slots, S, slot0, ngroup,   \
                         ids, bag_offs, nwork, x, ldx);                                                              \
    else                                                                                                             \
      hipLaunchKernelGGL((k_embag_fwd_range<L, BPG, true>), grid, dim3(256), 0, st, emb, slots, S, slot0, ngroup,    \
                         ids, bag_offs, nwork, x, ldx);                                                              \
  } while (0)
  switch (dim) {
    case 4: WD_LAUNCH_RANGE(1); break;
    case 8: WD_LAUNCH_RANGE(2); break;
    case 16: WD_LAUNCH_RANGE(4); break;
    case 32: WD_LAUNCH_RANGE(8); break;
    case 64: WD_LAUNCH_RANGE(16); break;
    case 128: WD_LAUNCH_RANGE(32); break;
    default:
      wd::set_error("wd_embag_fwd_range: dim %d is not one of 4,8,16,32,64,128 (use wd_embag_fwd)", dim);
      return WD_ERR_UNSUPPORTED;
  }
#undef WD_LAUNCH_RANGE
  return wd::check_launch("wd_embag_fwd_range");
}

extern "C" int wd_input_layer_fwd(const float *emb, const wd_slot_t *slots, int32_t S, int32_t slot0, int32_t ngroup,
                                  int32_t dim, const int32_t *ids, const int32_t *bag_offs, int32_t one_id_per_bag,
                                  int64_t batch, float *x, int64_t ldx, const float *dense, int64_t ld_dense,
                                  const wd_dense_col_t *dense_cols, int32_t ncols, const float *wide,
                                  const float *bias, float *wide_out, wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(emb && slots && ids && bag_offs && x, "null pointer");
  WD_REQUIRE(slot0 >= 0 && ngroup > 0 && slot0 + ngroup <= S && ngroup <= MAXG, "slot range out of bounds");
  WD_REQUIRE(!wide || (bias && wide_out), "wide part needs bias and out");
  WD_REQUIRE(ncols == 0 || (dense && dense_cols), "numeric part needs dense and dense_cols");
  constexpr int BPG = 2;
  InputLayerArgs a{};
  a.emb = emb; a.slots = slots; a.ids = ids; a.bag_offs = bag_offs; a.x = x; a.ldx = ldx; a.batch = batch;
  a.nwork = batch * ngroup; a.S = S; a.slot0 = slot0; a.ngroup = ngroup;
  a.wide = wide; a.bias = bias; a.wide_out = wide_out; a.wide_stride = 4;
  a.dense = dense; a.ld_dense = ld_dense; a.cols = dense_cols; a.ncols = ncols;
  a.nb_wide = wide ? (int)wd::ceil_div(batch * 16, 256) : 0;
  const int nb_dense = ncols > 0 ? (int)wd::ceil_div(batch * ncols, 256) : 0;
  hipStream_t st = wd::as_stream(stream);
#define WD_LAUNCH_IL(L)                                                                                          \
  do {                                                                                                           \
    a.nb_emb = (int)wd::ceil_div(wd::ceil_div(a.nwork, BPG) * L, 256);                                            \
    const dim3 grid((unsigned)(a.nb_emb + a.nb_wide + nb_dense));                                                 \
    if (one_id_per_bag) hipLaunchKernelGGL((k_input_layer<L, BPG, true>), grid, dim3(256), 0, st, a);             \
    else hipLaunchKernelGGL((k_input_layer<L, BPG, false>), grid, dim3(256), 0, st, a);                           \
  } while (0)
  switch (dim) {
    case 4: WD_LAUNCH_IL(1); break;
    case 8: WD_LAUNCH_IL(2); break;
    case 16: WD_LAUNCH_IL(4); break;
    case 32: WD_LAUNCH_IL(8); break;
    case 64: WD_LAUNCH_IL(16); break;
    case 128: WD_LAUNCH_IL(32); break;
    default:
      wd::set_error("wd_input_layer_fwd: dim %d is not one of 4,8,16,32,64,128", dim);
      return WD_ERR_UNSUPPORTED;
  }
#undef WD_LAUNCH_IL
  return wd::check_launch("wd_input_layer_fwd");
}
