// wide_deep_amd/csrc/mlp.hip -- dense tower of the train step on gfx950 (fp32 MFMA).
//
// Replaces the TF ops selected by python/lib/dnn.py:92-234 (tf.layers.dense + activation +
// tf.layers.batch_normalization, the six connection modes) and the dense half of
// `dnn_optimizer.minimize` (python/lib/joint.py:233-241, tf.train.AdagradOptimizer).
//
// Numerics: v_mfma_f32_32x32x2_f32 -- exact fp32 (bitwise an fmaf chain), so the tower keeps the
// reference's fp32 semantics; there is no TF32-like path on gfx950.
//
// BN in the reference is ALWAYS the inference-mode affine y = a*gamma/sqrt(1+eps) + beta
// (tf.layers.batch_normalization called without training=True).  Because it is a
// fixed per-column affine sitting between two GEMMs, it is folded into the CONSUMER's weights each
// step (wd_fold_affine) and un-folded from the weight gradient (wd_mlp_finalize); activations are
// stored once (post-activation, pre-affine) and never re-read for BN.
//
// One GEMM template serves the three products of a layer:
//   NN  a_l   = act(a_{l-1} Wf + bf)            A: reduction-contiguous (RC), B: output-contiguous (OC)
//   NT  da_{l-1} (+)= dz_l Wf^T                 A: RC, B: RC
//   TN  G     = [a_{l-1} | 1]^T dz_l  (split-K) A: OC, B: OC   (the appended ones row yields db)
// Tile 64x64, 64-deep slabs double-buffered in LDS, 4 waves (2x2) each owning one 32x32 accumulator; LDS tiles are
// reduction-major so the MFMA fragment reads are conflict-free ds_read_b32 rows; the global loads of slab i+1 stay in
// flight across the 32 MFMAs of slab i; workgroup ids are remapped so the N-tiles that re-read one A row-panel run on
// the same XCD (shared L2).  `simple` towers whose widths fit run through mlp_chain.hip instead (one launch for all NN /
// NT products + head); what they still take from this file: the fold, the grouped TN launch, the finalize.
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64;
constexpr int LD_RC = 65;  // reduction-contiguous operand: transposed through 4 scalar LDS stores (2-way conflicts, free)
constexpr int LD_OC = 68;  // output-contiguous operand: float4 LDS stores need 16-byte rows

struct GemmArgs {
  const float *A;
  const float *B;
  float *C;
  const float *bias;
  const float *act_src; // NT epilogue: C = (A B^T) * act'(act_src[m, n])  (NULL: plain)
  int64_t lda, ldb, ldc, ld_act;
  int64_t M, N, K;      // output M x N, reduction K
  int64_t kchunk;       // reduction slice per blockIdx.z
  int64_t c_split;      // element stride between split-K partial outputs
  int64_t ones_row;     // TN: output row index of A^T that is synthesised as all-ones (-1: none)
  int32_t act;
  int32_t accumulate;
  int32_t bias_parts;   // bias = sum of `bias_parts` vectors of stride N
  int32_t a_vec, b_vec; // float4 loads legal (ld % 4 == 0 and 16-byte aligned base)
  int32_t tiles_m, tiles_n;
  int32_t wt;           // TN partials stored write-through (common.h)
};

__device__ __forceinline__ float act_fwd(float v, int act) {
  switch (act) {
    case WD_ACT_RELU: return fmaxf(v, 0.f);
    case WD_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case WD_ACT_TANH: return tanhf(v);
    case WD_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
    case WD_ACT_LEAKY_RELU: return v > 0.f ? v : 0.2f * v;
    case WD_ACT_ELU: return v > 0.f ? v : expm1f(v);
    case WD_ACT_SELU: return v > 0.f ? 1.0507009873554805f * v : 1.0507009873554805f * 1.6732632423543772f * expm1f(v);
    case WD_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));
    case WD_ACT_SOFTSIGN: return v / (1.0f + fabsf(v));
    default: return v;
  }
}

// derivative expressed through the activation OUTPUT a
__device__ __forceinline__ float act_bwd(float a, int act) {
  switch (act) {
    case WD_ACT_RELU: return a > 0.f ? 1.f : 0.f;
    case WD_ACT_SIGMOID: return a * (1.f - a);
    case WD_ACT_TANH: return 1.f - a * a;
    case WD_ACT_RELU6: return (a > 0.f && a < 6.f) ? 1.f : 0.f;
    case WD_ACT_LEAKY_RELU: return a > 0.f ? 1.f : 0.2f;
    case WD_ACT_ELU: return a > 0.f ? 1.f : a + 1.f;
    case WD_ACT_SELU: return a > 0.f ? 1.0507009873554805f : a + 1.0507009873554805f * 1.6732632423543772f;
    case WD_ACT_SOFTPLUS: return 1.f - expf(-a);
    case WD_ACT_SOFTSIGN: { float t = 1.f - fabsf(a); return t * t; }
    default: return 1.f;
  }
}

// Branch-free tile loads: the address is clamped into the matrix (so the load is always legal and is issued
// unconditionally -> it stays in flight across the MFMA block), out-of-range elements are zeroed when the
// registers are written to LDS.  VEC: ld % 4 == 0 and a 16-byte aligned base, one 16-byte load; else 4 scalar loads.
template <bool VEC>
__device__ __forceinline__ float4 load4c(const float *__restrict__ P, int64_t row, int64_t col, int64_t ld, int64_t R,
                                         int64_t C) {
  const int64_t rc = row < R ? row : R - 1;
  if (VEC) {
    const int64_t cc = col < C ? col : 0;  // col % 4 == 0, C <= ld, ld % 4 == 0  =>  cc + 3 < ld
    return *reinterpret_cast<const float4 *>(P + rc * ld + cc);
  }
  const float *p = P + rc * ld;
  const int64_t cl = C - 1;
  float4 v;
  v.x = p[col < cl ? col : cl];
  v.y = p[col + 1 < cl ? col + 1 : cl];
  v.z = p[col + 2 < cl ? col + 2 : cl];
  v.w = p[col + 3 < cl ? col + 3 : cl];
  return v;
}

__device__ __forceinline__ float4 mask4(float4 v, int64_t row, int64_t col, int64_t R, int64_t C) {
  const bool r = row < R;
  v.x = (r && col < C) ? v.x : 0.f;
  v.y = (r && col + 1 < C) ? v.y : 0.f;
  v.z = (r && col + 2 < C) ? v.z : 0.f;
  v.w = (r && col + 3 < C) ? v.w : 0.f;
  return v;
}

// 64x64 output tile, BK-deep reduction slab, LDS double-buffered: the global loads of slab i+1 are issued
// before the BK/2 MFMAs of slab i and land in the other LDS buffer afterwards -> ONE barrier per slab and
// NV = BK/16 independent 16-byte loads per operand per lane in flight (the BK=16 single-stage version paid
// one full HBM/L2 latency per 8 MFMAs).
template <bool A_RC, bool B_RC, int EPI, int BK, bool VEC, bool XCD_REMAP = true>
__device__ __forceinline__ void gemm_body(const GemmArgs &g, const int orig, const int bz) {
  constexpr int LDA = A_RC ? LD_RC : LD_OC;
  constexpr int LDB = B_RC ? LD_RC : LD_OC;
  constexpr int NV = BK / 16;  // float4 per lane per operand per slab
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // Two workgroups share a CU, i.e. two waves share each SIMD's MFMA pipe.  With equal priority they fall into
  // lock-step (both in their MFMA phase at half rate, then both in their load/LDS-store phase with the pipe idle:
  // measured 46 % MFMA busy).  Give the wave in the odd hardware slot priority: it runs its MFMA phase at full
  // rate while the other one stores / waits, and vice versa.
  if (__builtin_amdgcn_s_getreg((4) | (0 << 6) | (3 << 11)) & 1u) __builtin_amdgcn_s_setprio(2);  // HW_ID.WAVE_ID

  // XCD-aware tile mapping: hardware places block i on XCD i % 8; give each XCD a contiguous run of
  // tiles (n fastest) so the N-tiles sharing an A panel hit the same L2.  Bijective for any grid.
  int vid = orig;       // !XCD_REMAP: the caller has placed the tile (k_gemm_tn_group: a whole split per XCD)
  if (XCD_REMAP) {
    const int nwg = g.tiles_m * g.tiles_n;
    const int q = nwg / wd::kXCDs, r = nwg % wd::kXCDs;
    const int xcd = orig % wd::kXCDs, loc = orig / wd::kXCDs;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int64_t m0 = (int64_t)(vid / g.tiles_n) * BM;
  const int64_t n0 = (int64_t)(vid % g.tiles_n) * BN;

  const int64_t kbeg = (int64_t)bz * g.kchunk;
  const int64_t kend = kbeg + g.kchunk < g.K ? kbeg + g.kchunk : g.K;
  const int64_t a_cols = g.ones_row >= 0 ? g.ones_row : g.M;  // TN: the appended ones row is synthesised

  // loader coordinates of float4 number v (idx = t + 256 v):
  //   RC: 64 out rows x BK reduction, BK/4 float4 per row      OC: BK reduction rows x 64 out, 16 float4 per row
  auto load_a = [&](int64_t k0, float4 (&ra)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = t + v * 256;
      if (A_RC) ra[v] = load4c<VEC>(g.A, m0 + idx / (BK / 4), k0 + (idx % (BK / 4)) * 4, g.lda, g.M, kend);
      else ra[v] = load4c<VEC>(g.A, k0 + (idx >> 4), m0 + (idx & 15) * 4, g.lda, kend, a_cols);
    }
  };
  auto load_b = [&](int64_t k0, float4 (&rb)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = t + v * 256;
      if (B_RC) rb[v] = load4c<VEC>(g.B, n0 + idx / (BK / 4), k0 + (idx % (BK / 4)) * 4, g.ldb, g.N, kend);
      else rb[v] = load4c<VEC>(g.B, k0 + (idx >> 4), n0 + (idx & 15) * 4, g.ldb, kend, g.N);
    }
  };
  // registers -> LDS (zeroing what lies outside the matrix)
  auto store_a = [&](float *S, int64_t k0, const float4 (&rv)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = t + v * 256;
      if (A_RC) {
        const int o = idx / (BK / 4), kr = (idx % (BK / 4)) * 4;
        const float4 x = mask4(rv[v], m0 + o, k0 + kr, g.M, kend);
        S[(kr + 0) * LD_RC + o] = x.x; S[(kr + 1) * LD_RC + o] = x.y;
        S[(kr + 2) * LD_RC + o] = x.z; S[(kr + 3) * LD_RC + o] = x.w;
      } else {
        const int rr = idx >> 4, cc = (idx & 15) * 4;
        float4 x = mask4(rv[v], k0 + rr, m0 + cc, kend, a_cols);
        if (g.ones_row >= 0 && k0 + rr < kend) {
          const int64_t c = m0 + cc;
          if (c == g.ones_row) x.x = 1.f;
          if (c + 1 == g.ones_row) x.y = 1.f;
          if (c + 2 == g.ones_row) x.z = 1.f;
          if (c + 3 == g.ones_row) x.w = 1.f;
        }
        *reinterpret_cast<float4 *>(&S[rr * LD_OC + cc]) = x;
      }
    }
  };
  auto store_b = [&](float *S, int64_t k0, const float4 (&rv)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = t + v * 256;
      if (B_RC) {
        const int o = idx / (BK / 4), kr = (idx % (BK / 4)) * 4;
        const float4 x = mask4(rv[v], n0 + o, k0 + kr, g.N, kend);
        S[(kr + 0) * LD_RC + o] = x.x; S[(kr + 1) * LD_RC + o] = x.y;
        S[(kr + 2) * LD_RC + o] = x.z; S[(kr + 3) * LD_RC + o] = x.w;
      } else {
        const int rr = idx >> 4, cc = (idx & 15) * 4;
        *reinterpret_cast<float4 *>(&S[rr * LD_OC + cc]) = mask4(rv[v], k0 + rr, n0 + cc, kend, g.N);
      }
    }
  };

  // Fast path (interior tile, full slab, 16-byte loads): no clamps, no masks, addresses = uniform slab pointer +
  // per-lane 32-bit offsets computed once.  Both operand orientations share one offset formula because a
  // 64 x BK (RC) and a BK x 64 (OC) slab are both 16 float4 wide at BK = 64.
  static_assert(BK == 64 || (BK % 16 == 0 && !A_RC && !B_RC), "fast-path offsets: BK == 64, or any 16-multiple for two OC operands");
  const bool fast_tile = VEC && (m0 + BM <= (A_RC ? g.M : a_cols)) && (n0 + BN <= g.N);
  int32_t offA[NV], offB[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    offA[v] = ((t >> 4) + 16 * v) * (int32_t)g.lda + (t & 15) * 4;
    offB[v] = ((t >> 4) + 16 * v) * (int32_t)g.ldb + (t & 15) * 4;
  }
  const float *pa0 = A_RC ? g.A + m0 * g.lda : g.A + m0;  // tile origin, reduction index 0
  const float *pb0 = B_RC ? g.B + n0 * g.ldb : g.B + n0;
  const int64_t astep = A_RC ? 1 : g.lda, bstep = B_RC ? 1 : g.ldb;
  const int lds_rc = (t & 15) * 4 * LD_RC + (t >> 4);  // + i * LD_RC + 16 v
  const int lds_oc = (t >> 4) * LD_OC + (t & 15) * 4;  // + 16 v * LD_OC
  auto load_fast = [&](int64_t k0, float4 (&ra)[NV], float4 (&rb)[NV]) {
    const float *pa = pa0 + k0 * astep, *pb = pb0 + k0 * bstep;
#pragma unroll
    for (int v = 0; v < NV; ++v) ra[v] = *reinterpret_cast<const float4 *>(pa + offA[v]);
#pragma unroll
    for (int v = 0; v < NV; ++v) rb[v] = *reinterpret_cast<const float4 *>(pb + offB[v]);
  };
  auto store_fast = [&](float *Sa, float *Sb, const float4 (&ra)[NV], const float4 (&rb)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (A_RC) {
        float *q = Sa + lds_rc + 16 * v;
        q[0] = ra[v].x; q[LD_RC] = ra[v].y; q[2 * LD_RC] = ra[v].z; q[3 * LD_RC] = ra[v].w;
      } else {
        *reinterpret_cast<float4 *>(Sa + lds_oc + 16 * v * LD_OC) = ra[v];
      }
      if (B_RC) {
        float *q = Sb + lds_rc + 16 * v;
        q[0] = rb[v].x; q[LD_RC] = rb[v].y; q[2 * LD_RC] = rb[v].z; q[3 * LD_RC] = rb[v].w;
      } else {
        *reinterpret_cast<float4 *>(Sb + lds_oc + 16 * v * LD_OC) = rb[v];
      }
    }
  };

  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  const int fr = (lane >> 5), fc = (lane & 31);
  // one slab of MFMAs from LDS buffer `cur`
  auto compute = [&](int cur) {
    const float *Ac = As[cur] + fr * LDA + wm * 32 + fc;
    const float *Bc = Bs[cur] + fr * LDB + wn * 32 + fc;
    // software-pipelined fragment reads: the LDS reads of group gi+1 are issued BEFORE the MFMAs of group gi
    // (sched_barrier pins the order; left alone the scheduler sinks each read next to its MFMA and the
    // dependent MFMA chain then waits one LDS latency per pair)
    constexpr int FG = 8, NG = BK / 2 / FG;
    float fa[2][FG], fb[2][FG];
#pragma unroll
    for (int j = 0; j < FG; ++j) {
      fa[0][j] = Ac[2 * j * LDA];
      fb[0][j] = Bc[2 * j * LDB];
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) {
#pragma unroll
        for (int j = 0; j < FG; ++j) {
          fa[(gi + 1) & 1][j] = Ac[2 * ((gi + 1) * FG + j) * LDA];
          fb[(gi + 1) & 1][j] = Bc[2 * ((gi + 1) * FG + j) * LDB];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < FG; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gi & 1][j], fb[gi & 1][j], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  {
    float4 ra[NV], rb[NV];
    bool fast = fast_tile && (kbeg + BK <= kend);
    if (fast) {
      load_fast(kbeg, ra, rb);
      store_fast(As[0], Bs[0], ra, rb);
    } else {
      load_a(kbeg, ra);
      load_b(kbeg, rb);
      store_a(As[0], kbeg, ra);
      store_b(Bs[0], kbeg, rb);
    }
    __syncthreads();
    int cur = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = k0 + BK < kend;
      fast = fast_tile && (k0 + 2 * BK <= kend);  // the slab being prefetched is full
      if (more) {
        if (fast) {
          load_fast(k0 + BK, ra, rb);
        } else {
          load_a(k0 + BK, ra);
          load_b(k0 + BK, rb);
        }
      }
      compute(cur);
      if (more) {
        if (fast) {
          store_fast(As[cur ^ 1], Bs[cur ^ 1], ra, rb);
        } else {
          store_a(As[cur ^ 1], k0 + BK, ra);
          store_b(Bs[cur ^ 1], k0 + BK, rb);
        }
      }
      __syncthreads();
      cur ^= 1;
    }
  }

  // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int64_t n = n0 + wn * 32 + fc;
  if (n >= g.N) return;
  float *Cz = g.C + (EPI == 2 ? (int64_t)bz * g.c_split : 0);
  float bv = 0.f;
  if (EPI == 0 && g.bias) {
    for (int p = 0; p < g.bias_parts; ++p) bv += g.bias[(int64_t)p * g.N + n];
  }
#pragma unroll
  for (int rg = 0; rg < 16; ++rg) {
    const int64_t m = m0 + wm * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * fr;
    if (m >= g.M) continue;
    float v = acc[rg];
    if (EPI == 0) {
      v = act_fwd(v + bv, g.act);
    } else if (EPI == 1) {
      if (g.accumulate) v += Cz[m * g.ldc + n];
      if (g.act_src) v *= act_bwd(g.act_src[m * g.ld_act + n], g.act);
    }
    if (EPI == 2) wd::store1(&Cz[m * g.ldc + n], v, g.wt);
    else Cz[m * g.ldc + n] = v;
  }
}

template <bool A_RC, bool B_RC, int EPI, int BK, bool VEC>
__global__ void __launch_bounds__(256) k_gemm(GemmArgs g) {
  gemm_body<A_RC, B_RC, EPI, BK, VEC>(g, blockIdx.x, blockIdx.z);
}

// The weight-gradient products of ALL layers in one launch (wd_gemm_tn_splitk_group): the flat grid is cut into
// per-job ranges of tiles x splits; each workgroup runs the same body as a stand-alone launch would.  Three launches of
// 10-30 us, each paying the ~11 us fixed cost of a kernel boundary, become one.
constexpr int MAX_TN_JOBS = WD_TN_GROUP_MAX;
struct GroupArgs {
  GemmArgs job[MAX_TN_JOBS];
  int32_t first[MAX_TN_JOBS + 1];   // first flat workgroup id of job j (multiple of 8: keeps the XCD-aware tile order
                                    // of the body valid); [njobs] = grid size
  int32_t count[MAX_TN_JOBS];       // workgroups of job j (tiles x splits; split_xcd: tiles x 8 x ceil(splits / 8))
  int32_t colsum[MAX_TN_JOBS];      // job j is a column-sum job (B == NULL)
  int32_t nsplit[MAX_TN_JOBS];
  int32_t njobs;
};

template <int BK, bool VEC>
__global__ void __launch_bounds__(256) k_gemm_tn_group(GroupArgs G) {
  int j = 0;
  while (j + 1 < G.njobs && (int)blockIdx.x >= G.first[j + 1]) ++j;
  j = __builtin_amdgcn_readfirstlane(j);
  const GemmArgs &g = G.job[j];
  const int rem = blockIdx.x - G.first[j];
  if (rem >= G.count[j]) return;   // padding of the job's range
  if (G.colsum[j]) {
    // C[n] = sum_k A[k][n]: 64 columns per workgroup, 4 row groups (k = q, q+4, ..) with 16 independent loads in flight
    // each, combined in group order -- a fixed summation order, and no chain of exposed load latencies
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t n = (int64_t)rem * 64 + c;
    float acc = 0.f;
    if (n < g.N) {
      for (int64_t k0 = q; k0 < g.K; k0 += 64) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = k0 + 4 * u < g.K ? g.A[(k0 + 4 * u) * g.lda + n] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
      }
    }
    part[q][c] = acc;
    __syncthreads();
    if (q == 0 && n < g.N) g.C[n] = part[0][c] + part[1][c] + part[2][c] + part[3][c];
    return;
  }
  const int nwg = g.tiles_m * g.tiles_n;
  // Every tile of a split reads the same kchunk rows of A and B: with the whole split on one XCD (hardware: workgroup i runs on
  // XCD i % 8, and a job's range starts at a multiple of 8) its L2 fetches each panel once -- 1.4 MB per split of C2's first
  // layer -- instead of once per XCD that holds some of its tiles.  (ONE instantiation of the body: a second one -- the tiles of
  // a split spread over the XCDs, measured 1 % slower and switched off since round 4 -- carried its own static LDS slabs, 70 KB
  // per workgroup where the kernel uses 35: two workgroups per CU instead of four, and 19 KB left for the row update beside it.)
  const int xcd = rem % wd::kXCDs, loc = rem / wd::kXCDs;
  const int bz = xcd + wd::kXCDs * (loc / nwg);
  if (bz >= G.nsplit[j]) return;
  gemm_body<false, false, 2, BK, VEC, false>(g, loc % nwg, bz);
}


inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int kBK = 64;

template <bool A_RC, bool B_RC, int EPI>
int launch_gemm(GemmArgs g, int nsplit, hipStream_t st, const char *what) {
  g.tiles_m = (int)wd::ceil_div(g.M, BM);
  g.tiles_n = (int)wd::ceil_div(g.N, BN);
  g.a_vec = (g.lda % 4 == 0) && aligned16(g.A);
  g.b_vec = (g.ldb % 4 == 0) && aligned16(g.B);
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)nsplit);
  if (g.a_vec && g.b_vec) hipLaunchKernelGGL((k_gemm<A_RC, B_RC, EPI, kBK, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((k_gemm<A_RC, B_RC, EPI, kBK, false>), grid, dim3(256), 0, st, g);
  return wd::check_launch(what);
}

// ---- small per-layer kernels -------------------------------------------------------------------

// s[k] = gamma*inv (or 1), t[k] = beta (or 0);  Wf = diag(s) W;  bf_part[c] = (c == 0 ? b : 0) + t_c^T W_c
// grid (ceil(N/64), WD_FOLD_PARTS): block = 64 columns x 4 k-lanes over one K chunk; the partial bias
// sums stay separate per chunk (deterministic) and are added up by the GEMM epilogue.
constexpr int FOLD_TK = 256;      // rows of a K chunk whose transposed half copy is staged in LDS
constexpr int FOLD_TP = FOLD_TK + 2;   // pitch (halfs): odd in dwords -> the column-wise LDS writes do not conflict

// Optional half outputs for the fp16-input tower (mlp_half.hip):
//   WfT_h [N][ld_wft_h]  transposed folded kernel (NN operand), written through an LDS tile so that the stores are
//                        128-byte rows instead of 2-byte scatters (the scattered version took 112 us at C5)
//   wcat + cat_off[k]    row k of this layer, as the columns [.. + n] of the "pull" backward operand of the segment
//                        that produced input column k (see WideDeepEngine._tower_backward_h); cat_off[k] < 0: skip
__device__ __forceinline__ void fold_affine_body(const float *__restrict__ P, int64_t w_off, int64_t b_off,
                                                 const int32_t *__restrict__ gamma_idx,
                                                 const int32_t *__restrict__ beta_idx, float inv,
                                                 float *__restrict__ Wf, float *__restrict__ bf,
                                                 float *__restrict__ s_out, float *__restrict__ t_out, int64_t K,
                                                 int64_t N, int bx, int by, float (*red)[64],
                                                 _Float16 *__restrict__ WfT_h = nullptr, int64_t ld_wft_h = 0,
                                                 const int64_t *__restrict__ cat_off = nullptr,
                                                 _Float16 *__restrict__ wcat = nullptr, _Float16 *tileT = nullptr) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t n = (int64_t)bx * 64 + tx;
  const int64_t kc = (K + WD_FOLD_PARTS - 1) / WD_FOLD_PARTS;
  const int64_t k0 = (int64_t)by * kc;
  const int64_t k1 = k0 + kc < K ? k0 + kc : K;
  const float *W = P + w_off;
  const bool stage = WfT_h && tileT && kc <= FOLD_TK;
  float tb = 0.f;
  // U rows per trip, in three rounds of independent loads (indices -> affine parameters + kernel row -> stores): the
  // per-row chain idx -> P[idx] -> store used to cost one exposed round trip per row (128 us at C5's 5 M weights)
#ifndef WD_FOLD_U
#define WD_FOLD_U 8
#endif
  constexpr int U = WD_FOLD_U;      // 8: layers with thousands of rows and few column tiles (C5: K 3469 x N 128 = 32 workgroups
                                    // of 217 rows) are a serial chain of trips; 4 -> 8 rows per trip halves it
  for (int64_t kb = k0 + ty; kb < k1; kb += 4 * U) {
    int32_t gi[U], bi[U];
    int64_t co[U];
    float w[U], sk[U], tk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = kb + 4 * u;
      const bool live = k < k1;
      gi[u] = (live && gamma_idx) ? gamma_idx[k] : -1;
      bi[u] = (live && beta_idx) ? beta_idx[k] : -1;
      co[u] = (live && cat_off) ? cat_off[k] : -1;
      w[u] = (live && n < N) ? W[k * N + n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sk[u] = gi[u] >= 0 ? P[gi[u]] * inv : 1.0f;
      tk[u] = bi[u] >= 0 ? P[bi[u]] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = kb + 4 * u;
      if (k >= k1) break;
      if (bx == 0 && tx == 0) {
        s_out[k] = sk[u];
        t_out[k] = tk[u];
      }
      if (n < N) {
        const float v = sk[u] * w[u];
        Wf[k * N + n] = v;
        if (WfT_h) {
          if (stage) tileT[tx * FOLD_TP + (k - k0)] = (_Float16)v;
          else WfT_h[n * ld_wft_h + k] = (_Float16)v;
        }
        if (co[u] >= 0) wcat[co[u] + n] = (_Float16)v;
        tb += tk[u] * w[u];
      }
    }
  }
  red[ty][tx] = tb;
  __syncthreads();
  if (ty == 0 && n < N) {
    float v = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
    if (by == 0) v += P[b_off + n];
    bf[(int64_t)by * N + n] = v;
  }
  if (stage) {   // tile -> WfT_h: each wavefront writes one row segment (k contiguous) at a time
    const int kcn = (int)(k1 - k0);
    for (int nl = ty; nl < 64; nl += 4) {
      const int64_t nn = (int64_t)bx * 64 + nl;
      if (nn >= N) break;
      for (int kk = tx; kk < kcn; kk += 64) WfT_h[nn * ld_wft_h + k0 + kk] = tileT[nl * FOLD_TP + kk];
    }
  }
}

__global__ void __launch_bounds__(256)
k_fold_affine(const float *__restrict__ P, int64_t w_off, int64_t b_off, const int32_t *__restrict__ gamma_idx,
              const int32_t *__restrict__ beta_idx, float inv, float *__restrict__ Wf, float *__restrict__ bf,
              float *__restrict__ s_out, float *__restrict__ t_out, int64_t K, int64_t N) {
  __shared__ float red[4][64];
  fold_affine_body(P, w_off, b_off, gamma_idx, beta_idx, inv, Wf, bf, s_out, t_out, K, N, blockIdx.x, blockIdx.y, red);
}

// every layer of every tower in ONE launch: blockIdx.z = layer (descriptor table in HBM); also clears the
// step's small accumulators (loss, optional flat gradient buffer) so that the step needs no separate fills.
__global__ void __launch_bounds__(256)
k_fold_affine_all(const float *__restrict__ P, const wd_mlp_layer_t *__restrict__ layers, float inv,
                  float *__restrict__ zero_a, int64_t zero_a_n, float *__restrict__ zero_b, int64_t zero_b_n) {
  __shared__ float red[4][64];
  __shared__ _Float16 tileT[64 * FOLD_TP];
  const int64_t bid = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * gridDim.y * gridDim.z * 256;
  for (int64_t i = bid * 256 + threadIdx.x; i < zero_a_n; i += nthreads) zero_a[i] = 0.f;
  for (int64_t i = bid * 256 + threadIdx.x; i < zero_b_n; i += nthreads) zero_b[i] = 0.f;
  const wd_mlp_layer_t L = layers[blockIdx.z];
  if ((int64_t)blockIdx.x * 64 >= L.N) return;
  fold_affine_body(P, L.w_off, L.b_off, L.gamma_idx, L.beta_idx, inv, L.Wf, L.bf, L.s, L.t, L.K, L.N, blockIdx.x,
                   blockIdx.y, red, reinterpret_cast<_Float16 *>(L.WfT_h), L.ld_wft_h, L.cat_off,
                   reinterpret_cast<_Float16 *>(L.wcat), tileT);
}

__global__ void __launch_bounds__(256)
k_act_bwd(const float *__restrict__ da, int64_t ldda, const float *__restrict__ a, int64_t lda, int32_t act,
          float *__restrict__ dz, int64_t lddz, int64_t M, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  dz[m * lddz + n] = da[m * ldda + n] * act_bwd(a[m * lda + n], act);
}

// ---- dropout (python/lib/dnn.py:111-112 etc.: tf.layers.dropout(net, rate, training=True) after the activation, before
// BN, TRAIN mode only).  keep(b, n) = u >= rate with u a counter-based uniform of (seed, step, layer, b*N + n) -- the
// same function is evaluated again in the backward pass (and on the host by the tests), no mask is stored.
// TF: ret = x / keep_prob * binary_tensor.
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t step, uint32_t layer, uint64_t idx, float rate) {
  uint64_t z = seed + step * 0x632BE59BD9B4E019ull + (uint64_t)(layer + 1) * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);   // 24 bits -> [0, 1)
  return u >= rate;
}

__global__ void __launch_bounds__(256)
k_dropout_fwd(float *__restrict__ a, int64_t lda, int64_t M, int64_t N, float rate, const int64_t *__restrict__ seed,
              int32_t layer) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  const float keep_prob = 1.0f - rate;
  const float v = a[m * lda + n];
  a[m * lda + n] = dropout_keep((uint64_t)seed[0], (uint64_t)seed[1], layer, (uint64_t)i, rate) ? v / keep_prob : 0.f;
}

// dz = da * d(dropout(act(z)))/dz with a_drop the stored (post-dropout) activation
__global__ void __launch_bounds__(256)
k_act_bwd_dropout(const float *__restrict__ da, int64_t ldda, const float *__restrict__ a, int64_t lda, int32_t act,
                  float *__restrict__ dz, int64_t lddz, int64_t M, int64_t N, float rate,
                  const int64_t *__restrict__ seed, int32_t layer) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  const float keep_prob = 1.0f - rate;
  float v = 0.f;
  if (dropout_keep((uint64_t)seed[0], (uint64_t)seed[1], layer, (uint64_t)i, rate))
    v = da[m * ldda + n] / keep_prob * act_bwd(a[m * lda + n] * keep_prob, act);
  dz[m * lddz + n] = v;
}

__global__ void k_counter_tick(int64_t *c) { c[1] += 1; }

// One block per input column k (row of W): reduce the split-K partials, emit dW row, and the
// affine-parameter gradients of the producer of column k.  `store_affine`: the producer has exactly one
// consumer (simple mode) -> plain store; otherwise accumulate (launches are stream-ordered and k is unique
// within a launch, so no atomics are needed).
__device__ __forceinline__ void mlp_finalize_body(const float *__restrict__ Gpart, int32_t nsplit,
                                                  const float *P /* may alias Pw */, int64_t w_off, int64_t b_off,
                                                  const float *__restrict__ s, const float *__restrict__ t,
                                                  const int32_t *__restrict__ gamma_idx,
                                                  const int32_t *__restrict__ beta_idx, float inv,
                                                  float *__restrict__ Gflat, int64_t K, int64_t N, int64_t k,
                                                  bool store_affine, float *red_g, float *red_d,
                                                  float *Pw = nullptr, float *__restrict__ Pacc = nullptr,
                                                  float lr = 0.f) {
  // Pw / Pacc: apply dense Adagrad to every parameter right where its gradient is final (single-GPU default
  // optimizer: saves the separate elementwise launch).  W is read before it is updated by the same lane.
  auto apply = [&](int64_t idx, float gval) {
    if (Pacc) {
      const float a = Pacc[idx] + gval * gval;
      Pacc[idx] = a;
      Pw[idx] -= lr * gval / sqrtf(a);
    }
  };
  // thread (nx, zq): column nx (+ j*NT), splits zq, zq+ZQ, ...; the ZQ partial sums are combined in fixed order
  const int NT = N >= 256 ? 256 : (N > 128 ? 256 : (N > 64 ? 128 : (N > 32 ? 64 : (N > 16 ? 32 : 16))));
  const int ZQ = 256 / NT;
  const int nx = threadIdx.x % NT, zq = threadIdx.x / NT;
  const int64_t split_stride = (K + 1) * N;
  const float *W = P + w_off;
  float acc_s = 0.f, acc_t = 0.f;
  for (int64_t n0 = 0; n0 < N; n0 += NT) {
    const int64_t n = n0 + nx;
    float gk = 0.f, db = 0.f;
    if (n < N) {
#pragma unroll 4
      for (int32_t z = zq; z < nsplit; z += ZQ) {
        gk += Gpart[z * split_stride + k * N + n];
        db += Gpart[z * split_stride + K * N + n];
      }
    }
    if (ZQ > 1) {
      __syncthreads();
      red_g[zq * NT + nx] = gk;
      red_d[zq * NT + nx] = db;
      __syncthreads();
      if (zq == 0) {
        for (int j = 1; j < ZQ; ++j) {
          gk += red_g[j * NT + nx];
          db += red_d[j * NT + nx];
        }
      }
    }
    if (zq == 0 && n < N) {
      if (k == K) {
        Gflat[b_off + n] = db;
        apply(b_off + n, db);
      } else {
        const float gw = s[k] * gk + t[k] * db;
        Gflat[w_off + k * N + n] = gw;
        const float w = W[k * N + n];
        acc_s += w * gk;
        acc_t += w * db;
        apply(w_off + k * N + n, gw);
      }
    }
  }
  if (k == K) return;
  const int32_t gi = gamma_idx ? gamma_idx[k] : -1;
  const int32_t bi = beta_idx ? beta_idx[k] : -1;
  if (gi < 0 && bi < 0) return;
  // block reduction of (acc_s, acc_t) in fixed order (only zq == 0 threads hold non-zero values)
  for (int off = 32; off > 0; off >>= 1) {
    acc_s += __shfl_down(acc_s, off, 64);
    acc_t += __shfl_down(acc_t, off, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red_g[threadIdx.x >> 6] = acc_s;
    red_d[threadIdx.x >> 6] = acc_t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float ss = red_g[0] + red_g[1] + red_g[2] + red_g[3];
    const float tt = red_d[0] + red_d[1] + red_d[2] + red_d[3];
    if (store_affine) {
      if (gi >= 0) { Gflat[gi] = ss * inv; apply(gi, ss * inv); }
      if (bi >= 0) { Gflat[bi] = tt; apply(bi, tt); }
    } else {
      if (gi >= 0) Gflat[gi] += ss * inv;
      if (bi >= 0) Gflat[bi] += tt;
    }
  }
}

__global__ void __launch_bounds__(256)
k_mlp_finalize(const float *__restrict__ Gpart, int32_t nsplit, const float *__restrict__ P, int64_t w_off,
               int64_t b_off, const float *__restrict__ s, const float *__restrict__ t,
               const int32_t *__restrict__ gamma_idx, const int32_t *__restrict__ beta_idx, float inv,
               float *__restrict__ Gflat, int64_t K, int64_t N) {
  __shared__ float red_g[256], red_d[256];
  mlp_finalize_body(Gpart, nsplit, P, w_off, b_off, s, t, gamma_idx, beta_idx, inv, Gflat, K, N, blockIdx.x, false,
                    red_g, red_d);
}

// N % 4 == 0: the same reduction with 16-byte loads -- thread (vx, zq): columns 4 vx .. 4 vx + 3, splits zq, zq + ZQ, ...
// (N = 1024: ONE pass over the row instead of four dependent ones; 24 -> ~10 us per layer at C5).
__global__ void __launch_bounds__(256)
k_mlp_finalize4(const float *__restrict__ Gpart, int32_t nsplit, const float *__restrict__ P, int64_t w_off,
                int64_t b_off, const float *__restrict__ s, const float *__restrict__ t,
                const int32_t *__restrict__ gamma_idx, const int32_t *__restrict__ beta_idx, float inv,
                float *__restrict__ Gflat, int64_t K, int64_t N) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ f4 red_g[256], red_d[256];
  __shared__ float red_s[4], red_t[4];
  const int64_t k = blockIdx.x;
  const int NV = (int)(N >> 2);
  int NT = 16;
  while (NT < NV && NT < 256) NT <<= 1;
  const int ZQ = 256 / NT;
  const int vx = threadIdx.x % NT, zq = threadIdx.x / NT;
  const int64_t split_stride = (K + 1) * N;
  float acc_s = 0.f, acc_t = 0.f;
  const float sk = k < K ? s[k] : 0.f, tk = k < K ? t[k] : 0.f;
  for (int v0 = 0; v0 < NV; v0 += NT) {
    const int v = v0 + vx;
    f4 gk = (f4)(0.f), db = (f4)(0.f);
    if (v < NV) {
#pragma unroll 4
      for (int32_t z = zq; z < nsplit; z += ZQ) {
        gk += *reinterpret_cast<const f4 *>(Gpart + z * split_stride + k * N + 4 * v);
        db += *reinterpret_cast<const f4 *>(Gpart + z * split_stride + K * N + 4 * v);
      }
    }
    if (ZQ > 1) {
      __syncthreads();
      red_g[zq * NT + vx] = gk;
      red_d[zq * NT + vx] = db;
      __syncthreads();
      if (zq == 0) {
        for (int j = 1; j < ZQ; ++j) {
          gk += red_g[j * NT + vx];
          db += red_d[j * NT + vx];
        }
      }
    }
    if (zq == 0 && v < NV) {
      if (k == K) {
        *reinterpret_cast<f4 *>(Gflat + b_off + 4 * v) = db;
      } else {
        *reinterpret_cast<f4 *>(Gflat + w_off + k * N + 4 * v) = sk * gk + tk * db;
        const f4 w = *reinterpret_cast<const f4 *>(P + w_off + k * N + 4 * v);
        acc_s += w.x * gk.x; acc_s += w.y * gk.y; acc_s += w.z * gk.z; acc_s += w.w * gk.w;
        acc_t += w.x * db.x; acc_t += w.y * db.y; acc_t += w.z * db.z; acc_t += w.w * db.w;
      }
    }
  }
  if (k == K) return;
  const int32_t gi = gamma_idx ? gamma_idx[k] : -1;
  const int32_t bi = beta_idx ? beta_idx[k] : -1;
  if (gi < 0 && bi < 0) return;
  for (int off = 32; off > 0; off >>= 1) {
    acc_s += __shfl_down(acc_s, off, 64);
    acc_t += __shfl_down(acc_t, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red_s[threadIdx.x >> 6] = acc_s;
    red_t[threadIdx.x >> 6] = acc_t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float ss = red_s[0] + red_s[1] + red_s[2] + red_s[3];
    const float tt = red_t[0] + red_t[1] + red_t[2] + red_t[3];
    if (gi >= 0) Gflat[gi] += ss * inv;
    if (bi >= 0) Gflat[bi] += tt;
  }
}

// N == 1 (the logits layer; its partials come one per 16 / 64 examples from wd_logits_head: hundreds of splits): the row-per-
// workgroup geometry above reads one 4-byte value per 64-byte line (72 us at C5: 3470 rows x 512 splits).  Here a workgroup
// owns 16 consecutive rows k, thread (kx, zq) sums splits zq, zq + 16, ... of row k0 + kx -- 64-byte segments per split --
// and the 16 partial sums are combined in the same fixed order as above (bit-identical results).
__global__ void __launch_bounds__(256)
k_mlp_finalize_vec(const float *__restrict__ Gpart, int32_t nsplit, const float *__restrict__ P, int64_t w_off,
                   int64_t b_off, const float *__restrict__ s, const float *__restrict__ t,
                   const int32_t *__restrict__ gamma_idx, const int32_t *__restrict__ beta_idx, float inv,
                   float *__restrict__ Gflat, int64_t K) {
  __shared__ float red_g[256], red_d[256];
  const int kx = threadIdx.x & 15, zq = threadIdx.x >> 4;
  const int64_t k = (int64_t)blockIdx.x * 16 + kx;
  const int64_t stride = K + 1;
  float gk = 0.f, db = 0.f;
  if (k <= K) {
#pragma unroll 8
    for (int32_t z = zq; z < nsplit; z += 16) {
      gk += Gpart[z * stride + k];
      db += Gpart[z * stride + K];
    }
  }
  red_g[zq * 16 + kx] = gk;
  red_d[zq * 16 + kx] = db;
  __syncthreads();
  if (zq != 0 || k > K) return;
  for (int j = 1; j < 16; ++j) {
    gk += red_g[j * 16 + kx];
    db += red_d[j * 16 + kx];
  }
  if (k == K) {
    Gflat[b_off] = db;
    return;
  }
  Gflat[w_off + k] = s[k] * gk + t[k] * db;
  const float w = P[w_off + k];
  const int32_t gi = gamma_idx ? gamma_idx[k] : -1;
  const int32_t bi = beta_idx ? beta_idx[k] : -1;
  if (gi >= 0) Gflat[gi] += (w * gk) * inv;
  if (bi >= 0) Gflat[bi] += w * db;
}

// all layers in one launch (blockIdx.y = layer): legal only when every BN gamma/beta has ONE consumer layer
__global__ void __launch_bounds__(256)
k_mlp_finalize_all(const wd_mlp_layer_t *__restrict__ layers, const float *P, float inv, float *__restrict__ Gflat,
                   float *Pw, float *__restrict__ Pacc, float lr) {
  __shared__ float red_g[256], red_d[256];
  const wd_mlp_layer_t L = layers[blockIdx.y];
  if ((int64_t)blockIdx.x > L.K) return;
  mlp_finalize_body(L.Gpart, L.nsplit, P, L.w_off, L.b_off, L.s, L.t, L.gamma_idx, L.beta_idx, inv, Gflat, L.K, L.N,
                    blockIdx.x, true, red_g, red_d, Pw, Pacc, lr);
}

// ---- logits layer + head, forward AND backward of that layer in one launch ------------------------
// (python/lib/dnn.py:226-232 logits dense(units=1); python/lib/joint.py:216-222 add_n; head joint.py:264-269.)
// Block `blk` owns HEAD_CHUNK examples:
//   phase 1 (wave per example): dnn_logit = a[b, window] . wf + bias; logit = dnn + wide; CE loss, p, dlogit
//   phase 2 (lane per input column k): gradient wrt the window  out[b,k] = dlogit[b]*wf[k] (* act'(a[b,k]) when
//           `act` != 0: simple mode, `out` is then dz of the last hidden layer), and this block's partial of the
//           kernel gradient  Gpart[blk][k] = sum_b a[b,k]*dlogit[b],  Gpart[blk][K] = sum_b dlogit[b]  (fixed order).
// Two geometries: narrow logits layers (K <= 256: simple / resnet-lite towers) take 64 examples per workgroup with 4
// lanes per example; wide ones (dense / last_dense windows, K in the thousands) take 16 examples per workgroup with a
// whole wavefront per example, so that the row reads are 128-byte coalesced and 4x more workgroups are in flight.
constexpr int HEAD_K_WIDE = 256;
__host__ __device__ constexpr int head_chunk(int64_t K) { return K > HEAD_K_WIDE ? 16 : 64; }
template <typename TA, int HEAD_CHUNK, int HEAD_LPE>
__global__ void __launch_bounds__(256)
k_logits_head(const TA *__restrict__ a, int64_t ld_a, int64_t K, const float *__restrict__ wf,
              const float *__restrict__ bf, int32_t bias_parts, const float *__restrict__ wide_logit,
              const float *__restrict__ labels, const float *__restrict__ weights, int64_t batch,
              float *__restrict__ dnn_logit, float *__restrict__ logit, float *__restrict__ prob,
              float *__restrict__ dlogit, float *__restrict__ loss_sum, float *__restrict__ out, int64_t ld_out,
              int32_t act, float *__restrict__ Gpart) {
  __shared__ float sdl[HEAD_CHUNK];
  __shared__ float red[256];
  const int64_t b0 = (int64_t)blockIdx.x * HEAD_CHUNK;
  float bias = 0.f;
  for (int p = 0; p < bias_parts; ++p) bias += bf[p];
  // phase 1: HEAD_LPE lanes per example (k = part, part + HEAD_LPE, ...), 256 / HEAD_LPE examples at a time
  constexpr int PAR = 256 / HEAD_LPE;
  const int part = threadIdx.x % HEAD_LPE;
  float lsum = 0.f;
  for (int ex = threadIdx.x / HEAD_LPE; ex < HEAD_CHUNK; ex += PAR) {
    const int64_t b = b0 + ex;
    const bool live = b < batch;
    float d = 0.f;
    if (live) {
      const TA *ar = a + b * ld_a;
#pragma unroll 4
      for (int64_t k = part; k < K; k += HEAD_LPE) d += (float)ar[k] * wf[k];
    }
#pragma unroll
    for (int off = 1; off < HEAD_LPE; off <<= 1) d += __shfl_xor(d, off, 64);
    float dl = 0.f;
    if (live && part == 0) {
      const float dn = d + bias;
      const float x = dn + (wide_logit ? wide_logit[b] : 0.f);
      const float y = labels ? labels[b] : 0.f;
      const float w = weights ? weights[b] : 1.0f;
      const float e = expf(-fabsf(x));
      const float p = x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      dl = w * (p - y);
      lsum += w * (fmaxf(x, 0.f) - x * y + log1pf(e));
      if (dnn_logit) dnn_logit[b] = dn;
      if (logit) logit[b] = x;
      if (prob) prob[b] = p;
      if (dlogit) dlogit[b] = dl;
    }
    if (part == 0) sdl[ex] = dl;
  }
  for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
  if ((threadIdx.x & 63) == 0 && loss_sum) atomicAdd(loss_sum, lsum);
  if (!out && !Gpart) return;
  __syncthreads();
  // phase 2: thread (kx, q): column kx (+ j*KT when K > 256), examples q, q+Q, ...
  const int KT = K <= 64 ? 64 : (K <= 128 ? 128 : 256);
  const int Q = 256 / KT;
  const int kx = threadIdx.x % KT, q = threadIdx.x / KT;
  float *Gp = Gpart ? Gpart + (int64_t)blockIdx.x * (K + 1) : nullptr;
  for (int64_t k = kx; k < K; k += KT) {
    const float w = wf[k];
    float gw = 0.f;
#pragma unroll 4
    for (int i = q; i < HEAD_CHUNK; i += Q) {
      const int64_t b = b0 + i;
      if (b >= batch) break;
      const float av = (float)a[b * ld_a + k];
      const float dl = sdl[i];
      gw += av * dl;
      if (out) out[b * ld_out + k] = act ? dl * w * act_bwd(av, act) : dl * w;
    }
    if (Q == 1) {
      if (Gp) Gp[k] = gw;
    } else {
      red[q * KT + kx] = gw;  // K <= 128 <= KT: this loop body runs at most once per thread
    }
  }
  if (Q > 1) {  // combine the Q example-strided partials in fixed order
    __syncthreads();
    if (q == 0 && kx < K && Gp) {
      float v = red[kx];
      for (int j = 1; j < Q; ++j) v += red[j * KT + kx];
      Gp[kx] = v;
    }
  }
  if (threadIdx.x == 0 && Gp) {
    float v = 0.f;
    for (int i = 0; i < HEAD_CHUNK; ++i) v += sdl[i];
    Gp[K] = v;
  }
}

// Wide logits layers (dense / resnet / last_dense windows: K in the thousands) with 16-byte loads: the scalar geometry above
// issues one 2- or 4-byte load per element (C5: 106 us for 57 MB of half activations).  Same two phases, same outputs:
//   phase 1: one wavefront per example (4 at a time), lane <- 16-byte vectors lane, lane + 64, ... of the row
//   phase 2: thread <- one 16-byte vector of columns for all 16 examples of the workgroup (rows are L2-warm from phase 1)
// Needs a, wf 16-byte aligned and ld_a a multiple of the vector length (checked by the launcher); columns beyond the last
// whole vector are done element-wise.
template <typename TA>
__global__ void __launch_bounds__(256)
k_logits_head_wide(const TA *__restrict__ a, int64_t ld_a, int64_t K, const float *__restrict__ wf,
                   const float *__restrict__ bf, int32_t bias_parts, const float *__restrict__ wide_logit,
                   const float *__restrict__ labels, const float *__restrict__ weights, int64_t batch,
                   float *__restrict__ dnn_logit, float *__restrict__ logit, float *__restrict__ prob,
                   float *__restrict__ dlogit, float *__restrict__ loss_sum, float *__restrict__ out, int64_t ld_out,
                   int32_t act, float *__restrict__ Gpart) {
  constexpr int CH = 16;
  constexpr int V = 16 / (int)sizeof(TA);
  typedef TA vec_t __attribute__((ext_vector_type(V)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ float sdl[CH];
  const int64_t b0 = (int64_t)blockIdx.x * CH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t KV = K / V;
  float bias = 0.f;
  for (int p = 0; p < bias_parts; ++p) bias += bf[p];
  float lsum = 0.f;
  for (int ex = wave; ex < CH; ex += 4) {
    const int64_t b = b0 + ex;
    const bool live = b < batch;
    float d = 0.f;
    if (live) {
      const TA *ar = a + b * ld_a;
#pragma unroll 4
      for (int64_t c = lane; c < KV; c += 64) {
        const vec_t va = *reinterpret_cast<const vec_t *>(ar + c * V);
        const f4 *w4 = reinterpret_cast<const f4 *>(wf + c * V);
#pragma unroll
        for (int j = 0; j < V / 4; ++j) {
          const f4 w = w4[j];
          d += (float)va[4 * j] * w.x;
          d += (float)va[4 * j + 1] * w.y;
          d += (float)va[4 * j + 2] * w.z;
          d += (float)va[4 * j + 3] * w.w;
        }
      }
      for (int64_t k = KV * V + lane; k < K; k += 64) d += (float)ar[k] * wf[k];
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) d += __shfl_xor(d, off, 64);
    float dl = 0.f;
    if (live && lane == 0) {
      const float dn = d + bias;
      const float x = dn + (wide_logit ? wide_logit[b] : 0.f);
      const float y = labels ? labels[b] : 0.f;
      const float w = weights ? weights[b] : 1.0f;
      const float e = expf(-fabsf(x));
      const float p = x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      dl = w * (p - y);
      lsum += w * (fmaxf(x, 0.f) - x * y + log1pf(e));
      if (dnn_logit) dnn_logit[b] = dn;
      if (logit) logit[b] = x;
      if (prob) prob[b] = p;
      if (dlogit) dlogit[b] = dl;
    }
    if (lane == 0) sdl[ex] = dl;
  }
  for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
  if (lane == 0 && loss_sum) atomicAdd(loss_sum, lsum);
  if (!out && !Gpart) return;
  __syncthreads();
  float *Gp = Gpart ? Gpart + (int64_t)blockIdx.x * (K + 1) : nullptr;
  const int nlive = (int)(batch - b0 < CH ? batch - b0 : CH);
  const bool out_vec = out && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (ld_out & 3) == 0;
  for (int64_t c = threadIdx.x; c < KV; c += 256) {
    float w[V], gw[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { w[j] = wf[c * V + j]; gw[j] = 0.f; }
#pragma unroll 4
    for (int i = 0; i < nlive; ++i) {
      const int64_t b = b0 + i;
      const vec_t va = *reinterpret_cast<const vec_t *>(a + b * ld_a + c * V);
      const float dl = sdl[i];
      float o[V];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float av = (float)va[j];
        gw[j] += av * dl;
        o[j] = act ? dl * w[j] * act_bwd(av, act) : dl * w[j];
      }
      if (out) {
        float *op = out + b * ld_out + c * V;
        if (out_vec) {
#pragma unroll
          for (int j = 0; j < V / 4; ++j) *reinterpret_cast<f4 *>(op + 4 * j) = f4{o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]};
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) op[j] = o[j];
        }
      }
    }
    if (Gp) {
#pragma unroll
      for (int j = 0; j < V; ++j) Gp[c * V + j] = gw[j];
    }
  }
  for (int64_t k = KV * V + threadIdx.x; k < K; k += 256) {
    const float w = wf[k];
    float gw = 0.f;
    for (int i = 0; i < nlive; ++i) {
      const int64_t b = b0 + i;
      const float av = (float)a[b * ld_a + k];
      const float dl = sdl[i];
      gw += av * dl;
      if (out) out[b * ld_out + k] = act ? dl * w * act_bwd(av, act) : dl * w;
    }
    if (Gp) Gp[k] = gw;
  }
  if (threadIdx.x == 0 && Gp) {
    float v = 0.f;
    for (int i = 0; i < CH; ++i) v += sdl[i];
    Gp[K] = v;
  }
}

__global__ void __launch_bounds__(256)
k_adagrad_dense(float *__restrict__ w, float *__restrict__ accum, const float *__restrict__ g, int64_t n, float lr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float gi = g[i];
    const float a = accum[i] + gi * gi;
    accum[i] = a;
    w[i] -= lr * gi / sqrtf(a);
  }
}

}  // namespace

extern "C" int wd_gemm_nn_bias_act(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                                   int32_t bias_parts, int32_t act, float *C, int64_t ldc, int64_t M, int64_t N,
                                   int64_t K, wd_stream_t stream) {
  if (M <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(A && B && C, "null pointer");
  WD_REQUIRE(K > 0, "K must be > 0");
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.kchunk = K; g.ones_row = -1; g.act = act; g.bias_parts = bias_parts > 0 ? bias_parts : 1;
  return launch_gemm<true, false, 0>(g, 1, wd::as_stream(stream), "wd_gemm_nn_bias_act");
}

extern "C" int wd_gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, int64_t M,
                          int64_t N, int64_t K, int32_t accumulate, wd_stream_t stream) {
  if (M <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(A && B && C, "null pointer");
  WD_REQUIRE(K > 0, "K must be > 0");
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.kchunk = K; g.ones_row = -1; g.accumulate = accumulate;
  return launch_gemm<true, true, 1>(g, 1, wd::as_stream(stream), "wd_gemm_nt");
}

extern "C" int wd_gemm_nt_actbwd(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                                 int64_t M, int64_t N, int64_t K, const float *act_src, int64_t ld_act, int32_t act,
                                 wd_stream_t stream) {
  if (M <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(A && B && C && act_src, "null pointer");
  WD_REQUIRE(K > 0, "K must be > 0");
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.kchunk = K; g.ones_row = -1; g.accumulate = 0;
  g.act_src = act_src; g.ld_act = ld_act; g.act = act;
  return launch_gemm<true, true, 1>(g, 1, wd::as_stream(stream), "wd_gemm_nt_actbwd");
}

extern "C" int wd_gemm_tn_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, float *Cpart, int64_t M,
                                 int64_t N, int64_t K, int32_t nsplit, int32_t append_ones, wd_stream_t stream) {
  if (M <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(A && B && Cpart, "null pointer");
  WD_REQUIRE(K > 0 && nsplit > 0, "K and nsplit must be > 0");
  GemmArgs g{};
  const int64_t Mo = append_ones ? M + 1 : M;
  g.A = A; g.B = B; g.C = Cpart; g.lda = lda; g.ldb = ldb; g.ldc = N;
  g.M = Mo; g.N = N; g.K = K;
  g.kchunk = wd::ceil_div(wd::ceil_div(K, nsplit), kBK) * kBK;
  g.c_split = Mo * N;
  g.ones_row = append_ones ? M : -1;
  return launch_gemm<false, false, 2>(g, nsplit, wd::as_stream(stream), "wd_gemm_tn_splitk");
}

extern "C" int wd_gemm_tn_splitk_group(const wd_tn_job_t *jobs, int32_t njobs, wd_stream_t stream) {
  if (njobs <= 0) return WD_OK;
  WD_REQUIRE(jobs, "null pointer");
  WD_REQUIRE(njobs <= MAX_TN_JOBS, "njobs <= WD_TN_GROUP_MAX");
  // (round 5's register-streamed variant of these products -- 35-38 us alone against 41-45, no faster in the step, where the
  // row update beside it sets the pace -- was removed in round 6: profiles/r5_products_stream.md)
  GroupArgs G{};
  bool vec = true;
  int total = 0;
  const int split_xcd = 1;   // all tiles of a split on ONE XCD (split z -> XCD z % 8); measured on one box, alternating: 0.1696 / 0.1717 / 0.1690 (tiles spread) vs 0.1686 / 0.1699 / 0.1669 ms/step
  for (int j = 0; j < njobs; ++j) {
    const wd_tn_job_t &q = jobs[j];
    GemmArgs &g = G.job[j];
    if (!q.B) {    // column sums
      WD_REQUIRE(q.A && q.Cpart && q.N > 0 && q.K > 0, "column-sum job: A, Cpart, N, K");
      g.A = q.A; g.C = q.Cpart; g.lda = q.lda; g.N = q.N; g.K = q.K;
      G.colsum[j] = 1;
      G.first[j] = total;
      G.count[j] = (int)wd::ceil_div(q.N, 64);
      total += (G.count[j] + 7) / 8 * 8;
      continue;
    }
    WD_REQUIRE(q.A && q.B && q.Cpart, "null pointer");
    WD_REQUIRE(q.M > 0 && q.N > 0 && q.K > 0 && q.nsplit > 0, "M, N, K, nsplit must be > 0");
    const int64_t Mo = q.append_ones ? q.M + 1 : q.M;
    g.A = q.A; g.B = q.B; g.C = q.Cpart; g.lda = q.lda; g.ldb = q.ldb; g.ldc = q.N;
    g.M = Mo; g.N = q.N; g.K = q.K;
    g.kchunk = wd::ceil_div(wd::ceil_div(q.K, q.nsplit), kBK) * kBK;
    g.c_split = Mo * q.N;
    g.ones_row = q.append_ones ? q.M : -1;
    g.tiles_m = (int)wd::ceil_div(g.M, BM);
    g.tiles_n = (int)wd::ceil_div(g.N, BN);
    g.a_vec = (g.lda % 4 == 0) && aligned16(g.A);
    g.b_vec = (g.ldb % 4 == 0) && aligned16(g.B);
    g.wt = wd::wt_mask() & WD_WT_PRODUCTS ? 1 : 0;
    vec = vec && g.a_vec && g.b_vec;
    G.first[j] = total;
    G.nsplit[j] = q.nsplit;
    G.count[j] = g.tiles_m * g.tiles_n * (split_xcd ? wd::kXCDs * (int)wd::ceil_div(q.nsplit, wd::kXCDs) : q.nsplit);
    total += (G.count[j] + 7) / 8 * 8;
  }
  G.first[njobs] = total;
  G.njobs = njobs;
  // 32-deep slabs: 35 KB of LDS per workgroup instead of 70 -- the sparse update that runs beside this launch keeps
  // workgroups resident on the same CUs (WD_TN_BK=64 for the deeper slabs)
  static const bool bk64 = getenv("WD_TN_BK") && atoi(getenv("WD_TN_BK")) == 64;   // (16-deep slabs measured no better)
  static const size_t pad = getenv("WD_TN_LDS") ? (size_t)atoi(getenv("WD_TN_LDS")) : 0;   // diagnostics: extra LDS bytes per workgroup (34816 = the 70 KB of rounds 2-4)
  if (bk64) {
    if (vec) hipLaunchKernelGGL((k_gemm_tn_group<kBK, true>), dim3((unsigned)total), dim3(256), 0, wd::as_stream(stream), G);
    else hipLaunchKernelGGL((k_gemm_tn_group<kBK, false>), dim3((unsigned)total), dim3(256), 0, wd::as_stream(stream), G);
  } else {
    if (vec) hipLaunchKernelGGL((k_gemm_tn_group<32, true>), dim3((unsigned)total), dim3(256), pad, wd::as_stream(stream), G);
    else hipLaunchKernelGGL((k_gemm_tn_group<32, false>), dim3((unsigned)total), dim3(256), pad, wd::as_stream(stream), G);
  }
  return wd::check_launch("wd_gemm_tn_splitk_group");
}


extern "C" int wd_fold_affine(const float *P, int64_t w_off, int64_t b_off, const int32_t *gamma_idx,
                              const int32_t *beta_idx, float inv, float *Wf, float *bf, float *s, float *t, int64_t K,
                              int64_t N, wd_stream_t stream) {
  WD_REQUIRE(P && Wf && bf && s && t, "null pointer");
  WD_REQUIRE(K > 0 && N > 0, "K, N must be > 0");
  hipLaunchKernelGGL(k_fold_affine, dim3((unsigned)wd::ceil_div(N, 64), WD_FOLD_PARTS), dim3(256), 0,
                     wd::as_stream(stream), P, w_off, b_off, gamma_idx, beta_idx, inv, Wf, bf, s, t, K, N);
  return wd::check_launch("wd_fold_affine");
}

extern "C" int wd_act_bwd(const float *da, int64_t ldda, const float *a, int64_t lda, int32_t act, float *dz,
                          int64_t lddz, int64_t M, int64_t N, wd_stream_t stream) {
  if (M <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(da && a && dz, "null pointer");
  hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)wd::ceil_div(M * N, 256)), dim3(256), 0, wd::as_stream(stream), da,
                     ldda, a, lda, act, dz, lddz, M, N);
  return wd::check_launch("wd_act_bwd");
}

extern "C" int wd_dropout_fwd(float *a, int64_t lda, int64_t M, int64_t N, float rate, const int64_t *seed_step,
                              int32_t layer, wd_stream_t stream) {
  if (M <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(a && seed_step, "null pointer");
  WD_REQUIRE(rate > 0.f && rate < 1.f, "0 < rate < 1");
  hipLaunchKernelGGL(k_dropout_fwd, dim3((unsigned)wd::ceil_div(M * N, 256)), dim3(256), 0, wd::as_stream(stream), a, lda,
                     M, N, rate, seed_step, layer);
  return wd::check_launch("wd_dropout_fwd");
}

extern "C" int wd_act_bwd_dropout(const float *da, int64_t ldda, const float *a_drop, int64_t lda, int32_t act, float *dz,
                                  int64_t lddz, int64_t M, int64_t N, float rate, const int64_t *seed_step, int32_t layer,
                                  wd_stream_t stream) {
  if (M <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(da && a_drop && dz && seed_step, "null pointer");
  WD_REQUIRE(rate > 0.f && rate < 1.f, "0 < rate < 1");
  hipLaunchKernelGGL(k_act_bwd_dropout, dim3((unsigned)wd::ceil_div(M * N, 256)), dim3(256), 0, wd::as_stream(stream), da,
                     ldda, a_drop, lda, act, dz, lddz, M, N, rate, seed_step, layer);
  return wd::check_launch("wd_act_bwd_dropout");
}

extern "C" int wd_counter_tick(int64_t *seed_step, wd_stream_t stream) {
  WD_REQUIRE(seed_step, "null pointer");
  hipLaunchKernelGGL(k_counter_tick, dim3(1), dim3(1), 0, wd::as_stream(stream), seed_step);
  return wd::check_launch("wd_counter_tick");
}

extern "C" int wd_mlp_finalize(const float *Gpart, int32_t nsplit, const float *P, int64_t w_off, int64_t b_off,
                               const float *s, const float *t, const int32_t *gamma_idx, const int32_t *beta_idx,
                               float inv, float *Gflat, int64_t K, int64_t N, wd_stream_t stream) {
  WD_REQUIRE(Gpart && P && s && t && Gflat, "null pointer");
  WD_REQUIRE(K > 0 && N > 0 && nsplit > 0, "K, N, nsplit must be > 0");
  const bool al16 = ((reinterpret_cast<uintptr_t>(Gpart) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(Gflat)) & 15) == 0 &&
                    (w_off & 3) == 0 && (b_off & 3) == 0;
  if ((N & 3) == 0 && N >= 64 && al16 && getenv("WD_FINALIZE_ROWS") == nullptr)
    hipLaunchKernelGGL(k_mlp_finalize4, dim3((unsigned)(K + 1)), dim3(256), 0, wd::as_stream(stream), Gpart, nsplit, P,
                       w_off, b_off, s, t, gamma_idx, beta_idx, inv, Gflat, K, N);
  else if (N == 1 && nsplit >= 32 && getenv("WD_FINALIZE_ROWS") == nullptr)
    hipLaunchKernelGGL(k_mlp_finalize_vec, dim3((unsigned)wd::ceil_div(K + 1, 16)), dim3(256), 0, wd::as_stream(stream),
                       Gpart, nsplit, P, w_off, b_off, s, t, gamma_idx, beta_idx, inv, Gflat, K);
  else
    hipLaunchKernelGGL(k_mlp_finalize, dim3((unsigned)(K + 1)), dim3(256), 0, wd::as_stream(stream), Gpart, nsplit, P,
                       w_off, b_off, s, t, gamma_idx, beta_idx, inv, Gflat, K, N);
  return wd::check_launch("wd_mlp_finalize");
}

extern "C" int wd_adagrad_dense(float *w, float *accum, const float *g, int64_t n, float lr, wd_stream_t stream) {
  if (n <= 0) return WD_OK;
  WD_REQUIRE(w && accum && g, "null pointer");
  int blocks = (int)std::min<int64_t>(wd::ceil_div(n, 256), 2048);
  hipLaunchKernelGGL(k_adagrad_dense, dim3(blocks), dim3(256), 0, wd::as_stream(stream), w, accum, g, n, lr);
  return wd::check_launch("wd_adagrad_dense");
}

namespace {
// crelu layers (python/lib/utils/model_util.py:52, tf.nn.crelu) run as relu layers of width 2N whose kernel is [W | -W]:
// dL/dW[k][n] = G'[k][n] - G'[k][N + n]; the left half keeps it, the right half its negation (so that any sign-symmetric
// optimizer keeps the halves exact mirrors).  Row K of the grid = the bias [b | -b].
__global__ void __launch_bounds__(256)
k_crelu_tie(float *__restrict__ G, int64_t w_off, int64_t b_off, int64_t K, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (K + 1) * N) return;
  const int64_t k = i / N, n = i - k * N;
  float *row = k < K ? G + w_off + k * 2 * N : G + b_off;
  const float g = row[n] - row[N + n];
  row[n] = g;
  row[N + n] = -g;
}
}  // namespace

extern "C" int wd_crelu_tie(float *Gflat, int64_t w_off, int64_t b_off, int64_t K, int64_t N, wd_stream_t stream) {
  WD_REQUIRE(Gflat && K > 0 && N > 0, "null pointer / empty layer");
  hipLaunchKernelGGL(k_crelu_tie, dim3((unsigned)wd::ceil_div((K + 1) * N, 256)), dim3(256), 0, wd::as_stream(stream),
                     Gflat, w_off, b_off, K, N);
  return wd::check_launch("wd_crelu_tie");
}

extern "C" int wd_fold_affine_all(const float *P, const wd_mlp_layer_t *layers_dev, int32_t nlayers, int64_t max_n,
                                  float inv, float *zero_a, int64_t zero_a_n, float *zero_b, int64_t zero_b_n,
                                  wd_stream_t stream) {
  WD_REQUIRE(P && layers_dev, "null pointer");
  WD_REQUIRE(nlayers > 0 && max_n > 0, "nlayers, max_n must be > 0");
  hipLaunchKernelGGL(k_fold_affine_all, dim3((unsigned)wd::ceil_div(max_n, 64), WD_FOLD_PARTS, (unsigned)nlayers),
                     dim3(256), 0, wd::as_stream(stream), P, layers_dev, inv, zero_a, zero_a ? zero_a_n : 0, zero_b,
                     zero_b ? zero_b_n : 0);
  return wd::check_launch("wd_fold_affine_all");
}

extern "C" int wd_mlp_finalize_all(const wd_mlp_layer_t *layers_dev, int32_t nlayers, int64_t max_k, const float *P,
                                   float inv, float *Gflat, wd_stream_t stream) {
  WD_REQUIRE(P && layers_dev && Gflat, "null pointer");
  WD_REQUIRE(nlayers > 0 && max_k > 0, "nlayers, max_k must be > 0");
  hipLaunchKernelGGL(k_mlp_finalize_all, dim3((unsigned)(max_k + 1), (unsigned)nlayers), dim3(256), 0,
                     wd::as_stream(stream), layers_dev, P, inv, Gflat, (float *)nullptr, (float *)nullptr, 0.f);
  return wd::check_launch("wd_mlp_finalize_all");
}

extern "C" int wd_mlp_finalize_adagrad_all(const wd_mlp_layer_t *layers_dev, int32_t nlayers, int64_t max_k, float *P,
                                           float *Pacc, float inv, float *Gflat, float lr, wd_stream_t stream) {
  WD_REQUIRE(P && Pacc && layers_dev && Gflat, "null pointer");
  WD_REQUIRE(nlayers > 0 && max_k > 0, "nlayers, max_k must be > 0");
  hipLaunchKernelGGL(k_mlp_finalize_all, dim3((unsigned)(max_k + 1), (unsigned)nlayers), dim3(256), 0,
                     wd::as_stream(stream), layers_dev, P, inv, Gflat, P, Pacc, lr);
  return wd::check_launch("wd_mlp_finalize_adagrad_all");
}

extern "C" int64_t wd_logits_head_blocks(int64_t batch, int64_t K) { return wd::ceil_div(batch, head_chunk(K)); }

template <typename TA>
static void launch_head(hipStream_t st, const TA *a, int64_t ld_a, int64_t K, const float *wf, const float *bf,
                        int32_t bias_parts, const float *wide_logit, const float *labels, const float *weights,
                        int64_t batch, float *dnn_logit, float *logit, float *prob, float *dlogit, float *loss_sum,
                        float *out, int64_t ld_out, int32_t act, float *Gpart) {
  const dim3 grid((unsigned)wd::ceil_div(batch, head_chunk(K)));
  constexpr int64_t V = 16 / (int64_t)sizeof(TA);
  const bool vec_ok = (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(wf) & 15) == 0 &&
                      ld_a % V == 0 && getenv("WD_HEAD_SCALAR") == nullptr;
  if (K > HEAD_K_WIDE && vec_ok)
    hipLaunchKernelGGL((k_logits_head_wide<TA>), grid, dim3(256), 0, st, a, ld_a, K, wf, bf, bias_parts, wide_logit,
                       labels, weights, batch, dnn_logit, logit, prob, dlogit, loss_sum, out, ld_out, act, Gpart);
  else if (K > HEAD_K_WIDE)
    hipLaunchKernelGGL((k_logits_head<TA, 16, 64>), grid, dim3(256), 0, st, a, ld_a, K, wf, bf, bias_parts, wide_logit,
                       labels, weights, batch, dnn_logit, logit, prob, dlogit, loss_sum, out, ld_out, act, Gpart);
  else
    hipLaunchKernelGGL((k_logits_head<TA, 64, 4>), grid, dim3(256), 0, st, a, ld_a, K, wf, bf, bias_parts, wide_logit,
                       labels, weights, batch, dnn_logit, logit, prob, dlogit, loss_sum, out, ld_out, act, Gpart);
}

extern "C" int wd_logits_head(const float *a, int64_t ld_a, int64_t K, const float *wf, const float *bf,
                              int32_t bias_parts, const float *wide_logit, const float *labels, const float *weights,
                              int64_t batch, float *dnn_logit, float *logit, float *prob, float *dlogit,
                              float *loss_sum, float *out, int64_t ld_out, int32_t act, float *Gpart,
                              wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(a && wf && bf, "null pointer");
  WD_REQUIRE(K > 0 && bias_parts > 0, "K, bias_parts must be > 0");
  WD_REQUIRE(labels || (!dlogit && !out && !Gpart), "labels required for the backward outputs");
  launch_head<float>(wd::as_stream(stream), a, ld_a, K, wf, bf, bias_parts, wide_logit, labels, weights, batch,
                     dnn_logit, logit, prob, dlogit, loss_sum, out, ld_out, act, Gpart);
  return wd::check_launch("wd_logits_head");
}

extern "C" int wd_logits_head_h(const uint16_t *a_h, int64_t ld_a, int64_t K, const float *wf, const float *bf,
                                int32_t bias_parts, const float *wide_logit, const float *labels, const float *weights,
                                int64_t batch, float *dnn_logit, float *logit, float *prob, float *dlogit,
                                float *loss_sum, float *out, int64_t ld_out, int32_t act, float *Gpart,
                                wd_stream_t stream) {
  if (batch <= 0) return WD_OK;
  WD_REQUIRE(a_h && wf && bf, "null pointer");
  WD_REQUIRE(K > 0 && bias_parts > 0, "K, bias_parts must be > 0");
  WD_REQUIRE(labels || (!dlogit && !out && !Gpart), "labels required for the backward outputs");
  launch_head<_Float16>(wd::as_stream(stream), reinterpret_cast<const _Float16 *>(a_h), ld_a, K, wf, bf, bias_parts,
                        wide_logit, labels, weights, batch, dnn_logit, logit, prob, dlogit, loss_sum, out, ld_out, act,
                        Gpart);
  return wd::check_launch("wd_logits_head_h");
}
