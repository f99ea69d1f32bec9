// wide_deep_amd/csrc/mlp_half.hip -- fp16-input MFMA tower (BASELINE configs[4]: "fp16 MFMA dense path with fp32
// embedding"): the same layers as mlp.hip (python/lib/dnn.py:92-234) with the GEMM operands rounded to IEEE half and
// accumulated in fp32 on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate on gfx950).  Embeddings, the pooled input x,
// the gradient accumulators, the split-K partials, every optimizer state and the logits/head stay fp32.
//
// Layout decision that makes all three products of a layer "reduction-contiguous x reduction-contiguous":
//   activations live twice in half:  act_h [B][ld]  and its transpose  actT_h [ld][Bp]
//   dz likewise:                     dz_h  [B][N]   and                dzT_h  [N][Bp]
//   weights (folded with the BN affine each step, see wd_fold_affine_all):  Wf_h [K][Np]  and  WfT_h [N][Kp]
//   NN  a_l      = act(A Wf)       A = act_h  rows b, k contiguous    B = WfT_h rows n, k contiguous
//   NT  da / dz  = dZ Wf^T         A = dz_h   rows b, n contiguous    B = Wf_h  rows k, n contiguous
//   TN  G        = [A|1]^T dZ      A = actT_h rows k, b contiguous    B = dzT_h rows n, b contiguous
// so every tile is fetched with 16-byte loads, lands in LDS with conflict-free ds_write_b128, and every MFMA fragment
// is ONE ds_read_b128 (8 halfs of one row).  The GEMM epilogues write the half copy AND the transposed half copy of
// their output (4 consecutive rows of one column = one 8-byte store), so no separate transpose pass exists except for
// the gathered input x and for fp32 gradient buffers (k_cast_transpose).
//
// Tile: 2x2 waves, each wave WT x WT (WT = 64: four 32x32 accumulators; WT = 32 for narrow outputs), BK = 64 halfs,
// LDS double-buffered, global loads of slab i+1 in flight across the MFMAs of slab i.
#include "common.h"

namespace {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int HBK = 64;          // reduction slab (halfs)
constexpr int HLD = HBK + 8;     // LDS row pitch in halfs: 144 B -> 16 rows x 16 B cover all 64 banks once

enum { H_NN = 0, H_NT_ACC = 1, H_NT_DZ = 2, H_TN = 3 };

struct HArgs {
  const half_t *A, *B;          // A [M][lda], B [N][ldb], reduction index contiguous
  int64_t lda, ldb, M, N, K;
  int64_t kchunk;               // reduction slice per z
  int64_t ones_row;             // row of A synthesised as all ones (-1: none)
  half_t *Ch, *CT;              // half outputs [M][ldch] and transposed [N][ldct]   (H_NN, H_NT_DZ)
  int64_t ldch, ldct;
  float *C32;                   // fp32 output [M][ldc32] (H_NT_ACC) / split-K partials [z][M][N] (H_TN)
  int64_t ldc32, c_split;
  const float *bias;            // H_NN: sum of bias_parts vectors of stride N
  const half_t *act_src;        // H_NT_DZ: C *= act'(act_src[m][n])
  int64_t ld_act;
  int32_t bias_parts, act, accumulate, vec;   // vec: 16-byte loads legal for both operands
  int32_t tiles_m, tiles_n;
};

__device__ __forceinline__ float hact_fwd(float v, int act) {
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
__device__ __forceinline__ float hact_bwd(float a, int act) {
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

// 8 halfs of row `row` starting at reduction index k (k % 8 == 0); out-of-range elements are zero
__device__ __forceinline__ half8 load8(const half_t *__restrict__ P, int64_t row, int64_t k, int64_t ld, int64_t R,
                                       int64_t kend, bool vec, bool ones) {
  half8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (half_t)0.f;
  if (ones) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (k + i < kend) ? (half_t)1.f : (half_t)0.f;
    return v;
  }
  if (row >= R || k >= kend) return v;
  const half_t *p = P + row * ld + k;
  if (vec && k + 7 < kend) return *reinterpret_cast<const half8 *>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (k + i < kend) v[i] = p[i];
  return v;
}

template <int WT, int MODE>
__global__ void __launch_bounds__(256) k_hgemm(HArgs g) {
  constexpr int BT = 2 * WT;             // block tile (rows of A and of B)
  constexpr int NV = BT * 8 / 256;       // 16-byte loads per lane per operand per slab
  constexpr int NT = WT / 32;            // 32x32 accumulator tiles per wave and dimension
  __shared__ __attribute__((aligned(16))) half_t As[2][BT * HLD];
  __shared__ __attribute__((aligned(16))) half_t Bs[2][BT * HLD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane >> 5, fc = lane & 31;

  // XCD-aware tile order (n fastest inside an XCD's contiguous run), as in mlp.hip
  const int tiles = g.tiles_m * g.tiles_n;
  const int bz = blockIdx.x / tiles, orig = blockIdx.x % tiles;
  const int q = tiles / wd::kXCDs, r = tiles % wd::kXCDs;
  const int xcd = orig % wd::kXCDs, loc = orig / wd::kXCDs;
  const int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int64_t m0 = (int64_t)(vid / g.tiles_n) * BT;
  const int64_t n0 = (int64_t)(vid % g.tiles_n) * BT;
  const int64_t kbeg = (int64_t)bz * g.kchunk;
  const int64_t kend = kbeg + g.kchunk < g.K ? kbeg + g.kchunk : g.K;
  const int64_t a_rows = g.ones_row >= 0 ? g.ones_row : g.M;

  const bool fast_tile = g.vec && (m0 + BT <= a_rows) && (n0 + BT <= g.N);
  half8 ra[NV], rb[NV];
  auto load_slab = [&](int64_t k0) {
    const bool fast = fast_tile && (k0 + HBK <= kend);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = t + v * 256;
      const int row = idx >> 3, kc = (idx & 7) * 8;
      if (fast) {
        ra[v] = *reinterpret_cast<const half8 *>(g.A + (m0 + row) * g.lda + k0 + kc);
        rb[v] = *reinterpret_cast<const half8 *>(g.B + (n0 + row) * g.ldb + k0 + kc);
      } else {
        ra[v] = load8(g.A, m0 + row, k0 + kc, g.lda, a_rows, kend, g.vec, g.ones_row >= 0 && m0 + row == g.ones_row);
        rb[v] = load8(g.B, n0 + row, k0 + kc, g.ldb, g.N, kend, g.vec, false);
      }
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = t + v * 256;
      const int row = idx >> 3, kc = (idx & 7) * 8;
      *reinterpret_cast<half8 *>(&As[buf][row * HLD + kc]) = ra[v];
      *reinterpret_cast<half8 *>(&Bs[buf][row * HLD + kc]) = rb[v];
    }
  };

  floatx16 acc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_slab(kbeg);
  store_slab(0);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += HBK) {
    const bool more = k0 + HBK < kend;
    if (more) load_slab(k0 + HBK);
    const half_t *Ac = &As[cur][(wm * WT + fc) * HLD + fr * 8];
    const half_t *Bc = &Bs[cur][(wn * WT + fc) * HLD + fr * 8];
#pragma unroll
    for (int ks = 0; ks < HBK / 16; ++ks) {
      half8 fa[NT], fb[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) fa[i] = *reinterpret_cast<const half8 *>(Ac + i * 32 * HLD + ks * 16);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const half8 *>(Bc + j * 32 * HLD + ks * 16);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_slab(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // C/D layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  auto epilogue = [&](const floatx16 &tile, int i, int j) {
    const int64_t n = n0 + wn * WT + j * 32 + fc;
    const bool n_ok = n < g.N;
    float bv = 0.f;
    if (MODE == H_NN && g.bias && n_ok)
      for (int p = 0; p < g.bias_parts; ++p) bv += g.bias[(int64_t)p * g.N + n];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int64_t mb = m0 + wm * WT + i * 32 + 8 * g4 + 4 * fr;   // 4 consecutive rows mb..mb+3
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t m = mb + e;
        float x = tile[g4 * 4 + e];
        if (MODE == H_NN) {
          x = hact_fwd(x + bv, g.act);
        } else if (MODE == H_NT_DZ) {
          if (m < g.M && n_ok) x *= hact_bwd((float)g.act_src[m * g.ld_act + n], g.act);
        } else if (MODE == H_NT_ACC) {
          if (m < g.M && n_ok) {
            float *c = g.C32 + m * g.ldc32 + n;
            *c = g.accumulate ? *c + x : x;
          }
        } else {  // H_TN
          if (m < g.M && n_ok) g.C32[(int64_t)bz * g.c_split + m * g.N + n] = x;
        }
        v[e] = x;
      }
      if ((MODE == H_NN || MODE == H_NT_DZ) && n_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (mb + e < g.M) g.Ch[(mb + e) * g.ldch + n] = (half_t)v[e];
        if (g.CT) {
          half_t *ct = g.CT + n * g.ldct + mb;
          if (mb + 3 < g.M) {
            half4v pk;
            pk[0] = (half_t)v[0]; pk[1] = (half_t)v[1]; pk[2] = (half_t)v[2]; pk[3] = (half_t)v[3];
            *reinterpret_cast<half4v *>(ct) = pk;   // mb % 4 == 0 and ldct % 4 == 0: 8-byte aligned
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (mb + e < g.M) ct[e] = (half_t)v[e];
          }
        }
      }
    }
  };
  epilogue(acc[0][0], 0, 0);
  if constexpr (NT == 2) {
    epilogue(acc[0][1], 0, 1);
    epilogue(acc[1][0], 1, 0);
    epilogue(acc[1][1], 1, 1);
  }
}

// src fp32 [R][C] (ld lds) * optional act'(act_h) -> dst_h [R][ldh] and dstT_h [C][ldt]   (64x64 tiles through LDS)
__global__ void __launch_bounds__(256)
k_cast_transpose(const float *__restrict__ src, int64_t lds_, int64_t R, int64_t C, const half_t *__restrict__ act_h,
                 int64_t ld_act, int32_t act, half_t *__restrict__ dst, int64_t ldh, half_t *__restrict__ dstT,
                 int64_t ldt) {
  __shared__ half_t tile[64][66];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const int64_t r = r0 + rr, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) {
      v = src[r * lds_ + c];
      if (act_h) v *= hact_bwd((float)act_h[r * ld_act + c], act);
      if (dst) dst[r * ldh + c] = (half_t)v;
    }
    tile[rr][tx] = (half_t)v;
  }
  __syncthreads();
  if (!dstT) return;
  for (int cc = ty; cc < 64; cc += 4) {
    const int64_t c = c0 + cc, r = r0 + tx;
    if (c < C && r < R) dstT[c * ldt + r] = tile[tx][cc];
  }
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int MODE>
int launch_h(HArgs g, int nsplit, hipStream_t st, const char *what) {
  g.vec = (g.lda % 8 == 0) && (g.ldb % 8 == 0) && al16(g.A) && al16(g.B);
  // narrow outputs: 64x64 block tiles keep more workgroups in flight
  const bool small = (g.M <= 64 || g.N <= 128 || wd::ceil_div(g.M, 128) * wd::ceil_div(g.N, 128) * nsplit < 128);
  const int BT = small ? 64 : 128;
  g.tiles_m = (int)wd::ceil_div(g.M, BT);
  g.tiles_n = (int)wd::ceil_div(g.N, BT);
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n * nsplit));
  if (small) hipLaunchKernelGGL((k_hgemm<32, MODE>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((k_hgemm<64, MODE>), grid, dim3(256), 0, st, g);
  return wd::check_launch(what);
}

}  // namespace

extern "C" int wd_hgemm_nn(const wd_half_t *A, int64_t lda, const wd_half_t *WT, int64_t ldw, const float *bias,
                           int32_t bias_parts, int32_t act, wd_half_t *C, int64_t ldc, wd_half_t *CT, int64_t ldct,
                           int64_t M, int64_t N, int64_t K, wd_stream_t stream) {
  if (M <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(A && WT && C, "null pointer");
  WD_REQUIRE(K > 0 && (!CT || ldct % 4 == 0), "K must be > 0 and ldct a multiple of 4");
  HArgs g{};
  g.A = (const half_t *)A; g.B = (const half_t *)WT; g.lda = lda; g.ldb = ldw; g.M = M; g.N = N; g.K = K; g.kchunk = K;
  g.ones_row = -1; g.Ch = (half_t *)C; g.ldch = ldc; g.CT = (half_t *)CT; g.ldct = ldct; g.bias = bias;
  g.bias_parts = bias_parts > 0 ? bias_parts : 1; g.act = act;
  return launch_h<H_NN>(g, 1, wd::as_stream(stream), "wd_hgemm_nn");
}

extern "C" int wd_hgemm_nt(const wd_half_t *dZ, int64_t lddz, const wd_half_t *W, int64_t ldw, int64_t M, int64_t N,
                           int64_t K, float *C32, int64_t ldc32, int32_t accumulate, wd_half_t *Ch, int64_t ldch,
                           wd_half_t *CT, int64_t ldct, const wd_half_t *act_src, int64_t ld_act, int32_t act,
                           wd_stream_t stream) {
  if (M <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(dZ && W && (C32 || Ch), "null pointer");
  WD_REQUIRE(K > 0 && (!CT || ldct % 4 == 0), "K must be > 0 and ldct a multiple of 4");
  HArgs g{};
  g.A = (const half_t *)dZ; g.B = (const half_t *)W; g.lda = lddz; g.ldb = ldw; g.M = M; g.N = N; g.K = K; g.kchunk = K;
  g.ones_row = -1;
  if (C32) {
    g.C32 = C32; g.ldc32 = ldc32; g.accumulate = accumulate;
    return launch_h<H_NT_ACC>(g, 1, wd::as_stream(stream), "wd_hgemm_nt");
  }
  WD_REQUIRE(act_src, "act_src required for the fused activation-derivative epilogue");
  g.Ch = (half_t *)Ch; g.ldch = ldch; g.CT = (half_t *)CT; g.ldct = ldct; g.act_src = (const half_t *)act_src;
  g.ld_act = ld_act; g.act = act;
  return launch_h<H_NT_DZ>(g, 1, wd::as_stream(stream), "wd_hgemm_nt");
}

extern "C" int wd_hgemm_tn_splitk(const wd_half_t *AT, int64_t ldat, const wd_half_t *dZT, int64_t lddzt, float *Gpart,
                                  int64_t K, int64_t N, int64_t batch, int32_t nsplit, wd_stream_t stream) {
  if (K <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(AT && dZT && Gpart, "null pointer");
  WD_REQUIRE(batch > 0 && nsplit > 0, "batch and nsplit must be > 0");
  HArgs g{};
  g.A = (const half_t *)AT; g.B = (const half_t *)dZT; g.lda = ldat; g.ldb = lddzt; g.M = K + 1; g.N = N; g.K = batch;
  g.kchunk = wd::ceil_div(wd::ceil_div(batch, nsplit), HBK) * HBK;
  g.ones_row = K; g.C32 = Gpart; g.c_split = (K + 1) * N;
  return launch_h<H_TN>(g, nsplit, wd::as_stream(stream), "wd_hgemm_tn_splitk");
}

extern "C" int wd_cast_transpose_h(const float *src, int64_t ld_src, int64_t rows, int64_t cols, const wd_half_t *act_h,
                                   int64_t ld_act, int32_t act, wd_half_t *dst, int64_t ld_dst, wd_half_t *dstT,
                                   int64_t ld_dstT, wd_stream_t stream) {
  if (rows <= 0 || cols <= 0) return WD_OK;
  WD_REQUIRE(src && (dst || dstT), "null pointer");
  dim3 grid((unsigned)wd::ceil_div(cols, 64), (unsigned)wd::ceil_div(rows, 64));
  hipLaunchKernelGGL(k_cast_transpose, grid, dim3(256), 0, wd::as_stream(stream), src, ld_src, rows, cols,
                     (const half_t *)act_h, ld_act, act, (half_t *)dst, ld_dst, (half_t *)dstT, ld_dstT);
  return wd::check_launch("wd_cast_transpose_h");
}
