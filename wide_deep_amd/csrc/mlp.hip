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
// Tile 64x64x16, 4 waves (2x2) each owning one 32x32 accumulator; LDS tiles are reduction-major so
// the MFMA fragment reads are conflict-free ds_read_b32 rows; global->register prefetch of the next
// tile overlaps the 8 MFMAs of the current one; workgroup ids are remapped so the N-tiles that
// re-read one A row-panel run on the same XCD (shared L2).
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 16, LD = 68;

struct GemmArgs {
  const float *A;
  const float *B;
  float *C;
  const float *bias;
  int64_t lda, ldb, ldc;
  int64_t M, N, K;      // output M x N, reduction K
  int64_t kchunk;       // reduction slice per blockIdx.z
  int64_t c_split;      // element stride between split-K partial outputs
  int64_t ones_row;     // TN: output row index of A^T that is synthesised as all-ones (-1: none)
  int32_t act;
  int32_t accumulate;
  int32_t bias_parts;   // bias = sum of `bias_parts` vectors of stride N
  int32_t a_vec, b_vec; // float4 loads legal (ld % 4 == 0 and 16-byte aligned base)
  int32_t tiles_m, tiles_n;
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

// rows x cols window [R, C] of a row-major matrix; 4 consecutive columns starting at col (col % 4 == 0)
__device__ __forceinline__ float4 load4(const float *__restrict__ P, int64_t row, int64_t col, int64_t ld, int64_t R,
                                        int64_t C, bool vec_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row >= R || col >= C) return v;
  const float *p = P + row * ld + col;
  if (vec_ok && col + 3 < C) return *reinterpret_cast<const float4 *>(p);
  v.x = p[0];
  if (col + 1 < C) v.y = p[1];
  if (col + 2 < C) v.z = p[2];
  if (col + 3 < C) v.w = p[3];
  return v;
}

template <bool A_RC, bool B_RC, int EPI>
__global__ void __launch_bounds__(256) k_gemm(GemmArgs g) {
  __shared__ float As[BK * LD];
  __shared__ float Bs[BK * LD];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile mapping: hardware places block i on XCD i % 8; give each XCD a contiguous run of
  // tiles (n fastest) so the N-tiles sharing an A panel hit the same L2.  Bijective for any grid.
  const int nwg = g.tiles_m * g.tiles_n;
  const int orig = blockIdx.x;
  const int q = nwg / wd::kXCDs, r = nwg % wd::kXCDs;
  const int xcd = orig % wd::kXCDs, loc = orig / wd::kXCDs;
  const int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int64_t m0 = (int64_t)(vid / g.tiles_n) * BM;
  const int64_t n0 = (int64_t)(vid % g.tiles_n) * BN;

  const int64_t kbeg = (int64_t)blockIdx.z * g.kchunk;
  const int64_t kend = kbeg + g.kchunk < g.K ? kbeg + g.kchunk : g.K;

  // loader coordinates
  const int rc_o = t >> 2, rc_r = (t & 3) * 4;  // RC: 64 out rows x 16 reduction (4 float4 per row)
  const int oc_r = t >> 4, oc_o = (t & 15) * 4; // OC: 16 reduction rows x 64 out (16 float4 per row)

  auto load_a = [&](int64_t k0) -> float4 {
    if (A_RC) return load4(g.A, m0 + rc_o, k0 + rc_r, g.lda, g.M, kend, g.a_vec);
    float4 v = load4(g.A, k0 + oc_r, m0 + oc_o, g.lda, kend, g.ones_row >= 0 ? g.ones_row : g.M, g.a_vec);
    if (g.ones_row >= 0 && k0 + oc_r < kend) {
      const int64_t c = m0 + oc_o;
      if (c == g.ones_row) v.x = 1.f;
      if (c + 1 == g.ones_row) v.y = 1.f;
      if (c + 2 == g.ones_row) v.z = 1.f;
      if (c + 3 == g.ones_row) v.w = 1.f;
    }
    return v;
  };
  auto load_b = [&](int64_t k0) -> float4 {
    if (B_RC) return load4(g.B, n0 + rc_o, k0 + rc_r, g.ldb, g.N, kend, g.b_vec);
    return load4(g.B, k0 + oc_r, n0 + oc_o, g.ldb, kend, g.N, g.b_vec);
  };
  auto store_a = [&](float4 v) {
    if (A_RC) {
      As[(rc_r + 0) * LD + rc_o] = v.x; As[(rc_r + 1) * LD + rc_o] = v.y;
      As[(rc_r + 2) * LD + rc_o] = v.z; As[(rc_r + 3) * LD + rc_o] = v.w;
    } else {
      *reinterpret_cast<float4 *>(&As[oc_r * LD + oc_o]) = v;
    }
  };
  auto store_b = [&](float4 v) {
    if (B_RC) {
      Bs[(rc_r + 0) * LD + rc_o] = v.x; Bs[(rc_r + 1) * LD + rc_o] = v.y;
      Bs[(rc_r + 2) * LD + rc_o] = v.z; Bs[(rc_r + 3) * LD + rc_o] = v.w;
    } else {
      *reinterpret_cast<float4 *>(&Bs[oc_r * LD + oc_o]) = v;
    }
  };

  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  float4 ra = load_a(kbeg), rb = load_b(kbeg);
  store_a(ra);
  store_b(rb);
  __syncthreads();

  const int fr = (lane >> 5), fc = (lane & 31);
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      ra = load_a(k0 + BK);
      rb = load_b(k0 + BK);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a = As[(kk * 2 + fr) * LD + wm * 32 + fc];
      const float b = Bs[(kk * 2 + fr) * LD + wn * 32 + fc];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      store_a(ra);
      store_b(rb);
      __syncthreads();
    }
  }

  // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int64_t n = n0 + wn * 32 + fc;
  if (n >= g.N) return;
  float *Cz = g.C + (EPI == 2 ? (int64_t)blockIdx.z * g.c_split : 0);
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
    }
    Cz[m * g.ldc + n] = v;
  }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool A_RC, bool B_RC, int EPI>
int launch_gemm(GemmArgs g, int nsplit, hipStream_t st, const char *what) {
  g.tiles_m = (int)wd::ceil_div(g.M, BM);
  g.tiles_n = (int)wd::ceil_div(g.N, BN);
  g.a_vec = (g.lda % 4 == 0) && aligned16(g.A);
  g.b_vec = (g.ldb % 4 == 0) && aligned16(g.B);
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)nsplit);
  hipLaunchKernelGGL((k_gemm<A_RC, B_RC, EPI>), grid, dim3(256), 0, st, g);
  return wd::check_launch(what);
}

// ---- small per-layer kernels -------------------------------------------------------------------

// s[k] = gamma*inv (or 1), t[k] = beta (or 0);  Wf = diag(s) W;  bf_part[c] = (c == 0 ? b : 0) + t_c^T W_c
// grid (ceil(N/64), WD_FOLD_PARTS): block = 64 columns x 4 k-lanes over one K chunk; the partial bias
// sums stay separate per chunk (deterministic) and are added up by the GEMM epilogue.
__global__ void __launch_bounds__(256)
k_fold_affine(const float *__restrict__ P, int64_t w_off, int64_t b_off, const int32_t *__restrict__ gamma_idx,
              const int32_t *__restrict__ beta_idx, float inv, float *__restrict__ Wf, float *__restrict__ bf,
              float *__restrict__ s_out, float *__restrict__ t_out, int64_t K, int64_t N) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + tx;
  const int64_t kc = (K + WD_FOLD_PARTS - 1) / WD_FOLD_PARTS;
  const int64_t k0 = (int64_t)blockIdx.y * kc;
  const int64_t k1 = k0 + kc < K ? k0 + kc : K;
  const float *W = P + w_off;
  float tb = 0.f;
  for (int64_t k = k0 + ty; k < k1; k += 4) {
    const int32_t gi = gamma_idx ? gamma_idx[k] : -1;
    const int32_t bi = beta_idx ? beta_idx[k] : -1;
    const float sk = gi >= 0 ? P[gi] * inv : 1.0f;
    const float tk = bi >= 0 ? P[bi] : 0.0f;
    if (blockIdx.x == 0 && tx == 0) {
      s_out[k] = sk;
      t_out[k] = tk;
    }
    if (n < N) {
      const float w = W[k * N + n];
      Wf[k * N + n] = sk * w;
      tb += tk * w;
    }
  }
  red[ty][tx] = tb;
  __syncthreads();
  if (ty == 0 && n < N) {
    float v = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
    if (blockIdx.y == 0) v += P[b_off + n];
    bf[(int64_t)blockIdx.y * N + n] = v;
  }
}

__global__ void __launch_bounds__(256)
k_act_bwd(const float *__restrict__ da, int64_t ldda, const float *__restrict__ a, int64_t lda, int32_t act,
          float *__restrict__ dz, int64_t lddz, int64_t M, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  dz[m * lddz + n] = da[m * ldda + n] * act_bwd(a[m * lda + n], act);
}

// One block per input column k (row of W): reduce the split-K partials, emit dW row, and the
// affine-parameter gradients of the producer of column k.
__global__ void __launch_bounds__(256)
k_mlp_finalize(const float *__restrict__ Gpart, int32_t nsplit, const float *__restrict__ P, int64_t w_off,
               int64_t b_off, const float *__restrict__ s, const float *__restrict__ t,
               const int32_t *__restrict__ gamma_idx, const int32_t *__restrict__ beta_idx, float inv,
               float *__restrict__ Gflat, int64_t K, int64_t N) {
  __shared__ float red_s[4], red_t[4];
  const int64_t k = blockIdx.x;  // 0..K (row K = ones row = db)
  const int64_t split_stride = (K + 1) * N;
  const float *W = P + w_off;
  float acc_s = 0.f, acc_t = 0.f;
  for (int64_t n = threadIdx.x; n < N; n += blockDim.x) {
    float gk = 0.f, db = 0.f;
    for (int32_t z = 0; z < nsplit; ++z) {
      gk += Gpart[z * split_stride + k * N + n];
      db += Gpart[z * split_stride + K * N + n];
    }
    if (k == K) {
      Gflat[b_off + n] = db;
    } else {
      Gflat[w_off + k * N + n] = s[k] * gk + t[k] * db;
      const float w = W[k * N + n];
      acc_s += w * gk;
      acc_t += w * db;
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
    // several consumer layers may feed the same producer (dense / resnet modes): accumulate.
    // Launches are stream-ordered and k is unique within a launch, so no atomics are needed.
    if (gi >= 0) Gflat[gi] += ss * inv;
    if (bi >= 0) Gflat[bi] += tt;
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

extern "C" int wd_gemm_tn_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, float *Cpart, int64_t M,
                                 int64_t N, int64_t K, int32_t nsplit, int32_t append_ones, wd_stream_t stream) {
  if (M <= 0 || N <= 0) return WD_OK;
  WD_REQUIRE(A && B && Cpart, "null pointer");
  WD_REQUIRE(K > 0 && nsplit > 0, "K and nsplit must be > 0");
  GemmArgs g{};
  const int64_t Mo = append_ones ? M + 1 : M;
  g.A = A; g.B = B; g.C = Cpart; g.lda = lda; g.ldb = ldb; g.ldc = N;
  g.M = Mo; g.N = N; g.K = K;
  g.kchunk = wd::ceil_div(wd::ceil_div(K, nsplit), BK) * BK;
  g.c_split = Mo * N;
  g.ones_row = append_ones ? M : -1;
  return launch_gemm<false, false, 2>(g, nsplit, wd::as_stream(stream), "wd_gemm_tn_splitk");
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

extern "C" int wd_mlp_finalize(const float *Gpart, int32_t nsplit, const float *P, int64_t w_off, int64_t b_off,
                               const float *s, const float *t, const int32_t *gamma_idx, const int32_t *beta_idx,
                               float inv, float *Gflat, int64_t K, int64_t N, wd_stream_t stream) {
  WD_REQUIRE(Gpart && P && s && t && Gflat, "null pointer");
  WD_REQUIRE(K > 0 && N > 0 && nsplit > 0, "K, N, nsplit must be > 0");
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
