// wide_deep_amd/csrc/common.h -- shared helpers for the gfx950 kernels (host side of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/wd_hip.h"

#define WD_WT_TOWER 1
#define WD_WT_PRODUCTS 2
#define WD_WT_ROW_UPDATE 4
#define WD_WT_PREFETCH 8
#define WD_WT_DEFAULT 1

// Small latency-bound kernels that run BESIDE a long MFMA kernel (hash, routing, bucketing, exchange packing, pooling: a few loads
// per wavefront) raise their issue priority: the tower's wavefronts (two per SIMD, an MFMA / LDS stream without gaps) otherwise
// keep them waiting at the instruction arbiter -- the input-layer launch of the step went 26 -> 13 us that way, the tower did not
// notice (profiles/r4_gather_instep.txt).  -DWD_NO_SIDE_PRIO builds without it (A/B).
#ifdef WD_NO_SIDE_PRIO
#define WD_SIDE_PRIO() do { } while (0)
#else
#define WD_SIDE_PRIO() __builtin_amdgcn_s_setprio(2)
#endif

namespace wd {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return WD_ERR_LAUNCH;
  }
  return WD_OK;
}

inline hipStream_t as_stream(wd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;       // CDNA4 wavefront
constexpr int kCUs = 256;       // MI355X
constexpr int kXCDs = 8;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- write-through stores.  A kernel's plain stores stay dirty in its XCD's L2 until the end-of-kernel release writes them back
// (tens of MB per launch of this step: ~4 us at the end of a kernel on the critical path); stored at device scope they go
// through to memory while the kernel still computes.  Measured on the tower's outputs: -3.8 us per step (profiles/README.md).
typedef float wd_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store4_wt(float4 *p, float4 v) {
  const wd_f32x4 x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void store1(float *p, float v, bool wt) {
  if (wt) store_wt(p, v);
  else *p = v;
}
__device__ __forceinline__ void store4(float4 *p, float4 v, bool wt) {
  if (wt) store4_wt(p, v);
  else *p = v;
}
// WD_WT bit mask (diagnostics / A-B): which kernels store write-through
inline int wt_mask() {
  static const int m = getenv("WD_WT") ? atoi(getenv("WD_WT")) : WD_WT_DEFAULT;
  return m;
}

// ---- one dense parameter's share of the dense tail (python/lib/joint.py:233-241, tf.train.AdagradOptimizer on the dnn scope),
// k_chain_tail (mlp_chain.hip).
struct TailCtx {
  float *P, *Pacc, *Gflat;
  float inv, lr;
};

// gradient gv of parameter idx is final: store it, take the Adagrad step on (w, a) = (P[idx], Pacc[idx]) as loaded by the
// caller, and rewrite the two MFMA-packed copies of a hidden-layer kernel element e = k * N + n (e < 0: a vector parameter)
__device__ __forceinline__ void tail_apply(const wd_tail_layer_t &L, const TailCtx &c, int64_t idx, int64_t e, float gv, float w,
                                           float a, bool grad, bool upd, bool pack) {
  if (grad) c.Gflat[idx] = gv;
  if (upd) {
    const float gg = grad ? gv : c.Gflat[idx];      // update without grad: the (all-reduced) gradient buffer
    a = a + gg * gg;
    c.Pacc[idx] = a;
    w -= c.lr * gg / sqrtf(a);
    c.P[idx] = w;
  }
  if (pack && e >= 0 && L.Wpk) {
    const int64_t K = L.K, N = L.N;
    const int64_t k = e / N, n = e - k * N;
    L.Wpk[((n >> 5) * (K >> 3) + (k >> 3)) * 256 + (((k & 1) << 5) + (n & 31)) * 4 + ((k & 7) >> 1)] = w;
    if (L.nseg > 0) {
      // concatenating tower: row k of the kernel belongs to one segment of the layer's window; the element goes to that
      // segment's pull operand B[r][c], c = the row's column inside the segment, r = this layer's first reduction row there + n
      for (int q = 0; q < L.nseg; ++q) {
        const int64_t u = k - L.seg_k0[q];
        if (u < 0 || u >= L.seg_w[q]) continue;
        if (L.seg_wt[q]) {
          const int64_t r = L.seg_red0[q] + n, R = L.seg_kred[q];
          L.seg_wt[q][((u >> 5) * (R >> 3) + (r >> 3)) * 256 + (((r & 1) << 5) + (u & 31)) * 4 + ((r & 7) >> 1)] = w;
        }
        break;
      }
    } else if (L.WTpk) {
      L.WTpk[((k >> 5) * (N >> 3) + (n >> 3)) * 256 + (((n & 1) << 5) + (k & 31)) * 4 + ((n & 7) >> 1)] = w;
    }
  }
}

// element e of layer L's K*N + 3*N tail elements (kernel | bias | gamma | beta), gradient from the layer's partials / sums
__device__ __forceinline__ void tail_element(const wd_tail_layer_t &L, const TailCtx &c, int64_t e, int32_t mode) {
  const int64_t K = L.K, N = L.N;
  const int64_t nW = K * N;
  const bool grad = mode & WD_TAIL_GRAD, upd = mode & WD_TAIL_UPDATE, pack = mode & WD_TAIL_PACK;
  if (e >= nW + 3 * N) return;
  int64_t idx;
  float gv = 0.f;
  if (e < nW) {                       // kernel element (k, n)
    idx = L.w_off + e;
    if (grad) {
      const int64_t stride = (L.db_sum ? K : K + 1) * N;
      float v[16];
      for (int32_t z0 = 0; z0 < L.nsplit; z0 += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = z0 + u < L.nsplit ? L.Gpart[(z0 + u) * stride + e] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) gv += v[u];
      }
    }
  } else {
    const int64_t n = (e - nW) % N;
    const int which = (int)((e - nW) / N);    // 0 bias, 1 gamma, 2 beta
    if (which == 0) {
      idx = L.b_off + n;
      if (grad) {
        if (L.db_sum) {
          gv = L.db_sum[n];
        } else {                       // partials with an appended bias-gradient row (logits layer: one per row tile)
          const int64_t stride = (K + 1) * N;
          float v[16];
          for (int32_t z0 = 0; z0 < L.nsplit; z0 += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = z0 + u < L.nsplit ? L.Gpart[(z0 + u) * stride + K * N + n] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) gv += v[u];
          }
        }
      }
    } else if (which == 1) {
      if (L.gamma_off < 0) return;
      idx = L.gamma_off + n;
      if (grad) gv = L.dgamma_sum[n] * c.inv;
    } else {
      if (L.beta_off < 0) return;
      idx = L.beta_off + n;
      if (grad) gv = L.dbeta_sum[n];
    }
  }
  tail_apply(L, c, idx, e < nW ? e : -1, gv, c.P[idx], upd ? c.Pacc[idx] : 0.f, grad, upd, pack);
}

}  // namespace wd

#define WD_REQUIRE(cond, msg)                 \
  do {                                        \
    if (!(cond)) {                            \
      wd::set_error("%s: %s", __func__, msg); \
      return WD_ERR_INVALID;                  \
    }                                         \
  } while (0)
