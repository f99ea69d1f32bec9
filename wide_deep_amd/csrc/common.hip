// wide_deep_amd/csrc/common.hip -- error plumbing + trivial fills for the C ABI (include/wd_hip.h).
#include <stdarg.h>
#include "common.h"

namespace wd {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace wd

extern "C" const char *wd_last_error(void) { return wd::g_err; }
extern "C" int wd_abi_version(void) { return 1; }

__global__ void k_fill_f32(float *p, float v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

extern "C" int wd_fill_f32(float *p, float v, int64_t n, wd_stream_t stream) {
  if (n <= 0) return WD_OK;
  int blocks = (int)std::min<int64_t>(wd::ceil_div(n, 256), 2048);
  hipLaunchKernelGGL(k_fill_f32, dim3(blocks), dim3(256), 0, wd::as_stream(stream), p, v, n);
  return wd::check_launch("wd_fill_f32");
}

// ---- diagnostics (bench-only): ceilings the gather kernel is measured against -------------------------------------
// random 64-byte row reads: 4 lanes x float4 per row, `per` independent rows per lane group, one float out per wave
template <int PER>
__global__ void __launch_bounds__(256) k_diag_gather64(const float4 *__restrict__ tab, const int32_t *__restrict__ ids,
                                                       int64_t n, float *__restrict__ out) {
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
  const int lane = threadIdx.x & 3;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int32_t id[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int64_t j = grp * PER + q;
    id[q] = j < n ? ids[j] : -1;
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    if (id[q] >= 0) {
      const float4 r = tab[(int64_t)id[q] * 4 + lane];
      acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
    }
  }
  float v = acc.x + acc.y + acc.z + acc.w;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) out[((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6] = v;
}

// random accesses to up to three tables per item (bench-only): item j touches, in table s, the n16[s] 16-byte pieces at
// base[s] + ids[j] * stride16[s]; 16 lanes per item, `PER` items per lane group in flight; rmw: every piece is written back
// (+1).  Prices the layouts of a table row (separate emb / accumulator / wide lines vs one record).
struct DiagSegs {
  float4 *base[3];
  int64_t stride16[3];
  int32_t n16[3];
  int32_t nseg;
};
template <int PER>
__global__ void __launch_bounds__(256) k_diag_access(DiagSegs sg, const int32_t *__restrict__ ids, int64_t n, int rmw,
                                                     float *__restrict__ out) {
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int lane = threadIdx.x & 15;
  int seg = -1, piece = 0, acc16 = 0;
  for (int s = 0; s < sg.nseg; ++s) {
    if (seg < 0 && lane < acc16 + sg.n16[s]) { seg = s; piece = lane - acc16; }
    acc16 += sg.n16[s];
  }
  float4 r[PER];
  float4 *addr[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int64_t j = grp * PER + q;
    addr[q] = nullptr;
    r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < n && seg >= 0) addr[q] = sg.base[seg] + (int64_t)ids[j] * sg.stride16[seg] + piece;
  }
#pragma unroll
  for (int q = 0; q < PER; ++q)
    if (addr[q]) r[q] = *addr[q];
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    if (addr[q] && rmw) {
      r[q].x += 1.0f;
      *addr[q] = r[q];
    }
    v += r[q].x + r[q].y + r[q].z + r[q].w;
  }
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) out[(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6) & 0xFFFF] = v;
}

extern "C" int wd_diag_access(float *const *base, const int64_t *stride_bytes, const int32_t *bytes, int32_t nseg,
                              const int32_t *ids, int64_t n, int32_t per, int32_t rmw, float *out, wd_stream_t stream) {
  if (n <= 0) return WD_OK;
  WD_REQUIRE(nseg >= 1 && nseg <= 3, "1..3 segments");
  DiagSegs sg{};
  int tot = 0;
  for (int s = 0; s < nseg; ++s) {
    WD_REQUIRE(bytes[s] % 16 == 0 && stride_bytes[s] % 16 == 0, "16-byte pieces");
    sg.base[s] = reinterpret_cast<float4 *>(base[s]);
    sg.stride16[s] = stride_bytes[s] / 16;
    sg.n16[s] = bytes[s] / 16;
    tot += sg.n16[s];
  }
  WD_REQUIRE(tot <= 16, "at most 256 bytes per item");
  sg.nseg = nseg;
  hipStream_t st = wd::as_stream(stream);
  const unsigned blocks = (unsigned)wd::ceil_div(wd::ceil_div(n, per) * 16, 256);
  switch (per) {
    case 1: hipLaunchKernelGGL(k_diag_access<1>, dim3(blocks), dim3(256), 0, st, sg, ids, n, rmw, out); break;
    case 2: hipLaunchKernelGGL(k_diag_access<2>, dim3(blocks), dim3(256), 0, st, sg, ids, n, rmw, out); break;
    case 4: hipLaunchKernelGGL(k_diag_access<4>, dim3(blocks), dim3(256), 0, st, sg, ids, n, rmw, out); break;
    default: wd::set_error("wd_diag_access: per must be 1, 2 or 4"); return WD_ERR_INVALID;
  }
  return wd::check_launch("wd_diag_access");
}

extern "C" int wd_diag_gather64(const float *table, const int32_t *ids, int64_t n, int32_t per, float *out,
                                wd_stream_t stream) {
  if (n <= 0) return WD_OK;
  hipStream_t st = wd::as_stream(stream);
  const float4 *t4 = reinterpret_cast<const float4 *>(table);
  const unsigned blocks = (unsigned)wd::ceil_div(wd::ceil_div(n, per) * 4, 256);
  switch (per) {
    case 1: hipLaunchKernelGGL(k_diag_gather64<1>, dim3(blocks), dim3(256), 0, st, t4, ids, n, out); break;
    case 2: hipLaunchKernelGGL(k_diag_gather64<2>, dim3(blocks), dim3(256), 0, st, t4, ids, n, out); break;
    case 4: hipLaunchKernelGGL(k_diag_gather64<4>, dim3(blocks), dim3(256), 0, st, t4, ids, n, out); break;
    case 8: hipLaunchKernelGGL(k_diag_gather64<8>, dim3(blocks), dim3(256), 0, st, t4, ids, n, out); break;
    default: wd::set_error("wd_diag_gather64: per must be 1, 2, 4 or 8"); return WD_ERR_INVALID;
  }
  return wd::check_launch("wd_diag_gather64");
}

// ---- gather-strategy ceilings (bench-only, scripts/bench_gather_modes.py -> profiles/r3_gather_modes.txt) ----------------------
// n random rows of 16 floats out of records `rstride` floats apart, written to out[j][16] (the pooled write of the input
// layer) -- the work of wd_prefetch_onehot without slots / wide weights / numeric columns -- with the request issued in
// different ways:
//   0  4 lanes x 16 B per row, plain loads                     1  the same, nontemporal loads (what the step's launch does)
//   2  8 lanes x 16 B: the WHOLE 128-byte record is requested   3  one lane per row: four 16-byte loads per lane, 64 rows per
//      (64 bytes of it kept)                                       wavefront instruction group in flight
//   4  LDS-DMA: global_load_lds_dwordx4, 16 B per lane straight 5  mode 1 with nontemporal STORES of the rows
//      into LDS (no VGPR round trip), then LDS -> HBM
//   6  mode 1 without the row write (one float per wavefront):  7  mode 1 with `per` = 4 rows per lane group in flight
//      the read side alone
__global__ void __launch_bounds__(256) k_diag_gather_modes(const float *__restrict__ rec, int64_t rstride,
                                                           const int32_t *__restrict__ ids, int64_t n, int mode,
                                                           float *__restrict__ out) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ f4 stage[256];
  const int t = threadIdx.x;
  const int64_t gt = (int64_t)blockIdx.x * 256 + t;
  if (mode == 2) {
    const int64_t j = gt >> 3;
    const int l = t & 7;
    if (j >= n) return;
    const f4 v = *(reinterpret_cast<const f4 *>(rec + (int64_t)ids[j] * rstride) + l);
    if (l < 4) *(reinterpret_cast<f4 *>(out + j * 16) + l) = v;
    else if (v.x == 12345.678f) out[0] = v.y;          // (keeps the upper half of the record a real request)
    return;
  }
  if (mode == 3) {
    const int64_t j = gt;
    if (j >= n) return;
    const f4 *src = reinterpret_cast<const f4 *>(rec + (int64_t)ids[j] * rstride);
    const f4 a = src[0], b = src[1], c = src[2], d = src[3];
    f4 *dst = reinterpret_cast<f4 *>(out + j * 16);
    dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
    return;
  }
  if (mode == 7) {
    const int64_t g0 = (gt >> 2) * 4;
    const int l = t & 3;
    f4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v[q] = g0 + q < n ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(rec + (int64_t)ids[g0 + q] * rstride) + l) : (f4)(0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (g0 + q < n) *(reinterpret_cast<f4 *>(out + (g0 + q) * 16) + l) = v[q];
    return;
  }
  const int64_t j = gt >> 2;
  const int l = t & 3;
  if (mode == 4) {
    // every lane names its own 16 bytes; the hardware writes lane i's data at (wave-uniform LDS base) + 16 * i
    const int64_t jj = j < n ? j : n - 1;
    const f4 *src = reinterpret_cast<const f4 *>(rec + (int64_t)ids[jj] * rstride) + l;
    f4 *dst = stage + (t & ~63);
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                     (void __attribute__((address_space(3))) *)dst, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0)
    __syncthreads();
    if (j < n) *(reinterpret_cast<f4 *>(out + j * 16) + l) = stage[t];
    return;
  }
  if (j >= n) return;
  const f4 *src = reinterpret_cast<const f4 *>(rec + (int64_t)ids[j] * rstride) + l;
  const f4 v = mode == 0 ? *src : __builtin_nontemporal_load(src);
  if (mode == 6) {
    float x = v.x + v.y + v.z + v.w;
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((t & 63) == 0) out[(gt >> 6) & 0xFFFF] = x;
  } else if (mode == 5) {
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(out + j * 16) + l);
  } else {
    *(reinterpret_cast<f4 *>(out + j * 16) + l) = v;
  }
}

extern "C" int wd_diag_gather_modes(const float *rec, int64_t rec_stride, const int32_t *ids, int64_t n, int32_t mode,
                                    float *out, wd_stream_t stream) {
  if (n <= 0) return WD_OK;
  WD_REQUIRE(rec && ids && out && rec_stride >= 16 && rec_stride % 4 == 0 && mode >= 0 && mode <= 7, "bad arguments");
  const int lanes = mode == 2 ? 8 : (mode == 3 ? 1 : 4);
  const int64_t threads = mode == 7 ? wd::ceil_div(n, 4) * 4 : n * lanes;
  hipLaunchKernelGGL(k_diag_gather_modes, dim3((unsigned)wd::ceil_div(threads, 256)), dim3(256), 0, wd::as_stream(stream), rec,
                     rec_stride, ids, n, mode, out);
  return wd::check_launch("wd_diag_gather_modes");
}
