// wide_deep_amd/csrc/sparse_update.hip -- backward sparse side: gradient scatter-add + fused optimizers.
//
// Replaces what `dnn_optimizer.minimize` / `linear_optimizer.minimize` do to the sparse variables
// (python/lib/joint.py:224-262 with python/lib/utils/model_util.py:84-90 choosing
// tf.train.AdagradOptimizer / tf.train.FtrlOptimizer): the gradient of an embedding lookup is an
// IndexedSlices whose duplicate rows are SUMMED before the optimizer formula is applied once per
// unique row.
//
// gfx950 design: instead of float atomics on hot rows (which serialise under Zipf skew and make the
// sum order non-deterministic) every (row, bag) occurrence is radix-sorted by row once per step; a
// segment of equal keys is then reduced by ONE lane group in ascending bag order (deterministic) and
// the optimizer update is fused into the same kernel, so each touched row / accumulator line is read
// once and written once.  The sort is shared by the embedding (Adagrad) and wide (FTRL) sides.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace {

// rocPRIM picks merge sort below 1M items (17 launches for the 213k occurrences of a C2 batch); the Onesweep
// radix path (histogram + scan + one launch per 8-bit digit) is ~3x fewer launches at these sizes.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::default_config, 0>;

__global__ void k_build_keys(const wd_slot_t *__restrict__ slots, int32_t S, const int32_t *__restrict__ ids,
                             const int32_t *__restrict__ bag_offs, int64_t nbags, uint32_t *__restrict__ keys,
                             int32_t *__restrict__ vals) {
  const int64_t bag = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (bag >= nbags) return;
  const int64_t base = slots[bag % S].row_base;
  const int32_t j0 = bag_offs[bag], j1 = bag_offs[bag + 1];
  for (int32_t j = j0; j < j1; ++j) {
    keys[j] = (uint32_t)(base + ids[j]);
    vals[j] = (int32_t)bag;
  }
}

// slot lookup for a key: slots are laid out in increasing row_base order
__device__ __forceinline__ int32_t slot_of_bag(int32_t bag, int32_t S) { return bag % S; }

template <int LANES>
__global__ void __launch_bounds__(256)
k_embag_bwd_adagrad(float *__restrict__ emb, float *__restrict__ accum, const wd_slot_t *__restrict__ slots, int32_t S,
                    const uint32_t *__restrict__ keys, const int32_t *__restrict__ vals, int64_t nnz,
                    const int32_t *__restrict__ bag_offs, const float *__restrict__ dx, int64_t ldx, float lr) {
  constexpr int D = LANES * 4;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = tid / LANES;  // sorted occurrence
  const int lane = (int)(tid % LANES);
  if (i >= nnz) return;
  const uint32_t key = keys[i];
  if (i > 0 && keys[i - 1] == key) return;  // not a segment head
  const int32_t bag0 = vals[i];
  const int32_t s = slot_of_bag(bag0, S);
  const wd_slot_t sl = slots[s];
  if (sl.kind != WD_SLOT_EMBEDDING || sl.dim != D) return;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  // segment reduce in ascending bag order (stable sort => deterministic sum order)
  for (int64_t j = i; j < nnz && keys[j] == key; ++j) {
    const int32_t bag = vals[j];
    const int64_t b = bag / S;
    const int32_t len = bag_offs[bag + 1] - bag_offs[bag];
    const float scale = len > 1 ? 1.0f / (float)len : 1.0f;
    const float4 d = *reinterpret_cast<const float4 *>(dx + b * ldx + sl.out_col + lane * 4);
    g.x += d.x * scale; g.y += d.y * scale; g.z += d.z * scale; g.w += d.w * scale;
  }
  const int64_t row = (int64_t)key - sl.row_base;
  const int64_t off = sl.emb_off + row * D + lane * 4;
  float4 a = *reinterpret_cast<float4 *>(accum + off);
  float4 w = *reinterpret_cast<float4 *>(emb + off);
  a.x += g.x * g.x; a.y += g.y * g.y; a.z += g.z * g.z; a.w += g.w * g.w;
  w.x -= lr * g.x / sqrtf(a.x);
  w.y -= lr * g.y / sqrtf(a.y);
  w.z -= lr * g.z / sqrtf(a.z);
  w.w -= lr * g.w / sqrtf(a.w);
  *reinterpret_cast<float4 *>(accum + off) = a;
  *reinterpret_cast<float4 *>(emb + off) = w;
}

__global__ void k_embag_bwd_adagrad_generic(float *__restrict__ emb, float *__restrict__ accum,
                                            const wd_slot_t *__restrict__ slots, int32_t S, int32_t D,
                                            const uint32_t *__restrict__ keys, const int32_t *__restrict__ vals,
                                            int64_t nnz, const int32_t *__restrict__ bag_offs,
                                            const float *__restrict__ dx, int64_t ldx, float lr) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = tid / D;
  const int d = (int)(tid % D);
  if (i >= nnz) return;
  const uint32_t key = keys[i];
  if (i > 0 && keys[i - 1] == key) return;
  const int32_t s = slot_of_bag(vals[i], S);
  const wd_slot_t sl = slots[s];
  if (sl.kind != WD_SLOT_EMBEDDING || sl.dim != D) return;
  float g = 0.f;
  for (int64_t j = i; j < nnz && keys[j] == key; ++j) {
    const int32_t bag = vals[j];
    const int32_t len = bag_offs[bag + 1] - bag_offs[bag];
    const float scale = len > 1 ? 1.0f / (float)len : 1.0f;
    g += dx[(int64_t)(bag / S) * ldx + sl.out_col + d] * scale;
  }
  const int64_t off = sl.emb_off + ((int64_t)key - sl.row_base) * D + d;
  float a = accum[off] + g * g;
  accum[off] = a;
  emb[off] -= lr * g / sqrtf(a);
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

__global__ void __launch_bounds__(256)
k_wide_bwd_ftrl(float *__restrict__ wide, const wd_slot_t *__restrict__ slots, int32_t S,
                const uint32_t *__restrict__ keys, const int32_t *__restrict__ vals, int64_t nnz,
                const float *__restrict__ dlogit, float lr, float l1, float l2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  const uint32_t key = keys[i];
  if (i > 0 && keys[i - 1] == key) return;
  if (!slots[slot_of_bag(vals[i], S)].wide) return;
  float g = 0.f;
  for (int64_t j = i; j < nnz && keys[j] == key; ++j) g += dlogit[vals[j] / S];
  float4 r = *reinterpret_cast<float4 *>(wide + (int64_t)key * 4);  // {w, z, n, -}
  ftrl_update(r.x, r.y, r.z, g, lr, l1, l2);
  *reinterpret_cast<float4 *>(wide + (int64_t)key * 4) = r;
}

// single block: deterministic tree reduction of dlogit, then dense FTRL on the bias triple
__global__ void __launch_bounds__(1024) k_bias_ftrl(float *__restrict__ bias, const float *__restrict__ dlogit,
                                                    int64_t batch, float lr, float l1, float l2) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int64_t i0 = threadIdx.x; i0 < batch; i0 += 8 * (int64_t)blockDim.x) {      // eight loads in flight, same order of adds
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = i0 + k * (int64_t)blockDim.x < batch ? dlogit[i0 + k * (int64_t)blockDim.x] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * (int64_t)blockDim.x < batch) acc += v[k];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float g = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) g += red[k];
    float w = bias[0], z = bias[1], n = bias[2];
    ftrl_update(w, z, n, g, lr, l1, l2);
    bias[0] = w; bias[1] = z; bias[2] = n;
  }
}

}  // namespace

extern "C" size_t wd_sort_workspace_bytes(int64_t nnz, int32_t key_bits) {
  size_t bytes = 0;
  if (nnz <= 0) return 256;
  hipError_t e = rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)nnz, 0u,
                                           (unsigned)key_bits, (hipStream_t)0);
  if (e != hipSuccess) return 0;
  return bytes + 256;
}

extern "C" int wd_build_sort_keys(const wd_slot_t *slots, int32_t S, const int32_t *ids, const int32_t *bag_offs,
                                  int64_t nbags, int64_t nnz, uint32_t *keys, int32_t *vals, wd_stream_t stream) {
  if (nbags <= 0 || nnz <= 0) return WD_OK;
  WD_REQUIRE(slots && ids && bag_offs && keys && vals, "null pointer");
  hipLaunchKernelGGL(k_build_keys, dim3((unsigned)wd::ceil_div(nbags, 256)), dim3(256), 0, wd::as_stream(stream),
                     slots, S, ids, bag_offs, nbags, keys, vals);
  return wd::check_launch("wd_build_sort_keys");
}

extern "C" int wd_sort_pairs(const uint32_t *keys_in, const int32_t *vals_in, uint32_t *keys_out, int32_t *vals_out,
                             int64_t nnz, int32_t key_bits, void *workspace, size_t workspace_bytes,
                             wd_stream_t stream) {
  if (nnz <= 0) return WD_OK;
  WD_REQUIRE(keys_in && vals_in && keys_out && vals_out && workspace, "null pointer");
  WD_REQUIRE(key_bits >= 1 && key_bits <= 32, "key_bits out of range");
  size_t need = 0;
  (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, need, keys_in, keys_out, vals_in, vals_out, (size_t)nnz, 0u, (unsigned)key_bits,
                            wd::as_stream(stream));
  if (need > workspace_bytes) {
    wd::set_error("wd_sort_pairs: workspace too small (%zu < %zu)", workspace_bytes, need);
    return WD_ERR_WORKSPACE;
  }
  hipError_t e = rocprim::radix_sort_pairs<SortConfig>(workspace, need, keys_in, keys_out, vals_in, vals_out, (size_t)nnz, 0u,
                                           (unsigned)key_bits, wd::as_stream(stream));
  if (e != hipSuccess) {
    wd::set_error("wd_sort_pairs: %s", hipGetErrorString(e));
    return WD_ERR_LAUNCH;
  }
  return WD_OK;
}

extern "C" int wd_embag_bwd_adagrad(float *emb, float *emb_accum, const wd_slot_t *slots, int32_t S, int32_t dim,
                                    const uint32_t *keys_sorted, const int32_t *vals_sorted, int64_t nnz,
                                    const int32_t *bag_offs, const float *dx, int64_t ldx, float lr,
                                    wd_stream_t stream) {
  if (nnz <= 0) return WD_OK;
  WD_REQUIRE(emb && emb_accum && slots && keys_sorted && vals_sorted && bag_offs && dx, "null pointer");
  hipStream_t st = wd::as_stream(stream);
#define WD_LAUNCH_BWD(L)                                                                                          \
  hipLaunchKernelGGL(k_embag_bwd_adagrad<L>, dim3((unsigned)wd::ceil_div(nnz * L, 256)), dim3(256), 0, st, emb,   \
                     emb_accum, slots, S, keys_sorted, vals_sorted, nnz, bag_offs, dx, ldx, lr)
  switch (dim) {
    case 4: WD_LAUNCH_BWD(1); break;
    case 8: WD_LAUNCH_BWD(2); break;
    case 16: WD_LAUNCH_BWD(4); break;
    case 32: WD_LAUNCH_BWD(8); break;
    case 64: WD_LAUNCH_BWD(16); break;
    case 128: WD_LAUNCH_BWD(32); break;
    default:
      hipLaunchKernelGGL(k_embag_bwd_adagrad_generic, dim3((unsigned)wd::ceil_div(nnz * dim, 256)), dim3(256), 0, st,
                         emb, emb_accum, slots, S, dim, keys_sorted, vals_sorted, nnz, bag_offs, dx, ldx, lr);
  }
#undef WD_LAUNCH_BWD
  return wd::check_launch("wd_embag_bwd_adagrad");
}

extern "C" int wd_wide_bwd_ftrl(float *wide, const wd_slot_t *slots, int32_t S, const uint32_t *keys_sorted,
                                const int32_t *vals_sorted, int64_t nnz, const float *dlogit, float lr, float l1,
                                float l2, wd_stream_t stream) {
  if (nnz <= 0) return WD_OK;
  WD_REQUIRE(wide && slots && keys_sorted && vals_sorted && dlogit, "null pointer");
  hipLaunchKernelGGL(k_wide_bwd_ftrl, dim3((unsigned)wd::ceil_div(nnz, 256)), dim3(256), 0, wd::as_stream(stream),
                     wide, slots, S, keys_sorted, vals_sorted, nnz, dlogit, lr, l1, l2);
  return wd::check_launch("wd_wide_bwd_ftrl");
}

extern "C" int wd_bias_ftrl(float *bias_wzn, const float *dlogit, int64_t batch, float lr, float l1, float l2,
                            wd_stream_t stream) {
  WD_REQUIRE(bias_wzn && dlogit, "null pointer");
  hipLaunchKernelGGL(k_bias_ftrl, dim3(1), dim3(1024), 0, wd::as_stream(stream), bias_wzn, dlogit, batch, lr, l1, l2);
  return wd::check_launch("wd_bias_ftrl");
}
