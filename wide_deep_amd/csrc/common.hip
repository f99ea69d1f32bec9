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
