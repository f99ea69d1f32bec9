// wide_deep_amd/csrc/common.h -- shared helpers for the gfx950 kernels (host side of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/wd_hip.h"

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

}  // namespace wd

#define WD_REQUIRE(cond, msg)                 \
  do {                                        \
    if (!(cond)) {                            \
      wd::set_error("%s: %s", __func__, msg); \
      return WD_ERR_INVALID;                  \
    }                                         \
  } while (0)
