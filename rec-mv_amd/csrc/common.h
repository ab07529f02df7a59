// Shared helpers for the gfx950 kernels of librecmv_hip.so (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/recmv_hip.h"

namespace recmv {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return RECMV_ERR_HIP;
  }
  return RECMV_OK;
}

#define RECMV_HIP_TRY(expr)                                              \
  do {                                                                   \
    hipError_t e__ = (expr);                                             \
    if (e__ != hipSuccess) {                                             \
      ::recmv::set_error("%s: %s", #expr, hipGetErrorString(e__));       \
      return RECMV_ERR_HIP;                                              \
    }                                                                    \
  } while (0)

#define RECMV_REQUIRE(cond, ...)                                         \
  do {                                                                   \
    if (!(cond)) {                                                       \
      ::recmv::set_error(__VA_ARGS__);                                   \
      return RECMV_ERR_ARG;                                              \
    }                                                                    \
  } while (0)

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kNumCU = 256;          // MI355X
constexpr int kNumXCD = 8;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid size for memory-bound grid-stride kernels: enough workgroups to fill 256 CUs x 8 and no more.
inline int stream_grid(int64_t work_items, int block) {
  int64_t g = ceil_div(work_items, block);
  const int64_t cap = (int64_t)kNumCU * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace recmv
