/*
 * torch_host_shim.h — TEST INFRASTRUCTURE (oracle/): what the kernel bodies of
 * /root/reference/MCAcc/cuda/GridSamplerMineKernel.cu and interp2x_boundary3d_kernel.cu use of ATen / torch beyond the
 * CUDA spellings of cuda_host_shim.h, given plain host meanings:
 *
 *   at::cuda::detail::TensorInfo<T, int>        -> {data, sizes[], strides[]}
 *   at::native::detail::GridSamplerInterpolation / GridSamplerPadding -> the enums with ATen's values
 *   torch::PackedTensorAccessor32<T, N>         -> {data, sizes, strides} with size(i) and chained operator[]
 *   ::min / ::max on float / double             -> the CUDA global overloads
 *   atomicAdd on float / double                 -> plain read-modify-write (the drivers run one "thread" at a time)
 *   __forceinline__, __launch_bounds__(n)       -> inline, nothing
 * Nothing under rec-mv_amd/ includes this.
 */
#ifndef RECMV_ORACLE_TORCH_HOST_SHIM_H
#define RECMV_ORACLE_TORCH_HOST_SHIM_H
#include <limits.h>
#include "cuda_host_shim.h"

#define __forceinline__ inline
#define __launch_bounds__(n)

static inline float min(float a, float b) { return a < b ? a : b; }
static inline float max(float a, float b) { return a > b ? a : b; }
static inline double min(double a, double b) { return a < b ? a : b; }
static inline double max(double a, double b) { return a > b ? a : b; }
static inline float atomicAdd(float* p, float v) { float old = *p; *p = old + v; return old; }
static inline double atomicAdd(double* p, double v) { double old = *p; *p = old + v; return old; }

template <typename T, typename I>
struct TensorInfo {
  T* data;
  I sizes[8];
  I strides[8];
};
enum class GridSamplerInterpolation { Bilinear, Nearest };
enum class GridSamplerPadding { Zeros, Border, Reflection };

namespace torch {
template <typename T, int N>
struct PackedTensorAccessor32 {
  T* data;
  const int* sizes;
  const int* strides;
  int size(int i) const { return sizes[i]; }
  PackedTensorAccessor32<T, N - 1> operator[](int i) const { return {data + (long)i * strides[0], sizes + 1, strides + 1}; }
};
template <typename T>
struct PackedTensorAccessor32<T, 1> {
  T* data;
  const int* sizes;
  const int* strides;
  int size(int i) const { return sizes[i]; }
  T& operator[](int i) const { return data[(long)i * strides[0]]; }
};
}  // namespace torch
#endif
