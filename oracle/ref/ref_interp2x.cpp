/*
 * ref_interp2x.cpp — TEST INFRASTRUCTURE (oracle/_ref): runs the REFERENCE's own 2x boundary upsampler kernels on the host.
 *
 * The kernel text — the anonymous namespace of /root/reference/MCAcc/cuda/interp2x_boundary3d_kernel.cu:8-240
 * (interp2x_boundary3d_cuda_forward_kernel, ..._backward_kernel) — is cut out of the reference tree at build time into
 * oracle/_ref/interp2x_kernels_extract.inc (git-ignored build output) and compiled through torch_host_shim.h.  Written here:
 * only the launch wrappers (ibid. :243-304): one "thread" per output (forward) / input-gradient (backward) element instead
 * of `<<<blocks, 1024>>>`; tensors are contiguous [B,C,D,H,W].
 */
#include "torch_host_shim.h"

#include "interp2x_kernels_extract.inc"

template <typename T>
static torch::PackedTensorAccessor32<T, 5> acc(T* data, const int* sizes, int* strides) {
  strides[4] = 1;
  for (int i = 3; i >= 0; --i) strides[i] = strides[i + 1] * sizes[i + 1];
  return {data, sizes, strides};
}

template <typename T>
static void fwd(const T* in, const int* is, T* out, bool* boundary, const int* os, float balance) {
  int ist[5], ost[5], bst[5];
  auto a_in = acc<T>(const_cast<T*>(in), is, ist);
  auto a_out = acc<T>(out, os, ost);
  auto a_b = acc<bool>(boundary, os, bst);
  const long n = (long)os[0] * os[1] * os[2] * os[3] * os[4];
  blockDim.x = 1; threadIdx.x = 0;
  for (long i = 0; i < n; ++i) { blockIdx.x = (int)i; interp2x_boundary3d_cuda_forward_kernel<T>(a_in, a_out, a_b, balance); }
}

template <typename T>
static void bwd(const T* go, const int* gos, T* gi, const int* gis) {
  int gost[5], gist[5];
  auto a_go = acc<T>(const_cast<T*>(go), gos, gost);
  auto a_gi = acc<T>(gi, gis, gist);
  const long n = (long)gis[0] * gis[1] * gis[2] * gis[3] * gis[4];
  blockDim.x = 1; threadIdx.x = 0;
  for (long i = 0; i < n; ++i) { blockIdx.x = (int)i; interp2x_boundary3d_cuda_backward_kernel<T>(a_go, a_gi); }
}

extern "C" {
void ref_interp2x_forward_float(const float* in, const int* is, float* out, bool* b, const int* os, float bal) { fwd<float>(in, is, out, b, os, bal); }
void ref_interp2x_forward_double(const double* in, const int* is, double* out, bool* b, const int* os, float bal) { fwd<double>(in, is, out, b, os, bal); }
void ref_interp2x_backward_float(const float* go, const int* gos, float* gi, const int* gis) { bwd<float>(go, gos, gi, gis); }
void ref_interp2x_backward_double(const double* go, const int* gos, double* gi, const int* gis) { bwd<double>(go, gos, gi, gis); }
}
