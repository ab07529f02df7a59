// Small fused element-wise / row-wise kernels that sit between the MFMA layers — gfx950.
//
//   act_grad   : out = gy * act'(z) expressed through y = act(z)   (softplus(beta): 1 - exp(-beta y); relu: y > 0;
//                tanh: 1 - y^2)            -> the dZ = dY (.) act'(Z) step of every layer's backward, one launch
//                                             instead of torch's expm1 / neg / mul (or gt / cast / mul) chain
//   act_grad2  : out = a * b * d/dy[act'(z) as a function of y]    (softplus: beta exp(-beta y); relu: 0; tanh: -2y)
//                                          -> the double-backward term of the same step
//   weight_norm: W[r,:] = g[r] * v[r,:] / ||v[r,:]||   (torch.nn.utils.weight_norm, dim=0: model/network.py:82-85)
//                forward and backward (gv, gg from gW), one workgroup per row.
// All memory-bound; 16-byte accesses where the row length allows.
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;

__device__ __forceinline__ float dact(float y, int act, float p) {
  switch (act) {
    case RECMV_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case RECMV_ACT_SOFTPLUS: return -expm1f(-p * y);
    case RECMV_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}
__device__ __forceinline__ float d2act(float y, int act, float p) {
  switch (act) {
    case RECMV_ACT_SOFTPLUS: return p * expf(-p * y);
    case RECMV_ACT_TANH: return -2.f * y;
    default: return 0.f;
  }
}

__global__ __launch_bounds__(kBlk) void act_grad_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                        float* __restrict__ out, int64_t n, int act, float p,
                                                        bool vec_ok) {
  const int64_t n4 = vec_ok ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlk) {
    const float4 a = reinterpret_cast<const float4*>(gy)[i], b = reinterpret_cast<const float4*>(y)[i];
    float4 o;
    o.x = a.x * dact(b.x, act, p);
    o.y = a.y * dact(b.y, act, p);
    o.z = a.z * dact(b.z, act, p);
    o.w = a.w * dact(b.w, act, p);
    reinterpret_cast<float4*>(out)[i] = o;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk)
    out[i] = gy[i] * dact(y[i], act, p);
}

__global__ __launch_bounds__(kBlk) void act_grad2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ y, float* __restrict__ out,
                                                         int64_t n, int act, float p) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk)
    out[i] = a[i] * b[i] * d2act(y[i], act, p);
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < kBlk / 64; ++w) t += sh[w];
  return t;
}

__global__ __launch_bounds__(kBlk) void weight_norm_fwd_kernel(const float* __restrict__ v,
                                                               const float* __restrict__ g, float* __restrict__ W,
                                                               float* __restrict__ norms, int cols) {
  __shared__ float sh[kBlk / 64];
  const int r = blockIdx.x;
  const float* vr = v + (int64_t)r * cols;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += kBlk) s += vr[c] * vr[c];
  const float nrm = sqrtf(block_sum(s, sh));
  const float scale = g[r] / nrm;
  for (int c = threadIdx.x; c < cols; c += kBlk) W[(int64_t)r * cols + c] = vr[c] * scale;
  if (threadIdx.x == 0) norms[r] = nrm;
}

// gv = g/n * (gW - vhat * <vhat, gW>),  gg = <vhat, gW>,  vhat = v/n
__global__ __launch_bounds__(kBlk) void weight_norm_bwd_kernel(const float* __restrict__ v,
                                                               const float* __restrict__ g,
                                                               const float* __restrict__ norms,
                                                               const float* __restrict__ gW, float* __restrict__ gv,
                                                               float* __restrict__ gg, int cols) {
  __shared__ float sh[kBlk / 64];
  const int r = blockIdx.x;
  const float* vr = v + (int64_t)r * cols;
  const float* gr = gW + (int64_t)r * cols;
  const float inv = 1.f / norms[r];
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += kBlk) s += vr[c] * gr[c];
  const float dot = block_sum(s, sh) * inv;   // <vhat, gW>
  const float gs = g[r] * inv;
  for (int c = threadIdx.x; c < cols; c += kBlk) gv[(int64_t)r * cols + c] = gs * (gr[c] - vr[c] * inv * dot);
  if (threadIdx.x == 0) gg[r] = dot;
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int recmv_act_grad(const float* gy, const float* y, float* out, int64_t n, int act, float act_param,
                              void* stream) {
  RECMV_REQUIRE(n >= 0, "act_grad: n < 0");
  if (n == 0) return RECMV_OK;
  RECMV_REQUIRE(gy && y && out, "act_grad: NULL pointer");
  const bool vec_ok =
      ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  hipLaunchKernelGGL(act_grad_kernel, dim3(stream_grid(vec_ok ? n / 4 + 1 : n, kBlk)), dim3(kBlk), 0,
                     (hipStream_t)stream, gy, y, out, n, act, act_param, vec_ok);
  return check_launch("act_grad");
}

extern "C" int recmv_act_grad2(const float* a, const float* b, const float* y, float* out, int64_t n, int act,
                               float act_param, void* stream) {
  RECMV_REQUIRE(n >= 0, "act_grad2: n < 0");
  if (n == 0) return RECMV_OK;
  RECMV_REQUIRE(a && b && y && out, "act_grad2: NULL pointer");
  hipLaunchKernelGGL(act_grad2_kernel, dim3(stream_grid(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, a, b, y, out, n,
                     act, act_param);
  return check_launch("act_grad2");
}

extern "C" int recmv_weight_norm_forward(const float* v, const float* g, float* W, float* norms, int64_t rows,
                                         int64_t cols, void* stream) {
  RECMV_REQUIRE(rows >= 0 && cols > 0 && rows < (1ll << 31) && cols < (1ll << 31), "weight_norm: bad size");
  if (rows == 0) return RECMV_OK;
  RECMV_REQUIRE(v && g && W && norms, "weight_norm_forward: NULL pointer");
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3((unsigned)rows), dim3(kBlk), 0, (hipStream_t)stream, v, g, W, norms,
                     (int)cols);
  return check_launch("weight_norm_forward");
}

extern "C" int recmv_weight_norm_backward(const float* v, const float* g, const float* norms, const float* gW,
                                          float* gv, float* gg, int64_t rows, int64_t cols, void* stream) {
  RECMV_REQUIRE(rows >= 0 && cols > 0 && rows < (1ll << 31) && cols < (1ll << 31), "weight_norm: bad size");
  if (rows == 0) return RECMV_OK;
  RECMV_REQUIRE(v && g && norms && gW && gv && gg, "weight_norm_backward: NULL pointer");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((unsigned)rows), dim3(kBlk), 0, (hipStream_t)stream, v, g, norms, gW,
                     gv, gg, (int)cols);
  return check_launch("weight_norm_backward");
}
