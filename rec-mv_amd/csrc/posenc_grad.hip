// Derivative kernels of the positional encoding (model/Embedder.py:4-65), so that every order of autograd the
// loss needs (the SDF normal / eikonal terms differentiate d gamma / dx again) is ONE launch instead of the
// ~40 elementwise launches (13 sin/cos + scale + cat, forward and again in backward) of a torch-op encoding.
//
//   gamma(x)[c]          = x_c
//   gamma(x)[3+6b+c]     = w_{2b}   sin(f_b x_c)          f_b = 2^b
//   gamma(x)[3+6b+3+c]   = w_{2b+1} cos(f_b x_c)
// gamma is separable per coordinate c, so its Jacobian is "diagonal":
//   vjp (x, g)    -> gx[c]  = g[c] + sum_b f_b ( w_{2b} cos(f_b x_c) g[3+6b+c] - w_{2b+1} sin(f_b x_c) g[3+6b+3+c] )
//   jvp (x, t)    -> out    = J(x) t  (same layout as gamma)
//   vjp2(x, g, t) -> out[c] = t_c * sum_b f_b^2 ( -w_{2b} sin(f_b x_c) g[3+6b+c] - w_{2b+1} cos(f_b x_c) g[3+6b+3+c] )
// vjp2 is d/dx of both <g, J(x) t> forms that appear when vjp or jvp is differentiated.
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;
struct PeW {
  float w[32];
};

__global__ __launch_bounds__(kBlk) void pe_vjp_kernel(const float* __restrict__ x, int64_t ldx,
                                                      const float* __restrict__ g, int64_t ldg,
                                                      const float* __restrict__ t /*null: plain vjp*/, int64_t ldt,
                                                      float* __restrict__ out, int64_t P, int L, PeW w) {
  const int64_t total = P * 3;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlk) {
    const int64_t p = e / 3;
    const int c = (int)(e % 3);
    const float xc = x[p * ldx + c];
    const float* gp = g + p * ldg;
    float acc = t ? 0.f : gp[c];
    for (int b = 0; b < L; ++b) {
      const float f = (float)(1 << b);
      float s, co;
      sincosf(xc * f, &s, &co);
      const float gs = gp[3 + 6 * b + c], gc = gp[3 + 6 * b + 3 + c];
      if (t)
        acc += f * f * (-w.w[2 * b] * s * gs - w.w[2 * b + 1] * co * gc);
      else
        acc += f * (w.w[2 * b] * co * gs - w.w[2 * b + 1] * s * gc);
    }
    out[e] = t ? acc * t[p * ldt + c] : acc;
  }
}

__global__ __launch_bounds__(kBlk) void pe_jvp_kernel(const float* __restrict__ x, int64_t ldx,
                                                      const float* __restrict__ t, int64_t ldt,
                                                      float* __restrict__ out, int64_t ldo, int64_t P, int L, PeW w) {
  const int nf = 1 + 2 * L;
  const int64_t total = P * nf;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlk) {
    const int64_t p = e / nf;
    const int fidx = (int)(e % nf);
    float* o = out + p * ldo + 3 * fidx;
    const float* xp = x + p * ldx;
    const float* tp = t + p * ldt;
    if (fidx == 0) {
      o[0] = tp[0];
      o[1] = tp[1];
      o[2] = tp[2];
    } else {
      const int b = (fidx - 1) >> 1;
      const float f = (float)(1 << b);
      const float wt = w.w[fidx - 1] * f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float a = xp[c] * f;
        o[c] = ((fidx - 1) & 1) ? -wt * sinf(a) * tp[c] : wt * cosf(a) * tp[c];
      }
    }
  }
}

inline void fill_w(PeW* pw, const float* weights_host, int L) {
  for (int i = 0; i < 32; ++i) pw->w[i] = (weights_host && i < 2 * L) ? weights_host[i] : 1.f;
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int recmv_posenc_vjp(const float* x, int64_t ldx, const float* g, int64_t ldg, const float* t,
                                int64_t ldt, float* out, int64_t P, int L, const float* weights_host,
                                void* stream) {
  RECMV_REQUIRE(P >= 0 && L >= 0 && L <= 16, "posenc_vjp: bad size");
  if (P == 0) return RECMV_OK;
  RECMV_REQUIRE(x && g && out && ldx >= 3 && ldg >= 3 + 6 * L && (!t || ldt >= 3 || ldt == 0), "posenc_vjp: bad argument");
  PeW w;
  fill_w(&w, weights_host, L);
  hipLaunchKernelGGL(pe_vjp_kernel, dim3(stream_grid(P * 3, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, x, ldx, g, ldg,
                     t, ldt, out, P, L, w);
  return check_launch("posenc_vjp");
}

extern "C" int recmv_posenc_jvp(const float* x, int64_t ldx, const float* t, int64_t ldt, float* out, int64_t ldo,
                                int64_t P, int L, const float* weights_host, void* stream) {
  RECMV_REQUIRE(P >= 0 && L >= 0 && L <= 16, "posenc_jvp: bad size");
  if (P == 0) return RECMV_OK;
  RECMV_REQUIRE(x && t && out && ldx >= 3 && (ldt >= 3 || ldt == 0) && ldo >= 3 + 6 * L, "posenc_jvp: bad argument");
  PeW w;
  fill_w(&w, weights_host, L);
  hipLaunchKernelGGL(pe_jvp_kernel, dim3(stream_grid(P * (1 + 2 * L), kBlk)), dim3(kBlk), 0, (hipStream_t)stream, x,
                     ldx, t, ldt, out, ldo, P, L, w);
  return check_launch("posenc_jvp");
}
