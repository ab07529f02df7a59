// Deformation regulariser of the render loss, value and gradient in one pass — gfx950.
//
// The reference (OptimGarmentNetwork.py:1143-1155) takes the Jacobian J [P,3,3] of the offset MLP at the sampled points, moves it
// to the HOST, runs torch.svd there, and forms  GM(sum_i log^2 sigma_i)  with the Geman-McClure error of utils/utils.py:87-91
// (square = True: 2 x / c^2 / (x / c^2 + 4)); autograd differentiates through the SVD.  Round 1 replaced the host SVD by the
// closed-form eigenvalues of J^T J written in ~20 torch launches (70 with their backward: loop.py singular_values_3x3).  Here one
// thread owns one matrix:
//   one-sided cyclic Jacobi on J's columns (five sweeps: converged to f32 rounding for any 3x3)  ->  J V = U Sigma, lambda_i = sigma_i^2
//   s_i = log sigma_i = 0.5 log max(lambda_i, 1e-20),  x = sum s_i^2,  y = GM(x)
//   dy/dJ = GM'(x) * 2 J V diag(s_i / lambda_i) V^T        (dx/dlambda_i = s_i / lambda_i, dlambda_i/dA = v_i v_i^T, dA/dJ: J (G + G^T))
// so the backward pass of the term is ONE scaling of the stored dy/dJ.  The analytic form has none of the closed form's
// acos / clamp conditioning near repeated eigenvalues (J close to a rotation — where a converged deformer lives).
// HBM-bound: 36 B in, 4 + 36 B out per matrix.
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;

// One rotation of the ONE-SIDED (Hestenes) Jacobi iteration: columns p and q of b (= J V so far) are turned until they are
// orthogonal, V takes the same turn.  At convergence the columns of b are sigma_i u_i: their norms are the singular values to
// RELATIVE f32 accuracy also when J is nearly singular — which the eigenvalues of an explicitly formed J^T J are not.
template <int p, int q>
__device__ __forceinline__ void jacobi_rotate(float (&b)[3][3], float (&v)[3][3]) {
  const float alpha = b[0][p] * b[0][p] + b[1][p] * b[1][p] + b[2][p] * b[2][p];
  const float beta = b[0][q] * b[0][q] + b[1][q] * b[1][q] + b[2][q] * b[2][q];
  const float gamma = b[0][p] * b[0][q] + b[1][p] * b[1][q] + b[2][p] * b[2][q];
  if (fabsf(gamma) < 1e-37f) return;
  const float zeta = (beta - alpha) / (2.f * gamma);
  const float t = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(zeta * zeta + 1.f));
  const float c = 1.f / sqrtf(t * t + 1.f);
  const float s = t * c;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float bp = b[k][p], bq = b[k][q];
    b[k][p] = c * bp - s * bq;
    b[k][q] = s * bp + c * bq;
    const float vp = v[k][p], vq = v[k][q];
    v[k][p] = c * vp - s * vq;
    v[k][q] = s * vp + c * vq;
  }
}

__global__ __launch_bounds__(kBlk) void def_regu_kernel(const float* __restrict__ J, int64_t P, float inv_c2, float* __restrict__ y,
                                                        float* __restrict__ gJ) {
  __shared__ float tile[kBlk * 9];
  const int64_t base = (int64_t)blockIdx.x * kBlk;
  const int n = (int)((P - base) < kBlk ? (P - base) : kBlk);
  for (int e = threadIdx.x; e < n * 9; e += kBlk) tile[e] = J[base * 9 + e];        // coalesced; stride-9 LDS reads: conflict-free
  __syncthreads();
  const int t = threadIdx.x;
  float g[9];
  if (t < n) {
    float m[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) m[i][j] = tile[t * 9 + 3 * i + j];
    float b[3][3], v[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        b[i][j] = m[i][j];
        v[i][j] = i == j ? 1.f : 0.f;
      }
#pragma unroll
    for (int sweep = 0; sweep < 5; ++sweep) {
      jacobi_rotate<0, 1>(b, v);
      jacobi_rotate<0, 2>(b, v);
      jacobi_rotate<1, 2>(b, v);
    }
    float x = 0.f, w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float lam = b[0][i] * b[0][i] + b[1][i] * b[1][i] + b[2][i] * b[2][i];     // sigma_i^2
      const bool live = lam > 1e-20f;                 // torch.clamp(min=1e-20): value from the bound, no gradient below it
      const float lc = live ? lam : 1e-20f;
      const float s = 0.5f * logf(lc);
      x += s * s;
      w[i] = live ? s / lc : 0.f;
    }
    const float u = x * inv_c2;
    const float d = 1.f / (u + 4.f);
    y[base + t] = 2.f * u * d;
    const float dy = 8.f * inv_c2 * d * d;            // d/dx [2 x/c^2 / (x/c^2 + 4)]
    // dy/dJ = dy * 2 J V diag(w) V^T = dy * 2 (J V) diag(w) V^T, and J V is what b holds
    const float k2 = 2.f * dy;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) g[3 * i + j] = k2 * (b[i][0] * w[0] * v[j][0] + b[i][1] * w[1] * v[j][1] + b[i][2] * w[2] * v[j][2]);
  }
  __syncthreads();
  if (t < n) {
#pragma unroll
    for (int e = 0; e < 9; ++e) tile[t * 9 + e] = g[e];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n * 9; e += kBlk) gJ[base * 9 + e] = tile[e];
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int recmv_def_regu(const float* J, int64_t P, float c, float* y, float* gJ, void* stream) {
  RECMV_REQUIRE(P >= 0, "def_regu: negative P");
  if (P == 0) return RECMV_OK;
  RECMV_REQUIRE(J && y && gJ, "def_regu: NULL pointer");
  RECMV_REQUIRE(c > 0.f, "def_regu: the Geman-McClure scale must be positive (got %g)", (double)c);
  hipLaunchKernelGGL(def_regu_kernel, dim3((unsigned)ceil_div(P, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, J, P, 1.f / (c * c), y,
                     gJ);
  return check_launch("def_regu");
}
