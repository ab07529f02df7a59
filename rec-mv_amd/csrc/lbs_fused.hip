// Fused linear-blend-skinning passes and the surface root finder's per-ray kernels — gfx950.
//
// Reference path (model/Deformer.py:405-445, utils/FindSurfacePs.py:316-337): per evaluation of the deformer on
// the ray points, a grid sample of the 24 skinning weights, a [P,24]x[24,16] blend, a per-batch-id python loop
// building T, the transform, then cross products / norms / arcsin for the ray energy — about 40 small launches
// forward and as many backward.  Here, one lane per point:
//
//   lbs_forward_kernel : normalise p, gather the 8 corner records (24 contiguous channels each, channels-last
//                        volume), blend T = sum_j w_j A[frame,j] with A staged in LDS, d = T [p;1] + trans,
//                        and (when rays are given) the ray energy |(d-c) x v| / |d-c|, its angle in degrees and
//                        its gradient wrt d.
//   lbs_vjp_kernel     : J_d(p)^T g — the same gather once more instead of saving [P,24] weights and [P,8,24]
//                        corners: g_w_j = g . A[frame,j] [p;1], sampler gradient wrt the grid coordinates
//                        (clip mask and W/2 scale as GridSamplerMineKernel.cu:534-545), plus T_33^T g.
//   rootfind_update    : stopping test, steepest-descent/Newton step p <- p - E grad / |grad|^2 on the still
//                        unfinished rays, and the count of unfinished rays for the host's early exit.
//
// HBM bytes per point: forward 12 (p) + 8 (frame id) + 36 (d, g_d...) + <= 768 gathered; all memory-bound.
#include <limits.h>
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;
constexpr int kJ = 24;             // SMPL joints
constexpr float kRad2Deg = 57.29577951308232f;

#include "gs3d_common.inc"

struct LbsGeom {
  int64_t D, H, W;                 // volume [D,H,W,24] channels-last
  float cx, cy, cz;                // bbox centre
  float sx, sy, sz;                // 2 / bbox extent
};

__device__ __forceinline__ void load24(const float* __restrict__ rec, float* v) {
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const float4 t = reinterpret_cast<const float4*>(rec)[q];
    v[4 * q] = t.x;
    v[4 * q + 1] = t.y;
    v[4 * q + 2] = t.z;
    v[4 * q + 3] = t.w;
  }
}

// A_s: [B][24][12] (rows 0..2 of the 4x4 transforms) in LDS
__device__ __forceinline__ void stage_A(const float* __restrict__ A, int B, float* A_s) {
  for (int e = threadIdx.x; e < B * kJ * 12; e += kBlk) {
    const int bj = e / 12, r = e % 12;
    A_s[e] = A[(int64_t)bj * 16 + r];
  }
  __syncthreads();
}

__global__ __launch_bounds__(kBlk) void lbs_forward_kernel(
    const float* __restrict__ ps, const int64_t* __restrict__ frame, const float* __restrict__ A,
    const float* __restrict__ trans, int B, const float* __restrict__ vol, LbsGeom G, int64_t P,
    const float* __restrict__ cam, const float* __restrict__ rays, float* __restrict__ d_out,
    float* __restrict__ loss2, float* __restrict__ angle, float* __restrict__ g_d) {
  extern __shared__ __attribute__((aligned(16))) float A_s[];
  stage_A(A, B, A_s);
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    const float px = ps[3 * i], py = ps[3 * i + 1], pz = ps[3 * i + 2];
    const int b = (int)frame[i];
    const Cell<float> c = make_cell<float>((px - G.cx) * G.sx, (py - G.cy) * G.sy, (pz - G.cz) * G.sz, G.W, G.H, G.D);
    float w[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) w[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      if (c.in_x[bx] && c.in_y[by] && c.in_z[bz]) {
        const float wk = c.fx[bx] * c.fy[by] * c.fz[bz];
        float v[kJ];
        load24(vol + (((int64_t)(c.z0 + bz) * G.H + (c.y0 + by)) * G.W + (c.x0 + bx)) * kJ, v);
#pragma unroll
        for (int j = 0; j < kJ; ++j) w[j] = fma(v[j], wk, w[j]);
      }
    }
    float T[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) T[r] = 0.f;
    const float* Ab = A_s + b * kJ * 12;
#pragma unroll 4
    for (int j = 0; j < kJ; ++j) {
#pragma unroll
      for (int r = 0; r < 12; ++r) T[r] = fma(w[j], Ab[j * 12 + r], T[r]);
    }
    const float dx = T[0] * px + T[1] * py + T[2] * pz + T[3] + trans[3 * b];
    const float dy = T[4] * px + T[5] * py + T[6] * pz + T[7] + trans[3 * b + 1];
    const float dz = T[8] * px + T[9] * py + T[10] * pz + T[11] + trans[3 * b + 2];
    d_out[3 * i] = dx;
    d_out[3 * i + 1] = dy;
    d_out[3 * i + 2] = dz;
    if (rays) {
      // E2 = |(d-c) x v| / |d-c|  (utils/FindSurfacePs.py:322-325) and dE2/dd
      const float ex = dx - cam[0], ey = dy - cam[1], ez = dz - cam[2];
      const float vx = rays[3 * i], vy = rays[3 * i + 1], vz = rays[3 * i + 2];
      const float ux = ey * vz - ez * vy, uy = ez * vx - ex * vz, uz = ex * vy - ey * vx;
      const float un = sqrtf(ux * ux + uy * uy + uz * uz), dn = sqrtf(ex * ex + ey * ey + ez * ez);
      const float ratio = un / dn;
      loss2[i] = ratio;
      angle[i] = asinf(ratio) * kRad2Deg;
      const float inv = 1.f / fmaxf(un * dn, 1e-30f);
      const float gux = ux * inv, guy = uy * inv, guz = uz * inv;
      const float k3 = un / (dn * dn * dn);
      // cross(v, g_up) - direct * un/dn^3
      g_d[3 * i] = (vy * guz - vz * guy) - ex * k3;
      g_d[3 * i + 1] = (vz * gux - vx * guz) - ey * k3;
      g_d[3 * i + 2] = (vx * guy - vy * gux) - ez * k3;
    }
  }
}

__global__ __launch_bounds__(kBlk) void lbs_vjp_kernel(const float* __restrict__ ps,
                                                       const int64_t* __restrict__ frame,
                                                       const float* __restrict__ A, int B,
                                                       const float* __restrict__ vol, LbsGeom G, int64_t P,
                                                       const float* __restrict__ g_d, float* __restrict__ g_p) {
  extern __shared__ __attribute__((aligned(16))) float A_s[];
  stage_A(A, B, A_s);
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    const float px = ps[3 * i], py = ps[3 * i + 1], pz = ps[3 * i + 2];
    const float gx = g_d[3 * i], gy = g_d[3 * i + 1], gz = g_d[3 * i + 2];
    const int b = (int)frame[i];
    const Cell<float> c = make_cell<float>((px - G.cx) * G.sx, (py - G.cy) * G.sy, (pz - G.cz) * G.sz, G.W, G.H, G.D);
    const float* Ab = A_s + b * kJ * 12;
    // cotangent of the blend weights: gw_j = g . (A_j [p;1])
    float gw[kJ];
#pragma unroll 4
    for (int j = 0; j < kJ; ++j) {
      const float* a = Ab + j * 12;
      const float r0 = a[0] * px + a[1] * py + a[2] * pz + a[3];
      const float r1 = a[4] * px + a[5] * py + a[6] * pz + a[7];
      const float r2 = a[8] * px + a[9] * py + a[10] * pz + a[11];
      gw[j] = gx * r0 + gy * r1 + gz * r2;
    }
    float w[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) w[j] = 0.f;
    float gix = 0.f, giy = 0.f, giz = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      if (c.in_x[bx] && c.in_y[by] && c.in_z[bz]) {
        float v[kJ];
        load24(vol + (((int64_t)(c.z0 + bz) * G.H + (c.y0 + by)) * G.W + (c.x0 + bx)) * kJ, v);
        const float wk = c.fx[bx] * c.fy[by] * c.fz[bz];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
          w[j] = fma(v[j], wk, w[j]);
          dot = fma(v[j], gw[j], dot);
        }
        const float tx = dot * c.fy[by] * c.fz[bz], ty = dot * c.fx[bx] * c.fz[bz], tz = dot * c.fx[bx] * c.fy[by];
        gix = bx ? gix + tx : gix - tx;
        giy = by ? giy + ty : giy - ty;
        giz = bz ? giz + tz : giz - tz;
      }
    }
    gix = c.mx * (float)((double)(gix * (float)G.W) / 2.);
    giy = c.my * (float)((double)(giy * (float)G.H) / 2.);
    giz = c.mz * (float)((double)(giz * (float)G.D) / 2.);
    // T_33^T g
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll 4
    for (int j = 0; j < kJ; ++j) {
      const float* a = Ab + j * 12;
      t0 = fma(w[j], a[0] * gx + a[4] * gy + a[8] * gz, t0);
      t1 = fma(w[j], a[1] * gx + a[5] * gy + a[9] * gz, t1);
      t2 = fma(w[j], a[2] * gx + a[6] * gy + a[10] * gz, t2);
    }
    g_p[3 * i] = t0 + gix * G.sx;
    g_p[3 * i + 1] = t1 + giy * G.sy;
    g_p[3 * i + 2] = t2 + giz * G.sz;
  }
}

// Parameter side of the skinning VJP, staged for two deterministic reductions over the points:
//   gA[b,j,i,k] = sum_{p in frame b} w_j(p) g_d[p,i] [p;1][k]  = (W^T Q)[j, b*12 + 4i + k]   (MFMA gemm_tn, K = P)
//   gtrans[b,i] = sum_{p in frame b} g_d[p,i]                   = colsum(Gs)[b*3 + i]
// with W [P,24] the sampled blend weights (one more gather of the corner records), Q [P, B*12] = g_d (x) [p;1]
// written into the point's frame block (zeros elsewhere) and Gs [P, B*3] likewise.
__global__ __launch_bounds__(kBlk) void lbs_vjp_params_stage_kernel(const float* __restrict__ ps,
                                                                    const int64_t* __restrict__ frame, int B,
                                                                    const float* __restrict__ vol, LbsGeom G, int64_t P,
                                                                    const float* __restrict__ g_d,
                                                                    float* __restrict__ Wout, float* __restrict__ Q,
                                                                    float* __restrict__ Gs) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    const float px = ps[3 * i], py = ps[3 * i + 1], pz = ps[3 * i + 2];
    const float g3[3] = {g_d[3 * i], g_d[3 * i + 1], g_d[3 * i + 2]};
    const int b = (int)frame[i];
    const Cell<float> c = make_cell<float>((px - G.cx) * G.sx, (py - G.cy) * G.sy, (pz - G.cz) * G.sz, G.W, G.H, G.D);
    float w[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) w[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      if (c.in_x[bx] && c.in_y[by] && c.in_z[bz]) {
        const float wk = c.fx[bx] * c.fy[by] * c.fz[bz];
        float v[kJ];
        load24(vol + (((int64_t)(c.z0 + bz) * G.H + (c.y0 + by)) * G.W + (c.x0 + bx)) * kJ, v);
#pragma unroll
        for (int j = 0; j < kJ; ++j) w[j] = fma(v[j], wk, w[j]);
      }
    }
    float4* wo = reinterpret_cast<float4*>(Wout + i * kJ);
#pragma unroll
    for (int q = 0; q < 6; ++q) wo[q] = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    const float ph[4] = {px, py, pz, 1.f};
    float* q = Q + i * (int64_t)(B * 12);
    float* gs = Gs + i * (int64_t)(B * 3);
    for (int bb = 0; bb < B; ++bb) {
      const float on = bb == b ? 1.f : 0.f;
#pragma unroll
      for (int r = 0; r < 12; ++r) q[bb * 12 + r] = on * g3[r >> 2] * ph[r & 3];
#pragma unroll
      for (int r = 0; r < 3; ++r) gs[bb * 3 + r] = on * g3[r];
    }
  }
}

// ---- the skinning stage as a JET: value and d v / d p in one pass, first-order reverse in one pass ----------------------------
// The render loss needs v(p) = T(p) [p;1] + trans and J(p) = dv/dp at the converged ray points and differentiates both once
// (utils/utils.py:133-156 builds J with three create_graph autograd.grad calls through sampler -> blend -> transform, ~50 launches,
// and the loss's backward walks the double-backward graph of all of it, ~150 more).  With w_j(p) the trilinear blend weights,
// y_j = A_j [p;1]:      v = sum_j w_j y_j + trans        J[i][k] = sum_j w_j A_j[i][k] + sum_j (dw_j/dp_k) y_j[i]
// and the reverse of (gv, gJ) needs the mixed second derivatives d2w_j / dp_k dp_m (k != m; the trilinear weight is linear in each
// coordinate) — all of it per point from the same 8 corner records.  dw/dp_k = (sum over corners of +-f f vol) * clip mask * size/2 *
// scale, as the sampler's backward forms it (GridSamplerMineKernel.cu:534-545).
struct JetW {
  float w[kJ];         // w_j
  float d[3][kJ];      // dw_j / dp_k
};

__device__ __forceinline__ void jet_scales(const Cell<float>& c, const LbsGeom& G, float* m) {
  m[0] = c.mx * (float)((double)((float)G.W) / 2.) * G.sx;
  m[1] = c.my * (float)((double)((float)G.H) / 2.) * G.sy;
  m[2] = c.mz * (float)((double)((float)G.D) / 2.) * G.sz;
}

__global__ __launch_bounds__(kBlk) void lbs_jet_forward_kernel(
    const float* __restrict__ ps, const int64_t* __restrict__ frame, const float* __restrict__ A,
    const float* __restrict__ trans, int B, const float* __restrict__ vol, LbsGeom G, int64_t P,
    float* __restrict__ v_out, float* __restrict__ J_out) {
  extern __shared__ __attribute__((aligned(16))) float A_s[];
  stage_A(A, B, A_s);
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    const float q[3] = {ps[3 * i], ps[3 * i + 1], ps[3 * i + 2]};
    const int b = (int)frame[i];
    const Cell<float> c = make_cell<float>((q[0] - G.cx) * G.sx, (q[1] - G.cy) * G.sy, (q[2] - G.cz) * G.sz, G.W, G.H, G.D);
    JetW a;
#pragma unroll
    for (int j = 0; j < kJ; ++j) a.w[j] = a.d[0][j] = a.d[1][j] = a.d[2][j] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      if (c.in_x[bx] && c.in_y[by] && c.in_z[bz]) {
        float v[kJ];
        load24(vol + (((int64_t)(c.z0 + bz) * G.H + (c.y0 + by)) * G.W + (c.x0 + bx)) * kJ, v);
        const float wk = c.fx[bx] * c.fy[by] * c.fz[bz];
        const float tx = bx ? c.fy[by] * c.fz[bz] : -(c.fy[by] * c.fz[bz]);
        const float ty = by ? c.fx[bx] * c.fz[bz] : -(c.fx[bx] * c.fz[bz]);
        const float tz = bz ? c.fx[bx] * c.fy[by] : -(c.fx[bx] * c.fy[by]);
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
          a.w[j] = fma(v[j], wk, a.w[j]);
          a.d[0][j] = fma(v[j], tx, a.d[0][j]);
          a.d[1][j] = fma(v[j], ty, a.d[1][j]);
          a.d[2][j] = fma(v[j], tz, a.d[2][j]);
        }
      }
    }
    float m[3];
    jet_scales(c, G, m);
    float T[12], S[9];                 // T = sum_j w_j A_j (rows 0..2), S[i][k] = sum_j y_j[i] * (raw dw_j/dk)
#pragma unroll
    for (int r = 0; r < 12; ++r) T[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 9; ++r) S[r] = 0.f;
    const float* Ab = A_s + b * kJ * 12;
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
      const float* aj = Ab + j * 12;
#pragma unroll
      for (int r = 0; r < 12; ++r) T[r] = fma(a.w[j], aj[r], T[r]);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float y = aj[4 * r] * q[0] + aj[4 * r + 1] * q[1] + aj[4 * r + 2] * q[2] + aj[4 * r + 3];
#pragma unroll
        for (int k = 0; k < 3; ++k) S[3 * r + k] = fma(a.d[k][j], y, S[3 * r + k]);
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      v_out[3 * i + r] = T[4 * r] * q[0] + T[4 * r + 1] * q[1] + T[4 * r + 2] * q[2] + T[4 * r + 3] + trans[3 * b + r];
#pragma unroll
      for (int k = 0; k < 3; ++k) J_out[9 * i + 3 * r + k] = T[4 * r + k] + S[3 * r + k] * m[k];
    }
  }
}

// Reverse of the jet: cotangents gv [P,3] of v and gJ [P,9] of J (either may be NULL = zeros) ->
//   g_p [P,3]                                                      (per point, written here)
//   gA[b,j,i,k], gtrans[b,i]                                       (sums over the points of a frame, staged for two fixed-order reductions)
// With D_jk = dw_j/dp_k:   dL/dA_j[i][k] = (gv_i w_j + sum_k' gJ[i][k'] D_jk') [p;1][k] + (k < 3) gJ[i][k] w_j
// is a sum of four separable terms, so  gA[j, b*12 + 4i + k] = recmv_gemm_tn(W4, Q4)  with four rows per point:
//   W4[4p + 0] = w,        Q4[4p + 0][b-block] = gv_i [p;1][k] + (k < 3) gJ[i][k]
//   W4[4p + 1 + k'] = D_k', Q4[4p + 1 + k'][b-block] = gJ[i][k'] [p;1][k]
// and gtrans = recmv_colsum(Gs), Gs [P, B*3] = gv in the point's frame block.
__global__ __launch_bounds__(kBlk) void lbs_jet_backward_kernel(
    const float* __restrict__ ps, const int64_t* __restrict__ frame, const float* __restrict__ A, int B,
    const float* __restrict__ vol, LbsGeom G, int64_t P, const float* __restrict__ gv_in,
    const float* __restrict__ gJ_in, float* __restrict__ g_p, float* __restrict__ W4, float* __restrict__ Q4,
    float* __restrict__ Gs) {
  extern __shared__ __attribute__((aligned(16))) float A_s[];
  stage_A(A, B, A_s);
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    const float q[3] = {ps[3 * i], ps[3 * i + 1], ps[3 * i + 2]};
    const int b = (int)frame[i];
    float gv[3], gJ[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) gv[r] = gv_in ? gv_in[3 * i + r] : 0.f;
#pragma unroll
    for (int r = 0; r < 9; ++r) gJ[r] = gJ_in ? gJ_in[9 * i + r] : 0.f;
    const Cell<float> c = make_cell<float>((q[0] - G.cx) * G.sx, (q[1] - G.cy) * G.sy, (q[2] - G.cz) * G.sz, G.W, G.H, G.D);
    JetW a;
    float h[3][kJ];                    // raw mixed second derivatives: h[0] = xy, h[1] = xz, h[2] = yz
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
      a.w[j] = a.d[0][j] = a.d[1][j] = a.d[2][j] = 0.f;
      h[0][j] = h[1][j] = h[2][j] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      if (c.in_x[bx] && c.in_y[by] && c.in_z[bz]) {
        float v[kJ];
        load24(vol + (((int64_t)(c.z0 + bz) * G.H + (c.y0 + by)) * G.W + (c.x0 + bx)) * kJ, v);
        const float wk = c.fx[bx] * c.fy[by] * c.fz[bz];
        const float tx = bx ? c.fy[by] * c.fz[bz] : -(c.fy[by] * c.fz[bz]);
        const float ty = by ? c.fx[bx] * c.fz[bz] : -(c.fx[bx] * c.fz[bz]);
        const float tz = bz ? c.fx[bx] * c.fy[by] : -(c.fx[bx] * c.fy[by]);
        const float hxy = (bx == by) ? c.fz[bz] : -c.fz[bz];
        const float hxz = (bx == bz) ? c.fy[by] : -c.fy[by];
        const float hyz = (by == bz) ? c.fx[bx] : -c.fx[bx];
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
          a.w[j] = fma(v[j], wk, a.w[j]);
          a.d[0][j] = fma(v[j], tx, a.d[0][j]);
          a.d[1][j] = fma(v[j], ty, a.d[1][j]);
          a.d[2][j] = fma(v[j], tz, a.d[2][j]);
          h[0][j] = fma(v[j], hxy, h[0][j]);
          h[1][j] = fma(v[j], hxz, h[1][j]);
          h[2][j] = fma(v[j], hyz, h[2][j]);
        }
      }
    }
    float m[3];
    jet_scales(c, G, m);
    const float mh[3] = {m[0] * m[1], m[0] * m[2], m[1] * m[2]};
    float gq[3] = {0.f, 0.f, 0.f};
    const float* Ab = A_s + b * kJ * 12;
    float* w4 = W4 + i * (int64_t)(4 * kJ);
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
      const float* aj = Ab + j * 12;
      const float D[3] = {a.d[0][j] * m[0], a.d[1][j] * m[1], a.d[2][j] * m[2]};
      const float Hxy = h[0][j] * mh[0], Hxz = h[1][j] * mh[1], Hyz = h[2][j] * mh[2];
      w4[j] = a.w[j];
      w4[kJ + j] = D[0];
      w4[2 * kJ + j] = D[1];
      w4[3 * kJ + j] = D[2];
      float cgy[3] = {0.f, 0.f, 0.f};       // c_k = sum_i gJ[i][k] y_i
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float y = aj[4 * r] * q[0] + aj[4 * r + 1] * q[1] + aj[4 * r + 2] * q[2] + aj[4 * r + 3];
        // sum_i gv_i J[i][m]  with  J[i][m] = w_j A_j[i][m] + D_jm y_i  (this joint's share)
        // sum_ik gJ[i][k] (D_jm A_j[i][k] + D_jk A_j[i][m])
        const float gja = gJ[3 * r] * aj[4 * r] + gJ[3 * r + 1] * aj[4 * r + 1] + gJ[3 * r + 2] * aj[4 * r + 2];
        const float gjd = gJ[3 * r] * D[0] + gJ[3 * r + 1] * D[1] + gJ[3 * r + 2] * D[2];
#pragma unroll
        for (int mm = 0; mm < 3; ++mm)
          gq[mm] += gv[r] * (a.w[j] * aj[4 * r + mm] + D[mm] * y) + D[mm] * gja + gjd * aj[4 * r + mm];
#pragma unroll
        for (int k = 0; k < 3; ++k) cgy[k] = fma(gJ[3 * r + k], y, cgy[k]);
      }
      // sum_ik gJ[i][k] H_j,km y_i = sum_{k != m} H_km c_k
      gq[0] += Hxy * cgy[1] + Hxz * cgy[2];
      gq[1] += Hxy * cgy[0] + Hyz * cgy[2];
      gq[2] += Hxz * cgy[0] + Hyz * cgy[1];
    }
    g_p[3 * i] = gq[0];
    g_p[3 * i + 1] = gq[1];
    g_p[3 * i + 2] = gq[2];
    const float ph[4] = {q[0], q[1], q[2], 1.f};
    float* q4 = Q4 + i * (int64_t)(4 * B * 12);
    float* gs = Gs + i * (int64_t)(B * 3);
    for (int bb = 0; bb < B; ++bb) {
      const float on = bb == b ? 1.f : 0.f;
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        const int ii = r >> 2, k = r & 3;
        q4[bb * 12 + r] = on * (gv[ii] * ph[k] + (k < 3 ? gJ[3 * ii + k] : 0.f));
#pragma unroll
        for (int kp = 0; kp < 3; ++kp) q4[(1 + kp) * (B * 12) + bb * 12 + r] = on * (gJ[3 * ii + kp] * ph[k]);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) gs[bb * 3 + r] = on * gv[r];
    }
  }
}

// One step of utils/FindSurfacePs.py:316-351 on all rays.
__global__ __launch_bounds__(kBlk) void rootfind_update_kernel(float* __restrict__ p, const float* __restrict__ f,
                                                               const float* __restrict__ gf,
                                                               const float* __restrict__ loss2,
                                                               const float* __restrict__ angle,
                                                               const float* __restrict__ gd,
                                                               uint8_t* __restrict__ unfinished,
                                                               int32_t* __restrict__ counter, int64_t P, float dthr,
                                                               float athr, float w1, float w2, int do_update) {
  int local = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    const float fi = f[i];
    const bool done = fabsf(fi) < dthr && angle[i] < athr;
    const bool un = unfinished[i] && !done;
    unfinished[i] = un ? 1 : 0;
    if (un) {
      ++local;
      if (do_update) {
        const float loss = w1 * fabsf(fi) + w2 * loss2[i];
        const float sg = fi > 0.f ? 1.f : (fi < 0.f ? -1.f : 0.f);
        const float g0 = w1 * sg * gf[3 * i] + w2 * gd[3 * i];
        const float g1 = w1 * sg * gf[3 * i + 1] + w2 * gd[3 * i + 1];
        const float g2 = w1 * sg * gf[3 * i + 2] + w2 * gd[3 * i + 2];
        const float t = -loss / (g0 * g0 + g1 * g1 + g2 * g2);
        p[3 * i] += t * g0;
        p[3 * i + 1] += t * g1;
        p[3 * i + 2] += t * g2;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(counter, local);
}

// The same step with its index kept on the device, so that every step of an iteration is the SAME launch (a captured
// step can be replayed): state[0] = index of the step being run; counters[step] receives the number of unfinished rays.
__global__ __launch_bounds__(kBlk) void rootfind_step_kernel(float* __restrict__ p, const float* __restrict__ f,
                                                             const float* __restrict__ gf,
                                                             const float* __restrict__ loss2,
                                                             const float* __restrict__ angle,
                                                             const float* __restrict__ gd,
                                                             uint8_t* __restrict__ unfinished,
                                                             int32_t* __restrict__ counters,
                                                             const int32_t* __restrict__ state, int64_t P, float dthr,
                                                             float athr, float w1, float w2, int times) {
  const int step = state[0];
  const bool do_update = step < times;
  int local = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    const float fi = f[i];
    const bool done = fabsf(fi) < dthr && angle[i] < athr;
    const bool un = unfinished[i] && !done;
    unfinished[i] = un ? 1 : 0;
    if (un) {
      ++local;
      if (do_update) {
        const float loss = w1 * fabsf(fi) + w2 * loss2[i];
        const float sg = fi > 0.f ? 1.f : (fi < 0.f ? -1.f : 0.f);
        const float g0 = w1 * sg * gf[3 * i] + w2 * gd[3 * i];
        const float g1 = w1 * sg * gf[3 * i + 1] + w2 * gd[3 * i + 1];
        const float g2 = w1 * sg * gf[3 * i + 2] + w2 * gd[3 * i + 2];
        const float t = -loss / (g0 * g0 + g1 * g1 + g2 * g2);
        p[3 * i] += t * g0;
        p[3 * i + 1] += t * g1;
        p[3 * i + 2] += t * g2;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(counters + step, local);
}

// marks[step] = counters[step] + 1 (0 = "step not run yet" for a host that polls a copy of marks); state[0] = step + 1
__global__ void rootfind_advance_kernel(const int32_t* __restrict__ counters, int32_t* __restrict__ marks,
                                        int32_t* __restrict__ state) {
  const int step = state[0];
  marks[step] = counters[step] + 1;
  state[0] = step + 1;
}

int check_geom(const recmv_lbs_grid* g) {
  RECMV_REQUIRE(g && g->volume, "lbs: NULL skinning grid");
  RECMV_REQUIRE(g->D > 0 && g->H > 0 && g->W > 0, "lbs: empty skinning grid");
  RECMV_REQUIRE((reinterpret_cast<uintptr_t>(g->volume) & 15) == 0, "lbs: skinning grid must be 16-byte aligned");
  return RECMV_OK;
}

LbsGeom to_geom(const recmv_lbs_grid* g) {
  LbsGeom G;
  G.D = g->D;
  G.H = g->H;
  G.W = g->W;
  G.cx = g->center[0];
  G.cy = g->center[1];
  G.cz = g->center[2];
  G.sx = g->scale[0];
  G.sy = g->scale[1];
  G.sz = g->scale[2];
  return G;
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int recmv_lbs_forward(const float* ps, const int64_t* frame, int64_t P, const float* A,
                                 const float* trans, int64_t B, const recmv_lbs_grid* grid, const float* cam,
                                 const float* rays, float* d, float* loss2, float* angle, float* g_d, void* stream) {
  RECMV_REQUIRE(P >= 0 && B >= 1 && B <= 128, "lbs_forward: bad size (P=%lld, B=%lld)", (long long)P, (long long)B);
  if (P == 0) return RECMV_OK;
  int rc = check_geom(grid);
  if (rc) return rc;
  RECMV_REQUIRE(ps && frame && A && trans && d, "lbs_forward: NULL pointer");
  RECMV_REQUIRE(!rays || (cam && loss2 && angle && g_d), "lbs_forward: ray outputs missing");
  const int lds = (int)(B * kJ * 12 * sizeof(float));
  hipLaunchKernelGGL(lbs_forward_kernel, dim3(stream_grid(P, kBlk)), dim3(kBlk), lds, (hipStream_t)stream, ps, frame,
                     A, trans, (int)B, grid->volume, to_geom(grid), P, cam, rays, d, loss2, angle, g_d);
  return check_launch("lbs_forward");
}

extern "C" int recmv_lbs_vjp_input(const float* ps, const int64_t* frame, int64_t P, const float* A, int64_t B,
                                   const recmv_lbs_grid* grid, const float* g_d, float* g_p, void* stream) {
  RECMV_REQUIRE(P >= 0 && B >= 1 && B <= 128, "lbs_vjp_input: bad size (P=%lld, B=%lld)", (long long)P, (long long)B);
  if (P == 0) return RECMV_OK;
  int rc = check_geom(grid);
  if (rc) return rc;
  RECMV_REQUIRE(ps && frame && A && g_d && g_p, "lbs_vjp_input: NULL pointer");
  const int lds = (int)(B * kJ * 12 * sizeof(float));
  hipLaunchKernelGGL(lbs_vjp_kernel, dim3(stream_grid(P, kBlk)), dim3(kBlk), lds, (hipStream_t)stream, ps, frame, A,
                     (int)B, grid->volume, to_geom(grid), P, g_d, g_p);
  return check_launch("lbs_vjp_input");
}

// Stage the parameter side of the skinning VJP: W [P,24], Q [P, B*12], Gs [P, B*3] (see the kernel); the caller then
// runs recmv_gemm_tn(W, Q) -> [24, B*12] and recmv_colsum(Gs) -> [B*3], both with a fixed summation order.
extern "C" int recmv_lbs_vjp_params_stage(const float* ps, const int64_t* frame, int64_t P, int64_t B,
                                          const recmv_lbs_grid* grid, const float* g_d, float* W, float* Q, float* Gs,
                                          void* stream) {
  RECMV_REQUIRE(P >= 0 && B >= 1 && B <= 128, "lbs_vjp_params_stage: bad size (P=%lld, B=%lld)", (long long)P,
                (long long)B);
  if (P == 0) return RECMV_OK;
  int rc = check_geom(grid);
  if (rc) return rc;
  RECMV_REQUIRE(ps && frame && g_d && W && Q && Gs, "lbs_vjp_params_stage: NULL pointer");
  RECMV_REQUIRE((reinterpret_cast<uintptr_t>(W) & 15) == 0, "lbs_vjp_params_stage: W must be 16-byte aligned");
  hipLaunchKernelGGL(lbs_vjp_params_stage_kernel, dim3(stream_grid(P, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, ps,
                     frame, (int)B, grid->volume, to_geom(grid), P, g_d, W, Q, Gs);
  return check_launch("lbs_vjp_params_stage");
}

// Skinning jet (value + d v / d p) and its first-order reverse; see the kernels.  v [P,3], J [P,9] (row i = grad of v_i).
extern "C" int recmv_lbs_jet_forward(const float* ps, const int64_t* frame, int64_t P, const float* A,
                                     const float* trans, int64_t B, const recmv_lbs_grid* grid, float* v, float* J,
                                     void* stream) {
  RECMV_REQUIRE(P >= 0 && B >= 1 && B <= 128, "lbs_jet_forward: bad size (P=%lld, B=%lld)", (long long)P, (long long)B);
  if (P == 0) return RECMV_OK;
  int rc = check_geom(grid);
  if (rc) return rc;
  RECMV_REQUIRE(ps && frame && A && trans && v && J, "lbs_jet_forward: NULL pointer");
  const int lds = (int)(B * kJ * 12 * sizeof(float));
  hipLaunchKernelGGL(lbs_jet_forward_kernel, dim3(stream_grid(P, kBlk)), dim3(kBlk), lds, (hipStream_t)stream, ps, frame,
                     A, trans, (int)B, grid->volume, to_geom(grid), P, v, J);
  return check_launch("lbs_jet_forward");
}

// g_p [P,3] and the staged parameter side: W4 [4P,24], Q4 [4P, B*12], Gs [P, B*3]; the caller runs
// recmv_gemm_tn(W4, Q4) -> gA[j, b*12 + 4i + k] and recmv_colsum(Gs) -> gtrans[b*3 + i].  gv or gJ may be NULL (zeros).
extern "C" int recmv_lbs_jet_backward_stage(const float* ps, const int64_t* frame, int64_t P, const float* A, int64_t B,
                                            const recmv_lbs_grid* grid, const float* gv, const float* gJ, float* g_p,
                                            float* W4, float* Q4, float* Gs, void* stream) {
  RECMV_REQUIRE(P >= 0 && B >= 1 && B <= 128, "lbs_jet_backward_stage: bad size (P=%lld, B=%lld)", (long long)P,
                (long long)B);
  if (P == 0) return RECMV_OK;
  int rc = check_geom(grid);
  if (rc) return rc;
  RECMV_REQUIRE(ps && frame && A && g_p && W4 && Q4 && Gs, "lbs_jet_backward_stage: NULL pointer");
  const int lds = (int)(B * kJ * 12 * sizeof(float));
  hipLaunchKernelGGL(lbs_jet_backward_kernel, dim3(stream_grid(P, kBlk)), dim3(kBlk), lds, (hipStream_t)stream, ps,
                     frame, A, (int)B, grid->volume, to_geom(grid), P, gv, gJ, g_p, W4, Q4, Gs);
  return check_launch("lbs_jet_backward_stage");
}

extern "C" int recmv_rootfind_step(float* p, const float* f, const float* gf, const float* loss2, const float* angle,
                                   const float* gd, uint8_t* unfinished, int32_t* counters, int32_t* marks,
                                   int32_t* state, int64_t P, float dthreshold, float athreshold, float w1, float w2,
                                   int times, void* stream) {
  RECMV_REQUIRE(P > 0 && times >= 0, "rootfind_step: bad sizes");
  RECMV_REQUIRE(p && f && gf && loss2 && angle && gd && unfinished && counters && marks && state,
                "rootfind_step: NULL pointer");
  hipLaunchKernelGGL(rootfind_step_kernel, dim3(stream_grid(P, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, p, f, gf,
                     loss2, angle, gd, unfinished, counters, state, P, dthreshold, athreshold, w1, w2, times);
  hipLaunchKernelGGL(rootfind_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counters, marks, state);
  return check_launch("rootfind_step");
}

extern "C" int recmv_rootfind_update(float* p, const float* f, const float* gf, const float* loss2,
                                     const float* angle, const float* gd, uint8_t* unfinished, int32_t* counter,
                                     int64_t P, float dthreshold, float athreshold, float w1, float w2,
                                     int do_update, void* stream) {
  RECMV_REQUIRE(P >= 0, "rootfind_update: negative P");
  if (P == 0) return RECMV_OK;
  RECMV_REQUIRE(p && f && gf && loss2 && angle && gd && unfinished && counter, "rootfind_update: NULL pointer");
  hipLaunchKernelGGL(rootfind_update_kernel, dim3(stream_grid(P, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, p, f, gf,
                     loss2, angle, gd, unfinished, counter, P, dthreshold, athreshold, w1, w2, do_update);
  return check_launch("rootfind_update");
}
