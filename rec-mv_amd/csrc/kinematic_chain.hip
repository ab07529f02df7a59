// SMPL kinematic chain of the LBS deformer, forward and backward, one kernel each — gfx950.
//
// Semantics follow LBSkinner.forward / posedSkeleton (model/Deformer.py:372-405, 311-334 of the reference):
//   R_j   = batch_rodrigues(pose_j)                    axis-angle -> quaternion -> matrix (see below)
//   G_0   = [R_0 | J_0],  G_j = G_parent(j) . [R_j | J_j - J_parent(j)]          (results, :384-396)
//   A_j   = G_j . init_pose_j                                                     (:405)
// `batch_rodrigues` is un-vendored in the reference (smpl_pytorch); this is the standard HMR form
// (angle = ||theta + 1e-8||, q = (cos(a/2), sin(a/2) theta/a), normalise, quat -> matrix;
// utils/utils.py:21-39 gives the quat -> matrix part) — parity unpinned, same statement as
// recmv/model/Deformer.py::batch_rodrigues which the tests compare against.
//
// Why a kernel: the reference (and a torch restatement) walks the 24 joints in a Python loop — ~250 tiny
// launches forward and ~500 in autograd backward, for every deformer call (hundreds per optimiser step).
// The whole chain is ~5 kFLOP per frame, so one thread per frame does it in registers; the backward
// re-runs the chain and reverse-accumulates, differentiating the rodrigues map with 3-direction dual
// numbers.  Latency-bound by construction (B = 3..90 frames); its job is to remove launches.
#include "common.h"

// No fma contraction in this file.  The chain restates the reference's separately rounded torch operations ("same operation order"
// below); contracted into fmas, the posed joint transforms differ from the reference's in the last bits — and because the chain
// positions EVERY vertex and ray of a frame, that difference is coherent across all points, not noise.  Round 6 found it to be the
// seed of the optimisation trajectories' divergence from the reference (tools/trajectory_seeds.py, profiles/r06_trajectory_seeds.txt:
// squared canonical Chamfer at 14 iterations 7.6e-6 -> 7.1e-8, at 35 iterations 2.2e-4 -> 1.4e-5, inside the reference's own
// run-to-run envelope; no other kernel's contraction, the hardware exp / log, the sampler's summation order or the regulariser's
// route moves it).  The kernel is latency-bound (one thread per frame): the cost is nil.
#pragma clang fp contract(off)

namespace recmv {
namespace {

constexpr int NJ = 24;

struct Dual {  // value + 3 tangents (d/d theta_x, d/d theta_y, d/d theta_z)
  float v, d[3];
};
__device__ __forceinline__ Dual mk(float v) { return {v, {0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) {
  return {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}};
}
__device__ __forceinline__ Dual operator-(Dual a, Dual b) {
  return {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}};
}
__device__ __forceinline__ Dual operator*(Dual a, Dual b) {
  return {a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ Dual operator*(float s, Dual a) { return {s * a.v, {s * a.d[0], s * a.d[1], s * a.d[2]}}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const float q = a.v / b.v, ib = 1.f / b.v;
  return {q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib}};
}
__device__ __forceinline__ Dual dsqrt(Dual a) {
  const float s = sqrtf(a.v), h = 0.5f / s;
  return {s, {a.d[0] * h, a.d[1] * h, a.d[2] * h}};
}
__device__ __forceinline__ Dual dsin(Dual a) {
  const float s = sinf(a.v), c = cosf(a.v);
  return {s, {c * a.d[0], c * a.d[1], c * a.d[2]}};
}
__device__ __forceinline__ Dual dcos(Dual a) {
  const float s = sinf(a.v), c = cosf(a.v);
  return {c, {-s * a.d[0], -s * a.d[1], -s * a.d[2]}};
}

// scalar rodrigues, same operation order as the torch restatement
template <typename S>
struct Ops;
template <>
struct Ops<float> {
  static __device__ __forceinline__ float c(float v) { return v; }
  static __device__ __forceinline__ float sq(float a) { return sqrtf(a); }
  static __device__ __forceinline__ float sn(float a) { return sinf(a); }
  static __device__ __forceinline__ float cs(float a) { return cosf(a); }
};
template <>
struct Ops<Dual> {
  static __device__ __forceinline__ Dual c(float v) { return mk(v); }
  static __device__ __forceinline__ Dual sq(Dual a) { return dsqrt(a); }
  static __device__ __forceinline__ Dual sn(Dual a) { return dsin(a); }
  static __device__ __forceinline__ Dual cs(Dual a) { return dcos(a); }
};

template <typename S>
__device__ __forceinline__ void rodrigues(S tx, S ty, S tz, S* R /*9*/) {
  using O = Ops<S>;
  const S e = O::c(1e-8f);
  const S ax = tx + e, ay = ty + e, az = tz + e;
  const S angle = O::sq(ax * ax + ay * ay + az * az);          // ||theta + 1e-8||
  const S nx = tx / angle, ny = ty / angle, nz = tz / angle;   // theta / angle (no epsilon)
  const S half = 0.5f * angle;
  const S qw = O::cs(half), sh = O::sn(half);
  S qx = sh * nx, qy = sh * ny, qz = sh * nz;
  const S qn = O::sq(qw * qw + qx * qx + qy * qy + qz * qz);   // quat2mat normalises first
  const S w = qw / qn, x = qx / qn, y = qy / qn, z = qz / qn;
  const S w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
  const S wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
  R[0] = w2 + x2 - y2 - z2;
  R[1] = 2.f * xy - 2.f * wz;
  R[2] = 2.f * wy + 2.f * xz;
  R[3] = 2.f * wz + 2.f * xy;
  R[4] = w2 - x2 + y2 - z2;
  R[5] = 2.f * yz - 2.f * wx;
  R[6] = 2.f * xz - 2.f * wy;
  R[7] = 2.f * wx + 2.f * yz;
  R[8] = w2 - x2 - y2 + z2;
}

// 3x4 affine (last row 0 0 0 1 implicit): C = A . B
__device__ __forceinline__ void aff_mul(const float* A, const float* B, float* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j] + A[i * 4 + 2] * B[2 * 4 + j];
      if (j == 3) s += A[i * 4 + 3];
      C[i * 4 + j] = s;
    }
  }
}

struct ChainConst {
  float Js[NJ * 3];
  int parents[NJ];
};

// forward: poses [B,24,3] -> G [B,24,4,4] (results) and A [B,24,4,4] = G . init_pose
__global__ void chain_fwd_kernel(const float* __restrict__ poses, ChainConst cc,
                                 const float* __restrict__ init_pose /*[24,4,4] or null*/, float* __restrict__ G,
                                 float* __restrict__ A, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float g[NJ][12];
  for (int j = 0; j < NJ; ++j) {
    float R[9];
    const float* th = poses + ((int64_t)b * NJ + j) * 3;
    rodrigues<float>(th[0], th[1], th[2], R);
    float L[12];
    const int p = cc.parents[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      L[i * 4 + 0] = R[i * 3 + 0];
      L[i * 4 + 1] = R[i * 3 + 1];
      L[i * 4 + 2] = R[i * 3 + 2];
      L[i * 4 + 3] = (j == 0) ? cc.Js[i] : cc.Js[j * 3 + i] - cc.Js[p * 3 + i];
    }
    if (j == 0) {
#pragma unroll
      for (int e = 0; e < 12; ++e) g[0][e] = L[e];
    } else {
      aff_mul(g[p], L, g[j]);
    }
  }
  for (int j = 0; j < NJ; ++j) {
    float* go = G + ((int64_t)b * NJ + j) * 16;
#pragma unroll
    for (int e = 0; e < 12; ++e) go[e] = g[j][e];
    go[12] = 0.f; go[13] = 0.f; go[14] = 0.f; go[15] = 1.f;
    if (A) {
      float* ao = A + ((int64_t)b * NJ + j) * 16;
      const float* ip = init_pose + j * 16;
      // general 4x4 product with init_pose (its last row is 0 0 0 1 for a rigid inverse, but stay general)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float gik = (i < 3) ? g[j][i * 4 + k] : (k == 3 ? 1.f : 0.f);
            s += gik * ip[k * 4 + c];
          }
          ao[i * 4 + c] = s;
        }
    }
  }
}

// backward: (gG, gA) -> gposes.  Either gradient pointer may be null.
__global__ void chain_bwd_kernel(const float* __restrict__ poses, ChainConst cc,
                                 const float* __restrict__ init_pose, const float* __restrict__ gG,
                                 const float* __restrict__ gA, float* __restrict__ gposes, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float g[NJ][12];   // forward results (recomputed)
  float L[NJ][12];
  for (int j = 0; j < NJ; ++j) {
    float R[9];
    const float* th = poses + ((int64_t)b * NJ + j) * 3;
    rodrigues<float>(th[0], th[1], th[2], R);
    const int p = cc.parents[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      L[j][i * 4 + 0] = R[i * 3 + 0];
      L[j][i * 4 + 1] = R[i * 3 + 1];
      L[j][i * 4 + 2] = R[i * 3 + 2];
      L[j][i * 4 + 3] = (j == 0) ? cc.Js[i] : cc.Js[j * 3 + i] - cc.Js[p * 3 + i];
    }
    if (j == 0) {
#pragma unroll
      for (int e = 0; e < 12; ++e) g[0][e] = L[0][e];
    } else {
      aff_mul(g[p], L[j], g[j]);
    }
  }
  // gradient wrt the top 3 rows of every G_j
  float dg[NJ][12];
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int e = 0; e < 12; ++e) dg[j][e] = gG ? gG[((int64_t)b * NJ + j) * 16 + e] : 0.f;
    if (gA) {
      const float* ga = gA + ((int64_t)b * NJ + j) * 16;
      const float* ip = init_pose + j * 16;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) s += ga[i * 4 + c] * ip[k * 4 + c];   // gA . init_pose^T
          dg[j][i * 4 + k] += s;
        }
    }
  }
  for (int j = NJ - 1; j >= 0; --j) {
    float gR[9];
    if (j == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) gR[i * 3 + c] = dg[0][i * 4 + c];
    } else {
      const int p = cc.parents[j];
      // G_j = G_p . L_j  (affine):  dG_p[:, :3] += dG_j[:, :3] . R_j^T ... handle the affine column explicitly
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) s += dg[j][i * 4 + c] * L[j][k * 4 + c];
          dg[p][i * 4 + k] += s;
        }
        dg[p][i * 4 + 3] += dg[j][i * 4 + 3];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i) s += g[p][i * 4 + k] * dg[j][i * 4 + c];   // (G_p[:3,:3])^T . dG_j
          gR[k * 3 + c] = s;
        }
    }
    const float* th = poses + ((int64_t)b * NJ + j) * 3;
    Dual Rd[9];
    Dual tx = {th[0], {1.f, 0.f, 0.f}}, ty = {th[1], {0.f, 1.f, 0.f}}, tz = {th[2], {0.f, 0.f, 1.f}};
    rodrigues<Dual>(tx, ty, tz, Rd);
    float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      gp[0] += gR[e] * Rd[e].d[0];
      gp[1] += gR[e] * Rd[e].d[1];
      gp[2] += gR[e] * Rd[e].d[2];
    }
    float* out = gposes + ((int64_t)b * NJ + j) * 3;
    out[0] = gp[0];
    out[1] = gp[1];
    out[2] = gp[2];
  }
}

}  // namespace
}  // namespace recmv

using namespace recmv;

static int fill_const(ChainConst* cc, const float* Js_host, const int32_t* parents_host) {
  RECMV_REQUIRE(Js_host && parents_host, "kinematic_chain: NULL Js/parents");
  for (int i = 0; i < NJ * 3; ++i) cc->Js[i] = Js_host[i];
  for (int j = 0; j < NJ; ++j) {
    cc->parents[j] = parents_host[j];
    RECMV_REQUIRE(j == 0 || (parents_host[j] >= 0 && parents_host[j] < j),
                  "kinematic_chain: parents must precede children (parents[%d]=%d)", j, parents_host[j]);
  }
  return RECMV_OK;
}

extern "C" int recmv_kinematic_chain_forward(const float* poses, const float* Js_host, const int32_t* parents_host,
                                             const float* init_pose, float* G, float* A, int64_t B, void* stream) {
  RECMV_REQUIRE(B >= 0, "kinematic_chain: B < 0");
  if (B == 0) return RECMV_OK;
  RECMV_REQUIRE(poses && G, "kinematic_chain_forward: NULL pointer");
  RECMV_REQUIRE(!A || init_pose, "kinematic_chain_forward: A requested without init_pose");
  ChainConst cc;
  int rc = fill_const(&cc, Js_host, parents_host);
  if (rc) return rc;
  hipLaunchKernelGGL(chain_fwd_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, poses, cc,
                     init_pose, G, A, (int)B);
  return check_launch("kinematic_chain_forward");
}

extern "C" int recmv_kinematic_chain_backward(const float* poses, const float* Js_host, const int32_t* parents_host,
                                              const float* init_pose, const float* gG, const float* gA,
                                              float* gposes, int64_t B, void* stream) {
  RECMV_REQUIRE(B >= 0, "kinematic_chain: B < 0");
  if (B == 0) return RECMV_OK;
  RECMV_REQUIRE(poses && gposes, "kinematic_chain_backward: NULL pointer");
  RECMV_REQUIRE(!gA || init_pose, "kinematic_chain_backward: gA without init_pose");
  ChainConst cc;
  int rc = fill_const(&cc, Js_host, parents_host);
  if (rc) return rc;
  hipLaunchKernelGGL(chain_bwd_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, poses, cc,
                     init_pose, gG, gA, gposes, (int)B);
  return check_launch("kinematic_chain_backward");
}
