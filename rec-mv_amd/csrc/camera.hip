// Camera of the hot path as three kernels: world points -> NDC / screen / pixel coordinates (forward + backward) and pixel ->
// unit world ray (forward + backward).  Replaces the element-wise torch chains of model/CameraMine.py:62-88, 104-142, 146-169
// (RectifiedPerspectiveCameras.transform_points / transform_points_screen / view_rays / project): ~385 ATen launches per optimiser
// iteration (round-6 census) become ~12.  HBM-bound and tiny: 24 B per point forward; the backward adds a two-stage, fixed-order
// reduction of the seven camera gradients (translation, focal length, principal point) — no float atomics, bit-reproducible.
//
// The forward arithmetic keeps the operation ORDER of the torch expressions it replaces (no fma contraction), so that a
// rasterised silhouette or a ray does not move against the previous build.
#include "common.h"

#pragma clang fp contract(off)

namespace recmv {
namespace {

constexpr int kBlk = 256;

struct Cam {
  float R[9], T[3], f[2], pp[2];
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ c) {
  Cam k;
#pragma unroll
  for (int i = 0; i < 9; ++i) k.R[i] = c[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) k.T[i] = c[9 + i];
  k.f[0] = c[12], k.f[1] = c[13], k.pp[0] = c[14], k.pp[1] = c[15];
  return k;
}

// v_j = sum_i p_i R[i][j] + T_j        ((ps.unsqueeze(-1) * R).sum(-2) + T)
__device__ __forceinline__ void to_view(const Cam& k, const float p[3], float v[3]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) v[j] = ((p[0] * k.R[j] + p[1] * k.R[3 + j]) + p[2] * k.R[6 + j]) + k.T[j];
}

// mode 0: (x_ndc, y_ndc, z_view)   1: (screen_x, screen_y, 1 / z_view)   2: pixel (x, y) of project()
__global__ void __launch_bounds__(kBlk) cam_project_kernel(const float* __restrict__ pts, int64_t P, const float* __restrict__ cam,
                                                           float W, float H, float cx, float cy, int mode,
                                                           float* __restrict__ out) {
  const Cam k = load_cam(cam);
  const int od = mode == 2 ? 2 : 3;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    const float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    float v[3];
    to_view(k, p, v);
    const float z = v[2];
    if (mode == 2) {
      out[2 * i] = k.pp[0] - v[0] * k.f[0] / z;
      out[2 * i + 1] = k.pp[1] - v[1] * k.f[1] / z;
      continue;
    }
    const float fx = k.f[0] / (W / 2.f), fy = k.f[1] / (H / 2.f);
    const float px = cx - k.pp[0] / (W / 2.f), py = cy - k.pp[1] / (H / 2.f);        // cx = 1 - 1/W, cy = 1 - 1/H (host doubles)
    const float xn = (fx * v[0] + px * z) / z, yn = (fy * v[1] + py * z) / z;
    if (mode == 0) {
      out[od * i] = xn, out[od * i + 1] = yn, out[od * i + 2] = z;
    } else {
      out[od * i] = (W - 1.f) / 2.f - W * xn / 2.f;
      out[od * i + 1] = (H - 1.f) / 2.f - H * yn / 2.f;
      out[od * i + 2] = 1.0f / z;
    }
  }
}

// block-wide sum of NV per-thread values -> partial[blockIdx][NV]; fixed order (lane tree, then waves in order)
template <int NV>
__device__ __forceinline__ void block_partial(float (&acc)[NV], float* __restrict__ partial) {
  __shared__ float red[kBlk / kWave][NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    float a = acc[e];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) a += __shfl_down(a, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave][e] = a;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kBlk / kWave; ++w) s += red[w][threadIdx.x];
    partial[(int64_t)blockIdx.x * NV + threadIdx.x] = s;
  }
}

// g_pts [P,3] (optional) and per-block partial sums of (gT[3], gf[2], gpp[2])
__global__ void __launch_bounds__(kBlk) cam_project_bwd_kernel(const float* __restrict__ pts, const float* __restrict__ g_out, int64_t P,
                                                               const float* __restrict__ cam, float W, float H, int mode,
                                                               float* __restrict__ g_pts, float* __restrict__ partial) {
  const Cam k = load_cam(cam);
  const int od = mode == 2 ? 2 : 3;
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    const float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    float v[3];
    to_view(k, p, v);
    const float z = v[2], iz = 1.f / z;
    float gv[3];
    if (mode == 2) {
      const float g0 = g_out[2 * i], g1 = g_out[2 * i + 1];
      gv[0] = -g0 * k.f[0] * iz;
      gv[1] = -g1 * k.f[1] * iz;
      gv[2] = (g0 * v[0] * k.f[0] + g1 * v[1] * k.f[1]) * iz * iz;
      acc[3] += -g0 * v[0] * iz, acc[4] += -g1 * v[1] * iz;
      acc[5] += g0, acc[6] += g1;
    } else {
      float g0 = g_out[od * i], g1 = g_out[od * i + 1], g2 = g_out[od * i + 2];
      if (mode == 1) {                                     // screen = (S-1)/2 - S ndc / 2, third = 1 / z
        g0 *= -0.5f * W, g1 *= -0.5f * H, g2 = -g2 * iz * iz;
      }
      const float fx = k.f[0] / (W / 2.f), fy = k.f[1] / (H / 2.f);
      gv[0] = g0 * fx * iz;
      gv[1] = g1 * fy * iz;
      gv[2] = g2 - (g0 * fx * v[0] + g1 * fy * v[1]) * iz * iz;
      acc[3] += g0 * v[0] * iz / (W / 2.f), acc[4] += g1 * v[1] * iz / (H / 2.f);
      acc[5] += -g0 / (W / 2.f), acc[6] += -g1 / (H / 2.f);
    }
    acc[0] += gv[0], acc[1] += gv[1], acc[2] += gv[2];
    if (g_pts) {
#pragma unroll
      for (int r = 0; r < 3; ++r) g_pts[3 * i + r] = (k.R[3 * r] * gv[0] + k.R[3 * r + 1] * gv[1]) + k.R[3 * r + 2] * gv[2];
    }
  }
  block_partial<7>(acc, partial);
}

// out[e] = sum over blocks of partial[b][e], blocks in order, accumulated in double
__global__ void cam_reduce_kernel(const float* __restrict__ partial, int nblocks, int nv, float* __restrict__ out) {
  const int e = threadIdx.x;
  if (e >= nv) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += (double)partial[(int64_t)b * nv + e];
  out[e] = (float)s;
}

// pixel (x, y, w) -> unit world ray.  pix_f: [P,3] float, or (col, row): int64 with w = 1
__global__ void __launch_bounds__(kBlk) cam_rays_kernel(const float* __restrict__ pix_f, const int64_t* __restrict__ col,
                                                        const int64_t* __restrict__ row, int64_t P, const float* __restrict__ cam,
                                                        float* __restrict__ out) {
  const Cam k = load_cam(cam);
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    float x, y, w;
    if (pix_f) x = pix_f[3 * i], y = pix_f[3 * i + 1], w = pix_f[3 * i + 2];
    else x = (float)col[i], y = (float)row[i], w = 1.f;
    const float r0 = -x / k.f[0] + w * k.pp[0] / k.f[0];
    const float r1 = -y / k.f[1] + w * k.pp[1] / k.f[1];
    const float n = sqrtf((r0 * r0 + r1 * r1) + w * w);
    const float u[3] = {r0 / n, r1 / n, w / n};
#pragma unroll
    for (int j = 0; j < 3; ++j) out[3 * i + j] = (u[0] * k.R[3 * j] + u[1] * k.R[3 * j + 1]) + u[2] * k.R[3 * j + 2];
  }
}

// partial sums of (gf[2], gpp[2]) for the rays
__global__ void __launch_bounds__(kBlk) cam_rays_bwd_kernel(const float* __restrict__ pix_f, const int64_t* __restrict__ col,
                                                            const int64_t* __restrict__ row, const float* __restrict__ g_out, int64_t P,
                                                            const float* __restrict__ cam, float* __restrict__ partial) {
  const Cam k = load_cam(cam);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < P; i += (int64_t)gridDim.x * kBlk) {
    float x, y, w;
    if (pix_f) x = pix_f[3 * i], y = pix_f[3 * i + 1], w = pix_f[3 * i + 2];
    else x = (float)col[i], y = (float)row[i], w = 1.f;
    const float r0 = -x / k.f[0] + w * k.pp[0] / k.f[0];
    const float r1 = -y / k.f[1] + w * k.pp[1] / k.f[1];
    const float n = sqrtf((r0 * r0 + r1 * r1) + w * w);
    const float u[3] = {r0 / n, r1 / n, w / n};
    float gu[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) gu[c] = (g_out[3 * i] * k.R[c] + g_out[3 * i + 1] * k.R[3 + c]) + g_out[3 * i + 2] * k.R[6 + c];
    const float dot = u[0] * gu[0] + u[1] * gu[1] + u[2] * gu[2];
    const float gr0 = (gu[0] - u[0] * dot) / n, gr1 = (gu[1] - u[1] * dot) / n;
    acc[0] += gr0 * (-r0 / k.f[0]), acc[1] += gr1 * (-r1 / k.f[1]);
    acc[2] += gr0 * w / k.f[0], acc[3] += gr1 * w / k.f[1];
  }
  block_partial<4>(acc, partial);
}

inline int bwd_blocks(int64_t P) {
  int64_t g = ceil_div(P, (int64_t)kBlk * 4);
  if (g > 1024) g = 1024;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int64_t recmv_cam_partial_floats(int64_t P) { return (int64_t)bwd_blocks(P) * 7; }

extern "C" int recmv_cam_project(const float* pts, int64_t P, const float* cam16, float W, float H, int mode, float* out, void* stream) {
  RECMV_REQUIRE(P >= 0, "cam_project: negative P");
  RECMV_REQUIRE(mode >= 0 && mode <= 2, "cam_project: mode must be 0 (ndc), 1 (screen) or 2 (pixel), got %d", mode);
  RECMV_REQUIRE(W > 0.f && H > 0.f, "cam_project: image size must be positive");
  if (P == 0) return RECMV_OK;
  RECMV_REQUIRE(pts && cam16 && out, "cam_project: NULL pointer");
  const float cx = (float)(1.0 - 1.0 / (double)W), cy = (float)(1.0 - 1.0 / (double)H);
  hipLaunchKernelGGL(cam_project_kernel, dim3((unsigned)stream_grid(P, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, pts, P, cam16, W, H,
                     cx, cy, mode, out);
  return check_launch("cam_project");
}

extern "C" int recmv_cam_project_backward(const float* pts, const float* g_out, int64_t P, const float* cam16, float W, float H, int mode,
                                          float* g_pts, float* g_cam7, float* partial, int64_t partial_floats, void* stream) {
  RECMV_REQUIRE(P >= 0, "cam_project_backward: negative P");
  RECMV_REQUIRE(mode >= 0 && mode <= 2, "cam_project_backward: bad mode %d", mode);
  RECMV_REQUIRE(g_cam7, "cam_project_backward: NULL g_cam7");
  if (P == 0) {
    RECMV_HIP_TRY(hipMemsetAsync(g_cam7, 0, 7 * sizeof(float), (hipStream_t)stream));
    return RECMV_OK;
  }
  RECMV_REQUIRE(pts && g_out && cam16 && partial, "cam_project_backward: NULL pointer");
  const int nb = bwd_blocks(P);
  RECMV_REQUIRE(partial_floats >= (int64_t)nb * 7, "cam_project_backward: partial buffer too small (%lld < %lld floats)",
                (long long)partial_floats, (long long)nb * 7);
  hipLaunchKernelGGL(cam_project_bwd_kernel, dim3(nb), dim3(kBlk), 0, (hipStream_t)stream, pts, g_out, P, cam16, W, H, mode, g_pts,
                     partial);
  hipLaunchKernelGGL(cam_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nb, 7, g_cam7);
  return check_launch("cam_project_backward");
}

extern "C" int recmv_cam_rays(const float* pix, const int64_t* col, const int64_t* row, int64_t P, const float* cam16, float* out,
                              void* stream) {
  RECMV_REQUIRE(P >= 0, "cam_rays: negative P");
  if (P == 0) return RECMV_OK;
  RECMV_REQUIRE(cam16 && out && (pix || (col && row)), "cam_rays: NULL pointer");
  hipLaunchKernelGGL(cam_rays_kernel, dim3((unsigned)stream_grid(P, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, pix, col, row, P, cam16,
                     out);
  return check_launch("cam_rays");
}

extern "C" int recmv_cam_rays_backward(const float* pix, const int64_t* col, const int64_t* row, const float* g_out, int64_t P,
                                       const float* cam16, float* g_cam4, float* partial, int64_t partial_floats, void* stream) {
  RECMV_REQUIRE(P >= 0, "cam_rays_backward: negative P");
  RECMV_REQUIRE(g_cam4, "cam_rays_backward: NULL g_cam4");
  if (P == 0) {
    RECMV_HIP_TRY(hipMemsetAsync(g_cam4, 0, 4 * sizeof(float), (hipStream_t)stream));
    return RECMV_OK;
  }
  RECMV_REQUIRE(cam16 && g_out && partial && (pix || (col && row)), "cam_rays_backward: NULL pointer");
  const int nb = bwd_blocks(P);
  RECMV_REQUIRE(partial_floats >= (int64_t)nb * 4, "cam_rays_backward: partial buffer too small");
  hipLaunchKernelGGL(cam_rays_bwd_kernel, dim3(nb), dim3(kBlk), 0, (hipStream_t)stream, pix, col, row, g_out, P, cam16, partial);
  hipLaunchKernelGGL(cam_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nb, 4, g_cam4);
  return check_launch("cam_rays_backward");
}
