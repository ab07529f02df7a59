// Point-cloud rasteriser (K nearest points per pixel) and alpha compositor, forward and backward — gfx950.
//
// What it computes: the silhouette renderer of the mask loss, `self.pcRender(Pointclouds(...))`
// (engineer/networks/OptimGarmentNetwork.py:937, wrapper model/CameraMine.py:306-415, settings
// engineer/networks/OptimNetwork.py:87-100: radius 0.006 / 0.00465 / 0.0041 NDC, points_per_pixel = 50,
// AlphaCompositor).  The reference gets it from pytorch3d 0.4.0 (`rasterize_points`, `alpha_composite`), which is not
// vendored in the reference tree; the arithmetic restates the published algorithm
// (csrc/rasterize_points/rasterize_points.cu `RasterizePointsNaiveCudaKernel` / `RasterizePointsBackwardCudaKernel`,
// csrc/compositing/alpha_composite.cu) and is checked against oracle/recmv_oracle.c.
//
// How: pytorch3d walks every pixel over every point (naive) or over per-bin point lists (coarse-to-fine).  A splat of
// radius 1.5 pixels covers ~7 pixel centres, so the work here is organised by POINT:
//   count    one thread per point: +1 on every pixel centre inside its disc
//   place    one thread per pixel: prefix sum of the counts + ONE atomicAdd per workgroup reserves list storage
//   fill     one thread per point: writes (depth bits << 32 | point index) into the lists of its pixels
//   resolve  one thread per pixel: K rounds of "smallest key greater than the last one" -> sorted by (depth, index),
//            recomputes the squared distance, pads with -1
// List storage order depends on scheduling, the OUTPUT does not (it is sorted with a total order).
// The compositor and its backward are one thread per pixel; the rasteriser's backward is one thread per point (a fixed
// summation order per point: deterministic, unlike upstream's float atomics); grad_features keeps float atomics.
#include "common.h"

namespace recmv {
namespace {

#pragma clang fp contract(off)

// upstream's guard in the compositor backward: a point exactly on a pixel centre has alpha = 1 and would divide by 0
constexpr float kAlphaEps = 1e-9f;

__device__ __forceinline__ float pix_to_ndc(int i, int S) { return 1.f - (2.f * (float)i + 1.f) / (float)S; }

// Conservative index range of the pixel centres within [c - r, c + r] (NDC decreases with the index).
__device__ __forceinline__ void disc_range(float c, float r, int S, int& i0, int& i1) {
  const float a = ((float)S * (1.f - (c + r)) - 1.f) * 0.5f;
  const float b = ((float)S * (1.f - (c - r)) - 1.f) * 0.5f;
  const float fa = floorf(a) - 1.f, fb = ceilf(b) + 1.f;
  if (!(fa == fa) || !(fb == fb)) {
    i0 = 0;
    i1 = -1;
    return;
  }
  i0 = fa < 0.f ? 0 : (fa > (float)S ? S : (int)fa);
  i1 = fb > (float)(S - 1) ? S - 1 : (fb < -1.f ? -1 : (int)fb);
}

template <bool FILL>
__global__ void __launch_bounds__(256)
points_scatter_kernel(const float* __restrict__ pts, const int64_t* __restrict__ first,
                      const int64_t* __restrict__ count, int H, int W, float radius, int* __restrict__ pix_count,
                      const int64_t* __restrict__ pix_offset, int* __restrict__ pix_cursor,
                      unsigned long long* __restrict__ entries) {
  const int n = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count[n]) return;
  const int64_t p = first[n] + i;
  const float px = pts[3 * p], py = pts[3 * p + 1], pz = pts[3 * p + 2];
  if (!(pz >= 0.f)) return;
  const float r2 = radius * radius;
  int c0, c1, r0, r1;
  disc_range(px, radius, W, c0, c1);
  disc_range(py, radius, H, r0, r1);
  const int64_t base = (int64_t)n * H * W;
  for (int row = r0; row <= r1; ++row) {
    const float dy = pix_to_ndc(row, H) - py;
    for (int col = c0; col <= c1; ++col) {
      const float dx = pix_to_ndc(col, W) - px;
      const float d2 = dx * dx + dy * dy;
      if (!(d2 < r2)) continue;
      const int64_t pix = base + (int64_t)row * W + col;
      if (!FILL) {
        atomicAdd(pix_count + pix, 1);
      } else {
        const int slot = atomicAdd(pix_cursor + pix, 1);
        entries[pix_offset[pix] + slot] =
            ((unsigned long long)__float_as_uint(pz + 0.f) << 32) | (unsigned long long)(uint32_t)p;
      }
    }
  }
}

// Reserves list storage: wave prefix sums, combined across the 4 waves of the workgroup through LDS, ONE atomicAdd per
// workgroup (the single counter is the serial resource of this pass).
__global__ void __launch_bounds__(256)
points_place_kernel(const int* __restrict__ pix_count, int64_t npix, int64_t* __restrict__ pix_offset,
                    unsigned long long* __restrict__ total) {
  __shared__ int wave_sum[4];
  __shared__ unsigned long long block_base;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int c = i < npix ? pix_count[i] : 0;
  int incl = c;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int up = __shfl_up(incl, d, kWave);
    if (lane >= d) incl += up;
  }
  if (lane == kWave - 1) wave_sum[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int s = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    block_base = s > 0 ? atomicAdd(total, (unsigned long long)s) : 0ull;
  }
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += wave_sum[w];
  if (i < npix) pix_offset[i] = (int64_t)block_base + before + (incl - c);
}

// idx / zbuf / dists arrive pre-filled with -1: only the listed points are written (4 of 5 pixels are empty).
// A pixel's list is ordered by RANK: the position of an entry is the number of entries with a smaller key (keys are
// unique: they carry the point index).  Pixels with up to kLightPixel candidates are ranked by their own thread;
// longer lists (silhouette limbs, folds) are queued and ranked by a whole wavefront each, so that one crowded pixel
// does not stall the 63 others of its wave for O(n^2) loads.
constexpr int kLightPixel = 12;

__device__ __forceinline__ void write_fragment(const float* __restrict__ pts, unsigned long long key, int64_t out,
                                               float xf, float yf, int* __restrict__ idx, float* __restrict__ zbuf,
                                               float* __restrict__ dists) {
  const int64_t p = (int64_t)(uint32_t)(key & 0xffffffffull);
  const float dx = xf - pts[3 * p], dy = yf - pts[3 * p + 1];
  idx[out] = (int)p;
  zbuf[out] = __uint_as_float((uint32_t)(key >> 32));
  dists[out] = dx * dx + dy * dy;
}

__global__ void __launch_bounds__(256)
points_resolve_kernel(const float* __restrict__ pts, const int* __restrict__ pix_count,
                      const int64_t* __restrict__ pix_offset, const unsigned long long* __restrict__ entries,
                      int64_t npix, int H, int W, int K, int* __restrict__ idx, float* __restrict__ zbuf,
                      float* __restrict__ dists, int* __restrict__ heavy_list, unsigned long long* __restrict__ heavy_n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const int cnt = pix_count[i];
  if (cnt == 0) return;
  if (cnt > kLightPixel) {
    heavy_list[atomicAdd(heavy_n, 1ull)] = (int)i;   // order of the queue is irrelevant
    return;
  }
  const unsigned long long* e = entries + pix_offset[i];
  unsigned long long v[kLightPixel];
#pragma unroll
  for (int j = 0; j < kLightPixel; ++j) v[j] = j < cnt ? e[j] : ~0ull;
  const int64_t pix = i % ((int64_t)H * W);
  const float xf = pix_to_ndc((int)(pix % W), W), yf = pix_to_ndc((int)(pix / W), H);
#pragma unroll
  for (int j = 0; j < kLightPixel; ++j) {
    if (j < cnt) {
      int rank = 0;
#pragma unroll
      for (int t = 0; t < kLightPixel; ++t) rank += (v[t] < v[j]) ? 1 : 0;
      if (rank < K) write_fragment(pts, v[j], i * K + rank, xf, yf, idx, zbuf, dists);
    }
  }
}

__global__ void __launch_bounds__(256)
points_resolve_heavy_kernel(const float* __restrict__ pts, const int* __restrict__ pix_count,
                            const int64_t* __restrict__ pix_offset, const unsigned long long* __restrict__ entries,
                            int H, int W, int K, int* __restrict__ idx, float* __restrict__ zbuf,
                            float* __restrict__ dists, const int* __restrict__ heavy_list,
                            const unsigned long long* __restrict__ heavy_n) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) / kWave;
  const int64_t total = (int64_t)*heavy_n;
  for (int64_t q = wave; q < total; q += nwaves) {
    const int64_t i = heavy_list[q];
    const int cnt = pix_count[i];
    const unsigned long long* e = entries + pix_offset[i];
    const int64_t pix = i % ((int64_t)H * W);
    const float xf = pix_to_ndc((int)(pix % W), W), yf = pix_to_ndc((int)(pix / W), H);
    for (int j = lane; j < cnt; j += kWave) {
      const unsigned long long mine = e[j];
      int rank = 0;
      for (int t = 0; t < cnt; ++t) rank += (e[t] < mine) ? 1 : 0;   // uniform address: one broadcast load per t
      if (rank < K) write_fragment(pts, mine, i * K + rank, xf, yf, idx, zbuf, dists);
    }
  }
}

// One thread per POINT (not per pixel): it walks the pixel centres of its disc in row-major order and looks itself up in
// each pixel's K-list.  Every point sums its own contributions in a fixed order — no atomics, the gradient is a pure
// function of the inputs (the per-pixel formulation adds them with float atomics in scheduling order, upstream included).
__global__ void __launch_bounds__(256)
points_backward_kernel(const float* __restrict__ pts, const int64_t* __restrict__ first,
                       const int64_t* __restrict__ count, const int* __restrict__ idx,
                       const float* __restrict__ g_dists, const float* __restrict__ g_zbuf, int H, int W, int K,
                       float radius, float* __restrict__ g_pts) {
  const int n = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count[n]) return;
  const int64_t p = first[n] + i;
  const float px = pts[3 * p], py = pts[3 * p + 1], pz = pts[3 * p + 2];
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (pz >= 0.f) {
    const float r2 = radius * radius;
    int c0, c1, r0, r1;
    disc_range(px, radius, W, c0, c1);
    disc_range(py, radius, H, r0, r1);
    const int64_t base = (int64_t)n * H * W;
    for (int row = r0; row <= r1; ++row) {
      const float yf = pix_to_ndc(row, H);
      const float dy = yf - py;
      for (int col = c0; col <= c1; ++col) {
        const float xf = pix_to_ndc(col, W);
        const float dx = xf - px;
        if (!(dx * dx + dy * dy < r2)) continue;
        const int64_t e = (base + (int64_t)row * W + col) * K;
        for (int k = 0; k < K; ++k) {
          const int q = idx[e + k];
          if (q < 0) break;                     // lists are packed: the first -1 ends them
          if (q != (int)p) continue;
          const float gd = g_dists ? g_dists[e + k] : 0.f;
          gx = gx + 2.f * gd * (px - xf);
          gy = gy + 2.f * gd * (py - yf);
          if (g_zbuf) gz = gz + g_zbuf[e + k];
          break;
        }
      }
    }
  }
  g_pts[3 * p] = gx;
  g_pts[3 * p + 1] = gy;
  g_pts[3 * p + 2] = gz;
}

// images[n,c,y,x] = sum_k a_k prod_{l<k} (1 - a_l) features[c, idx_k];  a = alphas, or 1 - alphas / radius2 when
// radius2 != 0 (then `alphas` holds the rasteriser's squared distances: PointsRendererWithFrags' 1 - d2 / r^2).
// Lists are packed to the front (rasteriser contract), so the first -1 ends a pixel.
__device__ __forceinline__ float alpha_of(float v, float radius2) {
  return radius2 != 0.f ? 1.f - v / radius2 : v;
}

__global__ void __launch_bounds__(256)
alpha_forward_kernel(const int* __restrict__ idx, const float* __restrict__ alphas,
                     const float* __restrict__ features, int64_t P, int64_t npix, int64_t HW, int K, int C,
                     float radius2, float* __restrict__ images) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const int64_t n = i / HW, pix = i % HW;
  for (int c = 0; c < C; ++c) {
    float cum = 1.f, res = 0.f;
    for (int k = 0; k < K; ++k) {
      const int p = idx[i * K + k];
      if (p < 0) break;
      const float a = alpha_of(alphas[i * K + k], radius2);
      res += cum * a * features[(int64_t)c * P + p];
      cum = cum * (1.f - a);
    }
    images[(n * C + c) * HW + pix] = res;
  }
}

// g_alphas arrives zero-filled; only listed entries are touched.  The published double loop
//     g_a[t] = sum_c g_c ( cum_t f_tc - sum_{k>t} f_kc cum_k a_k / (1 - a_t + eps) )
// is linear in g, so with F_k = sum_c g_c f_kc and the "composite seen from behind t"
//     R_t = sum_{k>t} F_k a_k prod_{t<l<k} (1 - a_l)        (R_{t-1} = F_t a_t + (1 - a_t) R_t, R_last = 0)
// it becomes  g_a[t] = cum_t ( F_t - R_t (1 - a_t) / (1 - a_t + eps) ):  O(K) per pixel instead of O(K^2), no
// cancellation and no amplification by 1 / (1 - a_t).  R_t is parked in g_alphas between the two sweeps.
__global__ void __launch_bounds__(256)
alpha_backward_kernel(const int* __restrict__ idx, const float* __restrict__ alphas,
                      const float* __restrict__ features, const float* __restrict__ g_images, int64_t P, int64_t npix,
                      int64_t HW, int K, int C, float radius2, float* __restrict__ g_alphas,
                      float* __restrict__ g_features) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  if (idx[i * K] < 0) return;
  const int64_t n = i / HW, pix = i % HW;
  int len = 0;
  if (g_features) {
    float cum = 1.f;
    for (; len < K; ++len) {
      const int p = idx[i * K + len];
      if (p < 0) break;
      const float a = alpha_of(alphas[i * K + len], radius2);
      for (int c = 0; c < C; ++c)
        atomicAdd(g_features + (int64_t)c * P + p, cum * a * g_images[(n * C + c) * HW + pix]);
      cum = cum * (1.f - a);
    }
  } else {
    while (len < K && idx[i * K + len] >= 0) ++len;
  }
  float R = 0.f;
  for (int t = len - 1; t >= 0; --t) {
    const int p = idx[i * K + t];
    const float a = alpha_of(alphas[i * K + t], radius2);
    float F = 0.f;
    for (int c = 0; c < C; ++c) F += g_images[(n * C + c) * HW + pix] * features[(int64_t)c * P + p];
    g_alphas[i * K + t] = R;
    R = F * a + (1.f - a) * R;
  }
  float cum = 1.f;
  for (int t = 0; t < len; ++t) {
    const int p = idx[i * K + t];
    const float a = alpha_of(alphas[i * K + t], radius2);
    float F = 0.f;
    for (int c = 0; c < C; ++c) F += g_images[(n * C + c) * HW + pix] * features[(int64_t)c * P + p];
    const float one_m = 1.f - a;
    float ga = cum * (F - g_alphas[i * K + t] * (one_m / (one_m + kAlphaEps)));
    if (radius2 != 0.f) ga = -ga / radius2;   // chain rule of a = 1 - d / radius2
    g_alphas[i * K + t] = ga;
    cum = cum * one_m;
  }
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int64_t recmv_rasterize_points_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t total_points,
                                                          float radius) {
  if (N < 0 || H <= 0 || W <= 0 || total_points < 0 || !(radius >= 0.f)) return -1;
  // a disc of radius r covers at most (ceil(r * S) + 1)^2 pixel centres (pixel pitch 2/S in NDC)
  const int64_t span_x = (int64_t)(radius * (float)W) + 2, span_y = (int64_t)(radius * (float)H) + 2;
  const int64_t npix = N * H * W;
  // counts i32 | cursors i32 | offsets i64 | total u64 (+pad) | entries u64
  return npix * (4 + 4 + 8) + 64 + total_points * span_x * span_y * 8;
}

extern "C" int recmv_rasterize_points(const float* points, const int64_t* cloud_first_point,
                                      const int64_t* cloud_num_points, int64_t N, int64_t total_points,
                                      int64_t max_points_per_cloud, int64_t H, int64_t W, float radius,
                                      int points_per_pixel, int32_t* idx, float* zbuf, float* dists, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  RECMV_REQUIRE(N >= 0 && H > 0 && W > 0 && total_points >= 0 && points_per_pixel > 0, "rasterize_points: bad sizes");
  RECMV_REQUIRE(total_points < (1ll << 31) && N < 65536, "rasterize_points: too many points / clouds");
  RECMV_REQUIRE(radius > 0.f, "rasterize_points: radius must be > 0");
  RECMV_REQUIRE(workspace_bytes >= recmv_rasterize_points_workspace_bytes(N, H, W, total_points, radius),
                "rasterize_points: workspace too small");
  if (N == 0) return RECMV_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = N * H * W;
  char* ws = (char*)workspace;
  int* pix_count = (int*)ws;
  int* pix_cursor = (int*)(ws + npix * 4);
  int64_t* pix_offset = (int64_t*)(ws + npix * 8);
  unsigned long long* total = (unsigned long long*)(ws + npix * 16);
  unsigned long long* entries = (unsigned long long*)(ws + npix * 16 + 64);
  RECMV_HIP_TRY(hipMemsetAsync(ws, 0, (size_t)(npix * 8), s));              // counts + cursors
  RECMV_HIP_TRY(hipMemsetAsync(total, 0, 64, s));
  const unsigned pgrid = (unsigned)ceil_div(npix, 256);
  if (total_points > 0 && max_points_per_cloud > 0) {
    dim3 grid((unsigned)ceil_div(max_points_per_cloud, 256), (unsigned)N);
    points_scatter_kernel<false><<<grid, 256, 0, s>>>(points, cloud_first_point, cloud_num_points, (int)H, (int)W,
                                                      radius, pix_count, nullptr, nullptr, nullptr);
    points_place_kernel<<<pgrid, 256, 0, s>>>(pix_count, npix, pix_offset, total);
    points_scatter_kernel<true><<<grid, 256, 0, s>>>(points, cloud_first_point, cloud_num_points, (int)H, (int)W,
                                                     radius, nullptr, pix_offset, pix_cursor, entries);
  } else {
    points_place_kernel<<<pgrid, 256, 0, s>>>(pix_count, npix, pix_offset, total);
  }
  RECMV_HIP_TRY(hipMemsetAsync(idx, 0xFF, (size_t)(npix * points_per_pixel * 4), s));                  // -1
  RECMV_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)zbuf, 0xBF800000, (size_t)(npix * points_per_pixel), s));   // -1.f
  RECMV_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)dists, 0xBF800000, (size_t)(npix * points_per_pixel), s));
  // the fill cursors are dead after the fill pass: their storage becomes the queue of crowded pixels
  unsigned long long* heavy_n = total + 1;
  points_resolve_kernel<<<pgrid, 256, 0, s>>>(points, pix_count, pix_offset, entries, npix, (int)H, (int)W,
                                              points_per_pixel, idx, zbuf, dists, pix_cursor, heavy_n);
  points_resolve_heavy_kernel<<<kNumCU * 4, 256, 0, s>>>(points, pix_count, pix_offset, entries, (int)H, (int)W,
                                                         points_per_pixel, idx, zbuf, dists, pix_cursor, heavy_n);
  return check_launch("rasterize_points");
}

extern "C" int recmv_rasterize_points_backward(const float* points, const int64_t* cloud_first_point,
                                               const int64_t* cloud_num_points, const int32_t* idx,
                                               const float* grad_dists, const float* grad_zbuf, int64_t N,
                                               int64_t total_points, int64_t max_points_per_cloud, int64_t H,
                                               int64_t W, float radius, int points_per_pixel, float* grad_points,
                                               void* stream) {
  RECMV_REQUIRE(N >= 0 && H > 0 && W > 0 && total_points >= 0 && points_per_pixel > 0 && max_points_per_cloud >= 0,
                "rasterize_points_backward: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  // points outside every cloud's range get a zero gradient
  RECMV_HIP_TRY(hipMemsetAsync(grad_points, 0, (size_t)(total_points * 3 * sizeof(float)), s));
  if (N == 0 || total_points == 0 || max_points_per_cloud == 0) return RECMV_OK;
  RECMV_REQUIRE(points && cloud_first_point && cloud_num_points && idx && grad_points,
                "rasterize_points_backward: NULL pointer");
  points_backward_kernel<<<dim3((unsigned)ceil_div(max_points_per_cloud, 256), (unsigned)N), 256, 0, s>>>(
      points, cloud_first_point, cloud_num_points, idx, grad_dists, grad_zbuf, (int)H, (int)W, points_per_pixel, radius,
      grad_points);
  return check_launch("rasterize_points_backward");
}

extern "C" int recmv_alpha_composite_forward(const int32_t* idx, const float* alphas, const float* features,
                                             int64_t N, int64_t H, int64_t W, int points_per_pixel, int64_t C,
                                             int64_t total_points, float radius2, float* images, void* stream) {
  RECMV_REQUIRE(N >= 0 && H > 0 && W > 0 && points_per_pixel > 0 && C > 0 && total_points >= 0,
                "alpha_composite_forward: bad sizes");
  const int64_t npix = N * H * W;
  if (npix == 0) return RECMV_OK;
  alpha_forward_kernel<<<(unsigned)ceil_div(npix, 256), 256, 0, (hipStream_t)stream>>>(
      idx, alphas, features, total_points, npix, H * W, points_per_pixel, (int)C, radius2, images);
  return check_launch("alpha_composite_forward");
}

extern "C" int recmv_alpha_composite_backward(const int32_t* idx, const float* alphas, const float* features,
                                              const float* grad_images, int64_t N, int64_t H, int64_t W,
                                              int points_per_pixel, int64_t C, int64_t total_points,
                                              float radius2, float* grad_alphas, float* grad_features,
                                              void* stream) {
  RECMV_REQUIRE(N >= 0 && H > 0 && W > 0 && points_per_pixel > 0 && C > 0 && total_points >= 0,
                "alpha_composite_backward: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (grad_features) RECMV_HIP_TRY(hipMemsetAsync(grad_features, 0, (size_t)(C * total_points * sizeof(float)), s));
  const int64_t npix = N * H * W;
  if (npix == 0) return RECMV_OK;
  RECMV_HIP_TRY(hipMemsetAsync(grad_alphas, 0, (size_t)(npix * points_per_pixel * sizeof(float)), s));
  alpha_backward_kernel<<<(unsigned)ceil_div(npix, 256), 256, 0, s>>>(idx, alphas, features, grad_images, total_points,
                                                                       npix, H * W, points_per_pixel, (int)C,
                                                                       radius2, grad_alphas, grad_features);
  return check_launch("alpha_composite_backward");
}
