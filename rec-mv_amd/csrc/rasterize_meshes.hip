// First-hit mesh rasteriser (one face per pixel) — gfx950.
//
// What it computes: the fragments the reference's `maskRender` hands to `utils.FindSurfacePs`
// (engineer/networks/OptimGarmentNetwork.py:742-767 -> utils/FindSurfacePs.py:7-37): for every pixel the
// nearest face whose projection covers the pixel centre, its depth, perspective-corrected barycentrics and
// signed squared edge distance.  The reference gets them from pytorch3d 0.4.0's
// `rasterize_meshes(faces_per_pixel=1, blur_radius=0, perspective_correct=True, clip_barycentric_coords=False)`
// (RasterizationSettings at OptimGarmentNetwork.py:2336-2347); pytorch3d is a third-party dependency that is not
// vendored in the reference tree, so the per-(pixel, face) arithmetic below restates its published algorithm
// (pytorch3d/csrc/rasterize_meshes/rasterize_meshes.cu `CheckPixelInsideFace`, csrc/utils/geometry_utils.cuh) and is
// checked bit for bit against oracle/recmv_oracle.c:oracle_rasterize_meshes.
//
// How it computes it is different.  pytorch3d loops every pixel over every face (naive) or over per-bin face lists
// (coarse-to-fine, with a `max_faces_per_bin` overflow the reference has to work around, :2341).  A marching-cubes
// garment has ~2 faces per covered pixel, so the work here is organised by FACE:
//   pass 1  one thread per face walks the pixel centres inside the face's bounding box (usually 0-2 of them) and
//           does a 64-bit atomicMin of (depth bits << 32 | face index) on the pixel's key.  Faces whose box holds
//           more than kInlinePixels (48) centres are queued and walked by a whole wavefront each (pass 1b).
//           The minimum is order independent: ties in depth go to the lowest face index, which is what the
//           first-come `pz < q_max_z` test of the per-pixel loop keeps.
//   pass 2  one thread per pixel decodes the winner and recomputes its outputs with the same arithmetic.
// Traffic: 36 B per face read once + 8 B key + 32 B outputs per pixel; no per-bin lists, no overflow, deterministic.
#include "common.h"

namespace recmv {
namespace {

#pragma clang fp contract(off)

constexpr float kEps = 1e-8f;
constexpr int kInlinePixels = 48;
constexpr unsigned long long kEmptyKey = ~0ull;

struct Tri {
  float x0, y0, z0, x1, y1, z1, x2, y2, z2;
};

struct Hit {
  float z, dist, b0, b1, b2;
};

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

__device__ __forceinline__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by) {
  const float dx = bx - ax, dy = by - ay;
  const float l2 = dx * dx + dy * dy;
  if (l2 <= kEps) return (px - bx) * (px - bx) + (py - by) * (py - by);
  float t = (dx * (px - ax) + dy * (py - ay)) / l2;
  t = fminf(fmaxf(t, 0.f), 1.f);
  const float qx = ax + t * dx, qy = ay + t * dy;
  return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

// NDC coordinate of the centre of pixel column/row `i` of the OUTPUT image (pytorch3d's flipped axes folded in).
__device__ __forceinline__ float pix_to_ndc(int i, int S) { return 1.f - (2.f * (float)i + 1.f) / (float)S; }

// Face-level rejection: returns false when no pixel can take this face.
__device__ __forceinline__ bool face_ok(const Tri& t, int cull_backfaces) {
  const float zmax = fmaxf(fmaxf(t.z0, t.z1), t.z2);
  const float area = edge_fn(t.x0, t.y0, t.x1, t.y1, t.x2, t.y2);
  const bool zero_area = (area <= kEps) && (area >= -kEps);
  if (zmax < 0.f || zero_area || (cull_backfaces && area < 0.f)) return false;
  return !(area != area);
}

// One (pixel centre, face) test.  `blur` is the squared blur radius of pytorch3d's settings.
__device__ __forceinline__ bool pixel_face(const Tri& t, float px, float py, float blur, int perspective, Hit& h) {
  const float r = sqrtf(blur);
  const float xmin = fminf(fminf(t.x0, t.x1), t.x2) - r, xmax = fmaxf(fmaxf(t.x0, t.x1), t.x2) + r;
  const float ymin = fminf(fminf(t.y0, t.y1), t.y2) - r, ymax = fmaxf(fmaxf(t.y0, t.y1), t.y2) + r;
  if (px < xmin || px > xmax || py < ymin || py > ymax) return false;
  const float area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
  const float w0 = edge_fn(px, py, t.x1, t.y1, t.x2, t.y2) / area;
  const float w1 = edge_fn(px, py, t.x2, t.y2, t.x0, t.y0) / area;
  const float w2 = edge_fn(px, py, t.x0, t.y0, t.x1, t.y1) / area;
  float b0 = w0, b1 = w1, b2 = w2;
  if (perspective) {
    const float t0 = w0 * t.z1 * t.z2, t1 = t.z0 * w1 * t.z2, t2 = t.z0 * t.z1 * w2;
    const float den = fmaxf(t0 + t1 + t2, kEps);
    b0 = t0 / den;
    b1 = t1 / den;
    b2 = t2 / den;
  }
  const float pz = b0 * t.z0 + b1 * t.z1 + b2 * t.z2;
  if (!(pz >= 0.f)) return false;
  const float d01 = seg_dist2(px, py, t.x0, t.y0, t.x1, t.y1);
  const float d02 = seg_dist2(px, py, t.x0, t.y0, t.x2, t.y2);
  const float d12 = seg_dist2(px, py, t.x1, t.y1, t.x2, t.y2);
  const float dist = fminf(fminf(d01, d02), d12);
  const bool inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
  if (!inside && dist >= blur) return false;
  h.z = pz + 0.f;  // -0 -> +0 so that the bit pattern orders like the value
  h.dist = inside ? -dist : dist;
  h.b0 = b0;
  h.b1 = b1;
  h.b2 = b2;
  return true;
}

__device__ __forceinline__ Tri load_tri(const float* __restrict__ fv, int64_t f) {
  const float* p = fv + f * 9;
  Tri t;
  t.x0 = p[0]; t.y0 = p[1]; t.z0 = p[2];
  t.x1 = p[3]; t.y1 = p[4]; t.z1 = p[5];
  t.x2 = p[6]; t.y2 = p[7]; t.z2 = p[8];
  return t;
}

// Range of pixel indices whose centres can lie in [lo, hi] (NDC, decreasing in the index).  centre(i) = 1 - (2i+1)/S
// in [lo, hi]  <=>  (S(1-hi)-1)/2 <= i <= (S(1-lo)-1)/2; the bounds are relaxed by 1e-3 of a pixel (orders of
// magnitude above the rounding of the expression, so no covered centre is lost) — the exact test is pixel_face's.
__device__ __forceinline__ void pixel_range(float lo, float hi, int S, int& i0, int& i1) {
  const float a = ((float)S * (1.f - hi) - 1.f) * 0.5f;
  const float b = ((float)S * (1.f - lo) - 1.f) * 0.5f;
  const float fa = ceilf(a - 1e-3f - 1e-6f * fabsf(a)), fb = floorf(b + 1e-3f + 1e-6f * fabsf(b));
  if (!(fa == fa) || !(fb == fb)) {
    i0 = 0;
    i1 = -1;
    return;
  }
  i0 = fa < 0.f ? 0 : (fa > (float)S ? S : (int)fa);
  i1 = fb > (float)(S - 1) ? S - 1 : (fb < -1.f ? -1 : (int)fb);
}

__device__ __forceinline__ void try_pixel(const Tri& t, int64_t f, int row, int col, int H, int W, float blur,
                                          int perspective, unsigned long long* __restrict__ keys) {
  Hit h;
  if (!pixel_face(t, pix_to_ndc(col, W), pix_to_ndc(row, H), blur, perspective, h)) return;
  const unsigned long long key = ((unsigned long long)__float_as_uint(h.z) << 32) | (unsigned long long)(uint32_t)f;
  atomicMin(keys + (int64_t)row * W + col, key);
}

__global__ void __launch_bounds__(256) raster_fill_kernel(unsigned long long* keys, int64_t n, int* big_count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) keys[i] = kEmptyKey;
  if (blockIdx.x == 0 && threadIdx.x == 0) *big_count = 0;
}

__global__ void __launch_bounds__(256)
raster_faces_kernel(const float* __restrict__ fv, const int64_t* __restrict__ first, const int64_t* __restrict__ count,
                    int N, int64_t max_faces, int H, int W, float blur, int perspective, int cull,
                    unsigned long long* __restrict__ keys, int* __restrict__ big_count, int64_t* __restrict__ big_list,
                    int64_t big_cap) {
  const int n = blockIdx.y;
  const int64_t nf = count[n];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nf) return;
  const int64_t f = first[n] + i;
  const Tri t = load_tri(fv, f);
  if (!face_ok(t, cull)) return;
  const float r = sqrtf(blur);
  int c0, c1, r0, r1;
  pixel_range(fminf(fminf(t.x0, t.x1), t.x2) - r, fmaxf(fmaxf(t.x0, t.x1), t.x2) + r, W, c0, c1);
  pixel_range(fminf(fminf(t.y0, t.y1), t.y2) - r, fmaxf(fmaxf(t.y0, t.y1), t.y2) + r, H, r0, r1);
  if (c1 < c0 || r1 < r0) return;
  const int64_t npx = (int64_t)(c1 - c0 + 1) * (r1 - r0 + 1);
  unsigned long long* k = keys + (int64_t)n * H * W;
  if (npx > kInlinePixels) {
    const int slot = atomicAdd(big_count, 1);
    if (slot < big_cap) {
      big_list[2 * (int64_t)slot] = f;
      big_list[2 * (int64_t)slot + 1] = n;
      return;
    }
    // queue full (cannot happen with big_cap = total faces): fall through and walk the box here
  }
  for (int row = r0; row <= r1; ++row)
    for (int col = c0; col <= c1; ++col) try_pixel(t, f, row, col, H, W, blur, perspective, k);
}

// One wavefront per queued face; lanes stride over the pixels of its box.
__global__ void __launch_bounds__(256)
raster_big_faces_kernel(const float* __restrict__ fv, int H, int W, float blur, int perspective,
                        unsigned long long* __restrict__ keys, const int* __restrict__ big_count,
                        const int64_t* __restrict__ big_list, int64_t big_cap) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) / kWave;
  int64_t total = *big_count;
  if (total > big_cap) total = big_cap;
  const float r = sqrtf(blur);
  for (int64_t q = wave; q < total; q += nwaves) {
    const int64_t f = big_list[2 * q];
    const int n = (int)big_list[2 * q + 1];
    const Tri t = load_tri(fv, f);
    int c0, c1, r0, r1;
    pixel_range(fminf(fminf(t.x0, t.x1), t.x2) - r, fmaxf(fmaxf(t.x0, t.x1), t.x2) + r, W, c0, c1);
    pixel_range(fminf(fminf(t.y0, t.y1), t.y2) - r, fmaxf(fmaxf(t.y0, t.y1), t.y2) + r, H, r0, r1);
    const int bw = c1 - c0 + 1;
    const int64_t npx = (int64_t)bw * (r1 - r0 + 1);
    unsigned long long* k = keys + (int64_t)n * H * W;
    for (int64_t p = lane; p < npx; p += kWave)
      try_pixel(t, f, r0 + (int)(p / bw), c0 + (int)(p % bw), H, W, blur, perspective, k);
  }
}

__global__ void __launch_bounds__(256)
raster_resolve_kernel(const float* __restrict__ fv, const unsigned long long* __restrict__ keys, int64_t npix, int H,
                      int W, float blur, int perspective, int64_t* __restrict__ pix_to_face, float* __restrict__ zbuf,
                      float* __restrict__ bary, float* __restrict__ dists) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
    const unsigned long long key = keys[i];
    int64_t face = -1;
    Hit h = {-1.f, -1.f, -1.f, -1.f, -1.f};
    if (key != kEmptyKey) {
      face = (int64_t)(uint32_t)(key & 0xffffffffull);
      const int64_t pix = i % ((int64_t)H * W);
      const Tri t = load_tri(fv, face);
      pixel_face(t, pix_to_ndc((int)(pix % W), W), pix_to_ndc((int)(pix / W), H), blur, perspective, h);
    }
    pix_to_face[i] = face;
    zbuf[i] = h.z;
    dists[i] = h.dist;
    bary[3 * i + 0] = h.b0;
    bary[3 * i + 1] = h.b1;
    bary[3 * i + 2] = h.b2;
  }
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int64_t recmv_rasterize_meshes_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t total_faces) {
  if (N < 0 || H < 0 || W < 0 || total_faces < 0) return -1;
  // keys [N*H*W] u64 | queue of (face, mesh) pairs [total_faces] | counter
  return N * H * W * 8 + total_faces * 16 + 64;
}

extern "C" int recmv_rasterize_meshes(const float* face_verts, const int64_t* mesh_first_face,
                                      const int64_t* mesh_num_faces, int64_t N, int64_t total_faces,
                                      int64_t max_faces_per_mesh, int64_t H, int64_t W, float blur_radius,
                                      int perspective_correct, int cull_backfaces, int64_t* pix_to_face, float* zbuf,
                                      float* bary_coords, float* dists, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
  RECMV_REQUIRE(N >= 0 && H > 0 && W > 0 && total_faces >= 0, "rasterize_meshes: bad sizes");
  RECMV_REQUIRE(total_faces < (1ll << 31) && N * H * W < (1ll << 40), "rasterize_meshes: too many faces / pixels");
  RECMV_REQUIRE(N < 65536, "rasterize_meshes: at most 65535 meshes per call");
  RECMV_REQUIRE(blur_radius >= 0.f, "rasterize_meshes: blur_radius must be >= 0");
  RECMV_REQUIRE(workspace_bytes >= recmv_rasterize_meshes_workspace_bytes(N, H, W, total_faces),
                "rasterize_meshes: workspace too small");
  if (N == 0) return RECMV_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = N * H * W;
  auto* keys = (unsigned long long*)workspace;
  auto* big_list = (int64_t*)((char*)workspace + npix * 8);
  auto* big_count = (int*)((char*)workspace + npix * 8 + total_faces * 16);
  raster_fill_kernel<<<stream_grid(npix, 256), 256, 0, s>>>(keys, npix, big_count);
  if (total_faces > 0 && max_faces_per_mesh > 0) {
    dim3 grid((unsigned)ceil_div(max_faces_per_mesh, 256), (unsigned)N);
    raster_faces_kernel<<<grid, 256, 0, s>>>(face_verts, mesh_first_face, mesh_num_faces, (int)N, max_faces_per_mesh,
                                             (int)H, (int)W, blur_radius, perspective_correct, cull_backfaces, keys,
                                             big_count, big_list, total_faces);
    raster_big_faces_kernel<<<kNumCU * 2, 256, 0, s>>>(face_verts, (int)H, (int)W, blur_radius, perspective_correct,
                                                       keys, big_count, big_list, total_faces);
  }
  raster_resolve_kernel<<<stream_grid(npix, 256), 256, 0, s>>>(face_verts, keys, npix, (int)H, (int)W, blur_radius,
                                                               perspective_correct, pix_to_face, zbuf, bary_coords,
                                                               dists);
  return check_launch("rasterize_meshes");
}
