// Device-side bookkeeping of the coarse-to-fine SDF-grid evaluation (Seg3dLossless) — gfx950.
//
// What it replaces: per pyramid level, MCAcc/seg3d_lossless.py:296-422 of the reference runs a 3^3 box filter over the
// upsampled boundary mask, erases the already-evaluated voxels through a coordinate list, `nonzero`s a transposed
// volume, gathers / scatters by index lists, keeps the evaluated set as a coordinate list that it doubles and
// de-duplicates with `unique(dim=1)` (a lexicographic sort) twice per level, and grows the 27-neighbourhoods of sign
// conflicts with another `unique(dim=0)` — a dozen full-volume torch passes and half a dozen host round trips per level.
// Here the evaluated set is ONE BIT PER VOXEL of the current level and each step is one kernel:
//   select : need = dilate3(boundary) & ~evaluated  -> compacted voxel list (wave ballot + one atomic per wave) and the
//            level's evaluated bits (parent bits on the even lattice | need), written as whole 32-bit words
//   points : voxel list -> world points of the final-resolution lattice (`batch_eval`'s arithmetic, :89-108)
//   apply  : scatter the queried values, flag sign conflicts against the interpolated values (:337-346)
//   expand : 27-neighbourhood of the conflicts, clamped, not yet evaluated -> next list; `atomicOr` on the bit volume
//            both tests and marks a voxel, so the list is duplicate-free by construction (no `unique`)
// The host reads back one counter per query (it sizes the MLP launch).  Values are identical to the reference route:
// a voxel's value depends only on its own query, never on the list order.
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;
#pragma clang fp contract(off)

__device__ __forceinline__ bool bit_of(const uint32_t* __restrict__ bits, int64_t i) { return (bits[i >> 5] >> (i & 31)) & 1u; }

// One lane per voxel of the level, lanes along the linear index (x fastest).
__global__ __launch_bounds__(kBlk) void seg3d_select_kernel(const uint8_t* __restrict__ bnd,
                                                            const uint32_t* __restrict__ done_prev, int D, int H, int W,
                                                            int Hp, int Wp, uint32_t* __restrict__ done,
                                                            int32_t* __restrict__ list, int32_t* __restrict__ count,
                                                            int64_t cap) {
  const int64_t n = (int64_t)D * H * W;
  const int lane = threadIdx.x & (kWave - 1);
  for (int64_t base = ((int64_t)blockIdx.x * kBlk + (threadIdx.x & ~(kWave - 1))); base < n;
       base += (int64_t)gridDim.x * kBlk) {
    const int64_t i = base + lane;
    bool need = false, was = false;
    if (i < n) {
      const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((int64_t)W * H));
      if (!((x | y | z) & 1)) was = bit_of(done_prev, ((int64_t)(z >> 1) * Hp + (y >> 1)) * Wp + (x >> 1));
      if (!was) {
        const int z0 = z > 0 ? z - 1 : 0, z1 = z < D - 1 ? z + 1 : D - 1;
        const int y0 = y > 0 ? y - 1 : 0, y1 = y < H - 1 ? y + 1 : H - 1;
        const int x0 = x > 0 ? x - 1 : 0, x1 = x < W - 1 ? x + 1 : W - 1;
        unsigned int any = 0;
        for (int zz = z0; zz <= z1; ++zz)
          for (int yy = y0; yy <= y1; ++yy) {
            const uint8_t* row = bnd + ((int64_t)zz * H + yy) * W;
            for (int xx = x0; xx <= x1; ++xx) any |= row[xx];
          }
        need = any != 0;
      }
    }
    const unsigned long long m_need = __ballot(need), m_done = __ballot(need || was);
    if (lane == 0) {                                    // base is a multiple of 64: two whole words
      done[base >> 5] = (uint32_t)m_done;
      if (base + 32 < ((n + 31) & ~(int64_t)31)) done[(base >> 5) + 1] = (uint32_t)(m_done >> 32);
    }
    if (m_need) {
      int32_t start = 0;
      if (lane == 0) start = atomicAdd(count, __popcll(m_need));
      start = __shfl(start, 0);
      const int64_t slot = (int64_t)start + __popcll(m_need & ((1ull << lane) - 1ull));
      if (need && slot < cap) list[slot] = (int32_t)i;
    }
  }
}

// coords (x*sx, y*sy, z*sz) of the final lattice -> world: (c / R + (1/R)/2) * (bmax - bmin) + bmin  (:99-101)
__global__ __launch_bounds__(kBlk) void seg3d_points_kernel(const int32_t* __restrict__ list, int64_t n, int H, int W,
                                                            int sx, int sy, int sz, float rx, float ry, float rz,
                                                            float ex, float ey, float ez, float mx, float my, float mz,
                                                            float* __restrict__ pts) {
  const int64_t t = (int64_t)blockIdx.x * kBlk + threadIdx.x;
  if (t >= n) return;
  const int i = list[t];
  const int x = i % W, y = (i / W) % H, z = i / (W * H);
  const float cx = (float)(x * sx) / rx + (1.0f / rx) / 2.f;
  const float cy = (float)(y * sy) / ry + (1.0f / ry) / 2.f;
  const float cz = (float)(z * sz) / rz + (1.0f / rz) / 2.f;
  pts[3 * t + 0] = cx * ex + mx;
  pts[3 * t + 1] = cy * ey + my;
  pts[3 * t + 2] = cz * ez + mz;
}

__global__ __launch_bounds__(kBlk) void seg3d_apply_kernel(const int32_t* __restrict__ list, const float* __restrict__ vals,
                                                           int64_t n, float balance, float* __restrict__ occ,
                                                           uint8_t* __restrict__ flags, int32_t* __restrict__ nconf) {
  const int64_t t = (int64_t)blockIdx.x * kBlk + threadIdx.x;
  bool conflict = false;
  if (t < n) {
    const int i = list[t];
    const float interp = occ[i], v = vals[t];
    occ[i] = v;
    conflict = (interp - balance) * (v - balance) < 0.f;
    flags[t] = conflict ? 1 : 0;
  }
  const unsigned long long m = __ballot(conflict);
  if (m && (threadIdx.x & (kWave - 1)) == 0) atomicAdd(nconf, __popcll(m));
}

// One lane per (flagged voxel, neighbour): neighbours of the 3^3 block, clamped to the grid (:357-366).
__global__ __launch_bounds__(kBlk) void seg3d_expand_kernel(const int32_t* __restrict__ list,
                                                            const uint8_t* __restrict__ flags, int64_t n, int D, int H,
                                                            int W, uint32_t* __restrict__ done,
                                                            int32_t* __restrict__ out, int32_t* __restrict__ count,
                                                            int64_t cap) {
  const int64_t t = (int64_t)blockIdx.x * kBlk + threadIdx.x;
  const int64_t v = t / 27;
  const int nb = (int)(t - v * 27);
  bool fresh = false;
  int j = 0;
  if (v < n && flags[v]) {
    const int i = list[v];
    int x = i % W + nb % 3 - 1, y = (i / W) % H + (nb / 3) % 3 - 1, z = i / (W * H) + nb / 9 - 1;
    x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
    z = z < 0 ? 0 : (z > D - 1 ? D - 1 : z);
    j = (z * H + y) * W + x;
    const uint32_t bit = 1u << (j & 31);
    fresh = !(atomicOr(done + (j >> 5), bit) & bit);   // the first lane to set the bit owns the voxel
  }
  const unsigned long long m = __ballot(fresh);
  if (m) {
    const int lane = threadIdx.x & (kWave - 1);
    int32_t start = 0;
    if (lane == 0) start = atomicAdd(count, __popcll(m));
    start = __shfl(start, 0);
    const int64_t slot = (int64_t)start + __popcll(m & ((1ull << lane) - 1ull));
    if (fresh && slot < cap) out[slot] = j;
  }
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int recmv_seg3d_select(const uint8_t* is_boundary, const uint32_t* done_prev, int64_t D, int64_t H, int64_t W,
                                  uint32_t* done, int32_t* list, int64_t capacity, int32_t* count_device, void* stream) {
  RECMV_REQUIRE(D > 0 && H > 0 && W > 0 && D * H * W < (1ll << 31), "seg3d_select: bad sizes");
  RECMV_REQUIRE((D & 1) && (H & 1) && (W & 1), "seg3d_select: level sizes must be odd (2n-1 nesting)");
  RECMV_REQUIRE(is_boundary && done_prev && done && list && count_device && capacity >= 0, "seg3d_select: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  RECMV_HIP_TRY(hipMemsetAsync(count_device, 0, 4, s));
  const int64_t n = D * H * W;
  hipLaunchKernelGGL(seg3d_select_kernel, dim3(stream_grid(n, kBlk)), dim3(kBlk), 0, s, is_boundary, done_prev, (int)D,
                     (int)H, (int)W, (int)((H + 1) / 2), (int)((W + 1) / 2), done, list, count_device, capacity);
  return check_launch("seg3d_select");
}

extern "C" int recmv_seg3d_points(const int32_t* list, int64_t n, int64_t H, int64_t W, const int32_t* stride_xyz,
                                  const float* res_xyz, const float* extent_xyz, const float* bmin_xyz, float* points,
                                  void* stream) {
  if (n == 0) return RECMV_OK;
  RECMV_REQUIRE(list && stride_xyz && res_xyz && extent_xyz && bmin_xyz && points && n > 0, "seg3d_points: bad arguments");
  hipLaunchKernelGGL(seg3d_points_kernel, dim3((unsigned)ceil_div(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, list, n,
                     (int)H, (int)W, stride_xyz[0], stride_xyz[1], stride_xyz[2], res_xyz[0], res_xyz[1], res_xyz[2],
                     extent_xyz[0], extent_xyz[1], extent_xyz[2], bmin_xyz[0], bmin_xyz[1], bmin_xyz[2], points);
  return check_launch("seg3d_points");
}

extern "C" int recmv_seg3d_apply(const int32_t* list, const float* values, int64_t n, float balance, float* occupancy,
                                 uint8_t* conflict_flags, int32_t* conflict_count_device, void* stream) {
  RECMV_REQUIRE(conflict_count_device, "seg3d_apply: NULL counter");
  hipStream_t s = (hipStream_t)stream;
  RECMV_HIP_TRY(hipMemsetAsync(conflict_count_device, 0, 4, s));
  if (n == 0) return RECMV_OK;
  RECMV_REQUIRE(list && values && occupancy && conflict_flags && n > 0, "seg3d_apply: bad arguments");
  hipLaunchKernelGGL(seg3d_apply_kernel, dim3((unsigned)ceil_div(n, kBlk)), dim3(kBlk), 0, s, list, values, n, balance,
                     occupancy, conflict_flags, conflict_count_device);
  return check_launch("seg3d_apply");
}

extern "C" int recmv_seg3d_expand(const int32_t* list, const uint8_t* conflict_flags, int64_t n, int64_t D, int64_t H,
                                  int64_t W, uint32_t* done, int32_t* list_out, int64_t capacity, int32_t* count_device,
                                  void* stream) {
  RECMV_REQUIRE(count_device, "seg3d_expand: NULL counter");
  hipStream_t s = (hipStream_t)stream;
  RECMV_HIP_TRY(hipMemsetAsync(count_device, 0, 4, s));
  if (n == 0) return RECMV_OK;
  RECMV_REQUIRE(list && conflict_flags && done && list_out && n > 0 && capacity >= 0, "seg3d_expand: bad arguments");
  hipLaunchKernelGGL(seg3d_expand_kernel, dim3((unsigned)ceil_div(n * 27, kBlk)), dim3(kBlk), 0, s, list, conflict_flags,
                     n, (int)D, (int)H, (int)W, done, list_out, count_device, capacity);
  return check_launch("seg3d_expand");
}
