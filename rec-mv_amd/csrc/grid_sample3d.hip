// 3-D trilinear grid sampler (padding=border, align_corners=False) with backward and double backward —
// gfx950.
//
// Semantics follow MCAcc/cuda/GridSamplerMineKernel.cu of the reference:
//   forward :162-328 (unnormalise with a double intermediate :210-212, clip :33-35, corner order
//            tnw,tne,tsw,tse,bnw,bne,bsw,bse :269-309)
//   backward:333-570 (clip-with-grad-mask :44-60, grad_grid scale W/2 and mask :534-545)
//   dbackward:575-914 (mask folded into scale_x = 0.5*W*gix_mult :738-740)
// and accept arbitrary element strides like its TensorInfo arguments.
//
// Design (not the reference's): the reference walks C channel planes with scalar strided gathers
// (8*C cache lines per point) and always scatters 8*C atomics per point into a zero-filled grad_input
// that the hot path never uses (the volume is a frozen buffer).  Here
//   * a channels-last volume (stride[1]==1, what recmv's LBSkinner registers) is read with 16-byte
//     loads: each corner is one contiguous C*4-byte record, 8 records per point;
//   * grad_input is optional (NULL skips all atomics);
//   * one lane per sample point, wave64-coalesced grid/grad_grid/grad_output traffic, launch on the
//     caller's stream, grid capped at 256 CUs x 8 workgroups and grid-strided.
//
// Algorithmic bytes / point (C channels, f32): forward 12 + 4C (+ touched volume), backward
// (grad_grid only) 24 + 4C, dbackward 36 + 8C  (SURVEY.md §8d).
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;

struct Desc5 {
  int64_t size[5];
  int64_t stride[5];
};

inline Desc5 to_desc(const recmv_tensor5* t) {
  Desc5 d;
  for (int i = 0; i < 5; ++i) {
    d.size[i] = t->size[i];
    d.stride[i] = t->stride[i];
  }
  return d;
}

#include "gs3d_common.inc"

template <typename T, int VEC>
struct VecLoad;
template <typename T>
struct VecLoad<T, 1> {
  static __device__ __forceinline__ void ld(const T* p, T* v) { v[0] = *p; }
};
template <>
struct VecLoad<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float* v) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  }
};

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(kBlk) void gs3d_fwd_kernel(int64_t nthreads, const T* __restrict__ input,
                                                        Desc5 in, const T* __restrict__ grid, Desc5 gr,
                                                        T* __restrict__ output, Desc5 out) {
  const int64_t C = in.size[1], D = in.size[2], H = in.size[3], W = in.size[4];
  const int64_t oD = gr.size[1], oH = gr.size[2], oW = gr.size[3];
  for (int64_t index = (int64_t)blockIdx.x * kBlk + threadIdx.x; index < nthreads;
       index += (int64_t)gridDim.x * kBlk) {
    const int64_t w = index % oW, h = (index / oW) % oH, d = (index / (oH * oW)) % oD,
                  n = index / (oD * oH * oW);
    const T* g = grid + n * gr.stride[0] + d * gr.stride[1] + h * gr.stride[2] + w * gr.stride[3];
    const Cell<T> c = make_cell<T>(g[0], g[gr.stride[4]], g[2 * gr.stride[4]], W, H, D);
    T wgt[8];
    int64_t off[8];
    bool inb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      wgt[k] = c.fx[bx] * c.fy[by] * c.fz[bz];
      inb[k] = c.in_x[bx] && c.in_y[by] && c.in_z[bz];
      off[k] = (int64_t)(c.z0 + bz) * in.stride[2] + (int64_t)(c.y0 + by) * in.stride[3] +
               (int64_t)(c.x0 + bx) * in.stride[4];
    }
    const T* inp = input + n * in.stride[0];
    T* o = output + n * out.stride[0] + d * out.stride[2] + h * out.stride[3] + w * out.stride[4];
    for (int64_t ch = 0; ch < C; ch += VEC) {
      T acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = (T)0;
      T val[8][VEC];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (inb[k]) VecLoad<T, VEC>::ld(inp + off[k] + ch * in.stride[1], val[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (inb[k]) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = fma(val[k][v], wgt[k], acc[v]);
        }
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[(ch + v) * out.stride[1]] = acc[v];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, bool WITH_GINP>
__global__ __launch_bounds__(kBlk) void gs3d_bwd_kernel(int64_t nthreads, const T* __restrict__ input,
                                                        Desc5 in, const T* __restrict__ grid, Desc5 gr,
                                                        const T* __restrict__ gout, Desc5 go,
                                                        T* __restrict__ ginp, Desc5 gi,
                                                        T* __restrict__ ggrid) {
  const int64_t C = in.size[1], D = in.size[2], H = in.size[3], W = in.size[4];
  const int64_t oD = gr.size[1], oH = gr.size[2], oW = gr.size[3];
  for (int64_t index = (int64_t)blockIdx.x * kBlk + threadIdx.x; index < nthreads;
       index += (int64_t)gridDim.x * kBlk) {
    const int64_t w = index % oW, h = (index / oW) % oH, d = (index / (oH * oW)) % oD,
                  n = index / (oD * oH * oW);
    const T* g = grid + n * gr.stride[0] + d * gr.stride[1] + h * gr.stride[2] + w * gr.stride[3];
    const Cell<T> c = make_cell<T>(g[0], g[gr.stride[4]], g[2 * gr.stride[4]], W, H, D);
    int64_t off[8];
    bool inb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      inb[k] = c.in_x[bx] && c.in_y[by] && c.in_z[bz];
      off[k] = (int64_t)(c.z0 + bz) * in.stride[2] + (int64_t)(c.y0 + by) * in.stride[3] +
               (int64_t)(c.x0 + bx) * in.stride[4];
    }
    const T* inp = input + n * in.stride[0];
    const T* gop = gout + n * go.stride[0] + d * go.stride[2] + h * go.stride[3] + w * go.stride[4];
    T gix = (T)0, giy = (T)0, giz = (T)0;
    for (int64_t ch = 0; ch < C; ch += VEC) {
      T val[8][VEC];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (inb[k]) VecLoad<T, VEC>::ld(inp + off[k] + ch * in.stride[1], val[k]);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const T gO = gop[(ch + v) * go.stride[1]];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          RECMV_CORNER_BITS(k);
          if (WITH_GINP && inb[k]) {
            const int64_t goff = n * gi.stride[0] + (ch + v) * gi.stride[1] +
                                 (int64_t)(c.z0 + bz) * gi.stride[2] +
                                 (int64_t)(c.y0 + by) * gi.stride[3] + (int64_t)(c.x0 + bx) * gi.stride[4];
            atomicAdd(ginp + goff, c.fx[bx] * c.fy[by] * c.fz[bz] * gO);
          }
          if (inb[k]) {
            const T vv = val[k][v];
            const T tx = vv * c.fy[by] * c.fz[bz] * gO;
            const T ty = vv * c.fx[bx] * c.fz[bz] * gO;
            const T tz = vv * c.fx[bx] * c.fy[by] * gO;
            gix = bx ? gix + tx : gix - tx;
            giy = by ? giy + ty : giy - ty;
            giz = bz ? giz + tz : giz - tz;
          }
        }
      }
    }
    gix = (T)((double)(gix * (T)W) / 2.);
    giy = (T)((double)(giy * (T)H) / 2.);
    giz = (T)((double)(giz * (T)D) / 2.);
    T* gg = ggrid + index * 3;
    gg[0] = c.mx * gix;
    gg[1] = c.my * giy;
    gg[2] = c.mz * giz;
  }
}

// ---------------------------------------------------------------------------------------------
// double backward
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, bool WITH_GGI, bool WITH_GINP>
__global__ __launch_bounds__(kBlk) void gs3d_dbwd_kernel(
    int64_t nthreads, const T* __restrict__ ggI, Desc5 gI, const T* __restrict__ ggG, Desc5 gG,
    const T* __restrict__ input, Desc5 in, const T* __restrict__ grid, Desc5 gr,
    const T* __restrict__ gout, Desc5 go, T* __restrict__ ginp, Desc5 gi, T* __restrict__ ggrid,
    T* __restrict__ ggout, Desc5 ggo) {
  const int64_t C = in.size[1], D = in.size[2], H = in.size[3], W = in.size[4];
  const int64_t oD = gr.size[1], oH = gr.size[2], oW = gr.size[3];
  for (int64_t index = (int64_t)blockIdx.x * kBlk + threadIdx.x; index < nthreads;
       index += (int64_t)gridDim.x * kBlk) {
    const int64_t w = index % oW, h = (index / oW) % oH, d = (index / (oH * oW)) % oD,
                  n = index / (oD * oH * oW);
    const T* g = grid + n * gr.stride[0] + d * gr.stride[1] + h * gr.stride[2] + w * gr.stride[3];
    const Cell<T> c = make_cell<T>(g[0], g[gr.stride[4]], g[2 * gr.stride[4]], W, H, D);
    const T* gg = ggG + n * gG.stride[0] + d * gG.stride[1] + h * gG.stride[2] + w * gG.stride[3];
    const T ggx = gg[0], ggy = gg[gG.stride[4]], ggz = gg[2 * gG.stride[4]];
    const T scale_x = (T)(0.5 * (double)(T)W * (double)c.mx);
    const T scale_y = (T)(0.5 * (double)(T)H * (double)c.my);
    const T scale_z = (T)(0.5 * (double)(T)D * (double)c.mz);
    const T scale_xy = scale_x * scale_y, scale_xz = scale_x * scale_z, scale_yz = scale_y * scale_z;

    int64_t off[8], offI[8];
    bool inb[8];
    T wgt[8], tmp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      inb[k] = c.in_x[bx] && c.in_y[by] && c.in_z[bz];
      off[k] = (int64_t)(c.z0 + bz) * in.stride[2] + (int64_t)(c.y0 + by) * in.stride[3] +
               (int64_t)(c.x0 + bx) * in.stride[4];
      if (WITH_GGI)
        offI[k] = (int64_t)(c.z0 + bz) * gI.stride[2] + (int64_t)(c.y0 + by) * gI.stride[3] +
                  (int64_t)(c.x0 + bx) * gI.stride[4];
      wgt[k] = c.fx[bx] * c.fy[by] * c.fz[bz];
      // tmp_k = sx*ggx*scale_x*fy*fz + sy*ggy*scale_y*fx*fz + sz*ggz*scale_z*fx*fy   (:746-753)
      const T a = (bx ? ggx : -ggx) * scale_x * c.fy[by] * c.fz[bz];
      const T b = ggy * scale_y * c.fx[bx] * c.fz[bz];
      const T e = ggz * scale_z * c.fx[bx] * c.fy[by];
      T t = by ? a + b : a - b;
      tmp[k] = bz ? t + e : t - e;
    }
    const T* inp = input + n * in.stride[0];
    const T* gop = gout + n * go.stride[0] + d * go.stride[2] + h * go.stride[3] + w * go.stride[4];
    T* ggop = ggout + n * ggo.stride[0] + d * ggo.stride[2] + h * ggo.stride[3] + w * ggo.stride[4];
    T gix = (T)0, giy = (T)0, giz = (T)0;
    for (int64_t ch = 0; ch < C; ch += VEC) {
      T val[8][VEC], vI[8][VEC];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (inb[k]) {
          VecLoad<T, VEC>::ld(inp + off[k] + ch * in.stride[1], val[k]);
          if (WITH_GGI) VecLoad<T, VEC>::ld(ggI + n * gI.stride[0] + offI[k] + ch * gI.stride[1], vI[k]);
        }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const T gO = gop[(ch + v) * go.stride[1]];
        T ggO = (T)0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          RECMV_CORNER_BITS(k);
          if (WITH_GINP && inb[k]) {
            const int64_t goff = n * gi.stride[0] + (ch + v) * gi.stride[1] +
                                 (int64_t)(c.z0 + bz) * gi.stride[2] +
                                 (int64_t)(c.y0 + by) * gi.stride[3] + (int64_t)(c.x0 + bx) * gi.stride[4];
            atomicAdd(ginp + goff, tmp[k] * gO);
          }
          if (inb[k]) {
            if (WITH_GGI) {
              const T t = vI[k][v];
              const T tx = t * c.fy[by] * c.fz[bz] * gO * scale_x;
              const T ty = t * c.fx[bx] * c.fz[bz] * gO * scale_y;
              const T tz = t * c.fx[bx] * c.fy[by] * gO * scale_z;
              gix = bx ? gix + tx : gix - tx;
              giy = by ? giy + ty : giy - ty;
              giz = bz ? giz + tz : giz - tz;
              ggO = ggO + t * wgt[k];
            }
            const T vv = val[k][v];
            // second derivatives of the trilinear weight: d2w/dxdy = sx*sy*fz, etc.
            const T dxy = (bx == by) ? c.fz[bz] : -c.fz[bz];
            const T dxz = (bx == bz) ? c.fy[by] : -c.fy[by];
            const T dyz = (by == bz) ? c.fx[bx] : -c.fx[bx];
            gix = gix + vv * (ggy * dxy * scale_xy + ggz * dxz * scale_xz) * gO;
            giy = giy + vv * (ggx * dxy * scale_xy + ggz * dyz * scale_yz) * gO;
            giz = giz + vv * (ggx * dxz * scale_xz + ggy * dyz * scale_yz) * gO;
            ggO = ggO + vv * tmp[k];
          }
        }
        ggop[(ch + v) * ggo.stride[1]] = ggO;
      }
    }
    T* out = ggrid + index * 3;
    out[0] = gix;
    out[1] = giy;
    out[2] = giz;
  }
}

// ---------------------------------------------------------------------------------------------
// backward / double backward, record-coalesced lanes (the default when only grad_grid is asked for)
// ---------------------------------------------------------------------------------------------
// The one-lane-per-point kernels above walk the C channels of a point serially (C/4 rounds of 8 dependent 16-byte gathers per
// lane) so that the channel sum has the reference's order.  north_star asks the GRADIENTS within an f32 tolerance, not bit for
// bit, so the product path gives that order up: lane t of a 256-thread workgroup serves point t / G and the V = C / G channels
// [cg V, cg V + V) of it (cg = t % G, G a power of two <= 8: C = 24 -> 8 lanes x 3 channels).  The G lanes of a point read ONE
// corner record (4 C contiguous bytes) with one instruction, every lane has its 8 gathers and its V grad_output loads in flight
// at once, and the per-point sums over channels are finished by a log2(G)-step butterfly (__shfl_xor inside the aligned lane
// group: a fixed order, reproducible run to run).  Algebra: with s_k = sum_c val_k[c] gO[c] per corner k,
//     backward   d/dx = sum_k (+-) fy fz s_k  (and cyclic), then the reference's W/2 scale and clip mask (:534-545);
//     dbackward  grad_grid_x = sum_k s_k (ggy dxy_k sxy + ggz dxz_k sxz) (and cyclic; :760-905 with ggI = 0),
//                ggO[c] = sum_k val_k[c] tmp_k  (tmp_k as in gs3d_dbwd_kernel).
// Same terms as the exact kernels, summed in a different order: |difference| <= a few ulp of sum |terms|
// (tests/test_gpu_kernels.py states the bound).  recmv_set_sampler_mode(1) / RECMV_SAMPLER_EXACT=1 selects the exact kernels.
template <int V>
struct RecLoad {
  struct alignas(4) Rec {      // 4-byte alignment is all the global_load_dwordx{2,3,4} of gfx950 need
    float v[V];
  };
  static __device__ __forceinline__ void ld(const float* p, float* v) {
    const Rec t = *reinterpret_cast<const Rec*>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = t.v[i];
  }
};

template <int G>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
  for (int m = 1; m < G; m <<= 1) x += __shfl_xor(x, m, 64);
  return x;
}

// What the record-coalesced kernels take from the host: a LIST of points per batch item (grid [N,1,1,P,3], the only shape the loop
// and the reference's callers use), every tensor small enough for 32-bit element offsets (checked by rec_ok / rec_fits).
struct RecGeom {
  int W, H, D;             // volume extent
  int P;                   // points per batch item
  int in_n, in_z, in_y, in_x;          // volume strides (elements); channel stride 1
  int gr_n, gr_p, gr_c;                // grid strides: batch, point, coordinate
  int go_n, go_c, go_p;                // grad_output strides: batch, channel, point
  int gg_n, gg_p, gg_c;                // ggG strides (double backward)
  int o_n, o_c, o_p;                   // grad_grad_output strides (double backward)
};

// Forward on the same lanes: out[c] = the fma chain over the 8 corners of gs3d_fwd_kernel per channel (bit-identical: a skipped
// corner and a zeroed one leave the accumulator unchanged); every lane stores its own V channels.
template <int G, int V>
__global__ __launch_bounds__(kBlk) void gs3d_fwd_rec_kernel(int nthreads, RecGeom q, const float* __restrict__ input,
                                                            const float* __restrict__ grid, float* __restrict__ output) {
  constexpr int PTS = kBlk / G;
  const int t = threadIdx.x, pt = t / G, cg = t - pt * G, ch = cg * V;
  const int tiles = (nthreads + PTS - 1) / PTS, per_xcd = (tiles + kNumXCD - 1) / kNumXCD;
  const int xcd = blockIdx.x % kNumXCD, tile_end = min((xcd + 1) * per_xcd, tiles);
  for (int tile = xcd * per_xcd + blockIdx.x / kNumXCD; tile < tile_end; tile += gridDim.x / kNumXCD) {
    const int index = tile * PTS + pt;
    if (index >= nthreads) continue;
    const int n = (unsigned)index / (unsigned)q.P, w = index - n * q.P;
    const float* g = grid + (n * q.gr_n + w * q.gr_p);
    const Cell<float> c = make_cell<float>(g[0], g[q.gr_c], g[2 * q.gr_c], q.W, q.H, q.D);
    const float* inp = input + (n * q.in_n + ch);
    const int ox[2] = {min(max(c.x0, 0), q.W - 1) * q.in_x, min(max(c.x0 + 1, 0), q.W - 1) * q.in_x};
    const int oy[2] = {min(max(c.y0, 0), q.H - 1) * q.in_y, min(max(c.y0 + 1, 0), q.H - 1) * q.in_y};
    const int oz[2] = {min(max(c.z0, 0), q.D - 1) * q.in_z, min(max(c.z0 + 1, 0), q.D - 1) * q.in_z};
    float val[8][V], acc[V];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      RecLoad<V>::ld(inp + (oz[bz] + oy[by] + ox[bx]), val[k]);
    }
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      const bool inb = c.in_x[bx] && c.in_y[by] && c.in_z[bz];
      const float wgt = c.fx[bx] * c.fy[by] * c.fz[bz];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = fma(inb ? val[k][v] : 0.f, wgt, acc[v]);
    }
    float* o = output + (n * q.go_n + w * q.go_p + ch * q.go_c);      // (the output descriptor travels in the go_* fields)
#pragma unroll
    for (int v = 0; v < V; ++v) o[v * q.go_c] = acc[v];
  }
}

template <int G, int V>
__global__ __launch_bounds__(kBlk) void gs3d_bwd_rec_kernel(int nthreads, RecGeom q, const float* __restrict__ input,
                                                            const float* __restrict__ grid, const float* __restrict__ gout,
                                                            float* __restrict__ ggrid) {
  constexpr int PTS = kBlk / G;
  const int t = threadIdx.x, pt = t / G, cg = t - pt * G, ch = cg * V;
  // Tiles of PTS consecutive points; workgroup b runs on XCD b % 8 (round-robin dispatch), so XCD x takes the x-th EIGHTH of the
  // tiles instead of every eighth tile: neighbouring points share corner records, and a contiguous range keeps them in one L2.
  const int tiles = (nthreads + PTS - 1) / PTS, per_xcd = (tiles + kNumXCD - 1) / kNumXCD;
  const int xcd = blockIdx.x % kNumXCD, tile_end = min((xcd + 1) * per_xcd, tiles);
  for (int tile = xcd * per_xcd + blockIdx.x / kNumXCD; tile < tile_end; tile += gridDim.x / kNumXCD) {
    const int base = tile * PTS;
    const bool live = base + pt < nthreads;
    const int index = live ? base + pt : nthreads - 1;      // (dead lanes shadow the last point: the butterfly needs them)
    const int n = (unsigned)index / (unsigned)q.P, w = index - n * q.P;
    const float* g = grid + (n * q.gr_n + w * q.gr_p);
    const Cell<float> c = make_cell<float>(g[0], g[q.gr_c], g[2 * q.gr_c], q.W, q.H, q.D);
    const float* inp = input + (n * q.in_n + ch);
    const int ox[2] = {min(max(c.x0, 0), q.W - 1) * q.in_x, min(max(c.x0 + 1, 0), q.W - 1) * q.in_x};
    const int oy[2] = {min(max(c.y0, 0), q.H - 1) * q.in_y, min(max(c.y0 + 1, 0), q.H - 1) * q.in_y};
    const int oz[2] = {min(max(c.z0, 0), q.D - 1) * q.in_z, min(max(c.z0 + 1, 0), q.D - 1) * q.in_z};
    const float* gop = gout + (n * q.go_n + w * q.go_p + ch * q.go_c);
    // every corner is gathered unconditionally from a clamped (always valid) address and zeroed afterwards when the reference
    // would have skipped it (out of the volume: a coordinate exactly on the far border, or a non-finite one) — no branches
    float val[8][V], gO[V];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      RecLoad<V>::ld(inp + (oz[bz] + oy[by] + ox[bx]), val[k]);
    }
#pragma unroll
    for (int v = 0; v < V; ++v) gO[v] = gop[v * q.go_c];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      const bool inb = c.in_x[bx] && c.in_y[by] && c.in_z[bz];
#pragma unroll
      for (int v = 0; v < V; ++v) val[k][v] = inb ? val[k][v] : 0.f;
    }
    float gix = 0.f, giy = 0.f, giz = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      float sk = 0.f;
#pragma unroll
      for (int v = 0; v < V; ++v) sk = fma(val[k][v], gO[v], sk);
      const float tx = sk * (c.fy[by] * c.fz[bz]), ty = sk * (c.fx[bx] * c.fz[bz]), tz = sk * (c.fx[bx] * c.fy[by]);
      gix = bx ? gix + tx : gix - tx;
      giy = by ? giy + ty : giy - ty;
      giz = bz ? giz + tz : giz - tz;
    }
    gix = group_sum<G>(gix);
    giy = group_sum<G>(giy);
    giz = group_sum<G>(giz);
    if (cg == 0 && live) {
      gix = (float)((double)(gix * (float)q.W) / 2.);
      giy = (float)((double)(giy * (float)q.H) / 2.);
      giz = (float)((double)(giz * (float)q.D) / 2.);
      float* gg = ggrid + (int64_t)index * 3;
      gg[0] = c.mx * gix;
      gg[1] = c.my * giy;
      gg[2] = c.mz * giz;
    }
  }
}

template <int G, int V>
__global__ __launch_bounds__(kBlk) void gs3d_dbwd_rec_kernel(int nthreads, RecGeom q, const float* __restrict__ ggG,
                                                             const float* __restrict__ input, const float* __restrict__ grid,
                                                             const float* __restrict__ gout, float* __restrict__ ggrid,
                                                             float* __restrict__ ggout) {
  constexpr int PTS = kBlk / G;
  const int t = threadIdx.x, pt = t / G, cg = t - pt * G, ch = cg * V;
  // Tiles of PTS consecutive points; workgroup b runs on XCD b % 8 (round-robin dispatch), so XCD x takes the x-th EIGHTH of the
  // tiles instead of every eighth tile: neighbouring points share corner records, and a contiguous range keeps them in one L2.
  const int tiles = (nthreads + PTS - 1) / PTS, per_xcd = (tiles + kNumXCD - 1) / kNumXCD;
  const int xcd = blockIdx.x % kNumXCD, tile_end = min((xcd + 1) * per_xcd, tiles);
  for (int tile = xcd * per_xcd + blockIdx.x / kNumXCD; tile < tile_end; tile += gridDim.x / kNumXCD) {
    const int base = tile * PTS;
    const bool live = base + pt < nthreads;
    const int index = live ? base + pt : nthreads - 1;
    const int n = (unsigned)index / (unsigned)q.P, w = index - n * q.P;
    const float* g = grid + (n * q.gr_n + w * q.gr_p);
    const Cell<float> c = make_cell<float>(g[0], g[q.gr_c], g[2 * q.gr_c], q.W, q.H, q.D);
    const float* gg = ggG + (n * q.gg_n + w * q.gg_p);
    const float ggx = gg[0], ggy = gg[q.gg_c], ggz = gg[2 * q.gg_c];
    const float* inp = input + (n * q.in_n + ch);
    const int ox[2] = {min(max(c.x0, 0), q.W - 1) * q.in_x, min(max(c.x0 + 1, 0), q.W - 1) * q.in_x};
    const int oy[2] = {min(max(c.y0, 0), q.H - 1) * q.in_y, min(max(c.y0 + 1, 0), q.H - 1) * q.in_y};
    const int oz[2] = {min(max(c.z0, 0), q.D - 1) * q.in_z, min(max(c.z0 + 1, 0), q.D - 1) * q.in_z};
    const float* gop = gout + (n * q.go_n + w * q.go_p + ch * q.go_c);
    // every corner is gathered unconditionally from a clamped (always valid) address and zeroed afterwards when the reference
    // would have skipped it (out of the volume: a coordinate exactly on the far border, or a non-finite one) — no branches
    float val[8][V], gO[V];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      RecLoad<V>::ld(inp + (oz[bz] + oy[by] + ox[bx]), val[k]);
    }
#pragma unroll
    for (int v = 0; v < V; ++v) gO[v] = gop[v * q.go_c];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      const bool inb = c.in_x[bx] && c.in_y[by] && c.in_z[bz];
#pragma unroll
      for (int v = 0; v < V; ++v) val[k][v] = inb ? val[k][v] : 0.f;
    }
    const float scale_x = (float)(0.5 * (double)(float)q.W * (double)c.mx);
    const float scale_y = (float)(0.5 * (double)(float)q.H * (double)c.my);
    const float scale_z = (float)(0.5 * (double)(float)q.D * (double)c.mz);
    const float scale_xy = scale_x * scale_y, scale_xz = scale_x * scale_z, scale_yz = scale_y * scale_z;
    float gix = 0.f, giy = 0.f, giz = 0.f, ggO[V];
#pragma unroll
    for (int v = 0; v < V; ++v) ggO[v] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      {
        // tmp_k = sx ggx scale_x fy fz + sy ggy scale_y fx fz + sz ggz scale_z fx fy   (:746-753)
        const float a = (bx ? ggx : -ggx) * scale_x * c.fy[by] * c.fz[bz];
        const float b = ggy * scale_y * c.fx[bx] * c.fz[bz];
        const float e = ggz * scale_z * c.fx[bx] * c.fy[by];
        const float tt = by ? a + b : a - b;
        const float tmp = bz ? tt + e : tt - e;
        // second derivatives of the trilinear weight: d2w/dxdy = sx sy fz, ...
        const float dxy = (bx == by) ? c.fz[bz] : -c.fz[bz];
        const float dxz = (bx == bz) ? c.fy[by] : -c.fy[by];
        const float dyz = (by == bz) ? c.fx[bx] : -c.fx[bx];
        float sk = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          sk = fma(val[k][v], gO[v], sk);
          ggO[v] = fma(val[k][v], tmp, ggO[v]);
        }
        gix = fma(sk, ggy * dxy * scale_xy + ggz * dxz * scale_xz, gix);
        giy = fma(sk, ggx * dxy * scale_xy + ggz * dyz * scale_yz, giy);
        giz = fma(sk, ggx * dxz * scale_xz + ggy * dyz * scale_yz, giz);
      }
    }
    gix = group_sum<G>(gix);
    giy = group_sum<G>(giy);
    giz = group_sum<G>(giz);
    if (live) {
      float* o = ggout + (n * q.o_n + w * q.o_p + ch * q.o_c);
#pragma unroll
      for (int v = 0; v < V; ++v) o[v * q.o_c] = ggO[v];
      if (cg == 0) {
        float* out = ggrid + (int64_t)index * 3;
        out[0] = gix;
        out[1] = giy;
        out[2] = giz;
      }
    }
  }
}

int g_sampler_mode = 0;      // 0: record-coalesced backward / double backward (tolerance), 1: the reference's summation order

// (G, V) for the record-coalesced kernels: the fewest lanes (a power of two <= 8) that leave at most 12 channels per lane — the per-
// point geometry is repeated in every lane of a point, and 8 x V values per lane are in flight at once; 0 when C does not split
// (profiles/r03_kernel_only_sampler_v2.txt: 2 lanes x 12 channels beats 4 x 6 beats 8 x 3 at C = 24, except the double backward below
// ~5e5 points and the forward below ~1.5e5).  Modes 2 / 3 / 4 force 8 x 3 / 4 x 6 / 2 x 12 at C = 24 (measurement).
inline int rec_lanes(int64_t C, int64_t count, int pass /* 0 forward, 1 backward, 2 double backward */) {
  if (C == 24 && g_sampler_mode >= 2) return g_sampler_mode == 2 ? 8 : g_sampler_mode == 3 ? 4 : 2;
  if (C == 24 && ((pass == 2 && count <= 500000) || (pass == 0 && count <= 150000))) return 4;
  for (int G = 1; G <= 8; G <<= 1)
    if (C % G == 0 && C / G <= 12) return G;
  return 0;
}

// largest element offset a descriptor can produce
inline int64_t span(const Desc5& d) {
  int64_t m = 0;
  for (int i = 0; i < 5; ++i) m += (d.size[i] > 0 ? d.size[i] - 1 : 0) * (d.stride[i] < 0 ? -d.stride[i] : d.stride[i]);
  return m;
}
inline bool fits32(const Desc5& d) {
  for (int i = 0; i < 5; ++i)
    if (d.stride[i] < 0) return false;
  return span(d) < (1ll << 31) - 64;
}

// record-coalesced kernels need: f32, dense channels-last records (stride[1] == 1) aligned for the V-wide loads, a list of points
// (grid [N,1,1,P,3]) and 32-bit element offsets everywhere
inline bool rec_ok(const void* p, const Desc5& d, const Desc5& gr, int dtype) {
  if (g_sampler_mode == 1 || dtype != RECMV_F32 || d.stride[1] != 1 || !rec_lanes(d.size[1], 0, 1)) return false;
  if (reinterpret_cast<uintptr_t>(p) % 4 != 0) return false;              // (RecLoad: 4-byte alignment suffices)
  return gr.size[1] * gr.size[2] == 1 && gr.size[0] * gr.size[3] < (1ll << 31) / 4 && fits32(d) && fits32(gr);
}

inline RecGeom rec_geom(const Desc5& in, const Desc5& gr, const Desc5& go) {
  RecGeom q = {};
  q.W = (int)in.size[4], q.H = (int)in.size[3], q.D = (int)in.size[2], q.P = (int)gr.size[3];
  q.in_n = (int)in.stride[0], q.in_z = (int)in.stride[2], q.in_y = (int)in.stride[3], q.in_x = (int)in.stride[4];
  q.gr_n = (int)gr.stride[0], q.gr_p = (int)gr.stride[3], q.gr_c = (int)gr.stride[4];
  q.go_n = (int)go.stride[0], q.go_c = (int)go.stride[1], q.go_p = (int)go.stride[4];
  return q;
}

#define RECMV_REC_DISPATCH(C, G, LAUNCH)                  \
  switch ((int)(C) * 16 + (G)) {                          \
    case 1 * 16 + 1: LAUNCH(1, 1); break;                 \
    case 2 * 16 + 1: LAUNCH(1, 2); break;                 \
    case 3 * 16 + 1: LAUNCH(1, 3); break;                 \
    case 4 * 16 + 1: LAUNCH(1, 4); break;                 \
    case 6 * 16 + 1: LAUNCH(1, 6); break;                 \
    case 8 * 16 + 1: LAUNCH(1, 8); break;                 \
    case 12 * 16 + 1: LAUNCH(1, 12); break;               \
    case 16 * 16 + 2: LAUNCH(2, 8); break;                \
    case 24 * 16 + 2: LAUNCH(2, 12); break;               \
    case 24 * 16 + 4: LAUNCH(4, 6); break;                \
    case 24 * 16 + 8: LAUNCH(8, 3); break;                \
    case 32 * 16 + 4: LAUNCH(4, 8); break;                \
    case 48 * 16 + 4: LAUNCH(4, 12); break;               \
    case 64 * 16 + 8: LAUNCH(8, 8); break;                \
    default: handled = false; break;                      \
  }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Channels-last f32 volume with 16-byte aligned records -> 4-wide loads.
inline bool vec4_ok(const void* p, const Desc5& d, int dtype) {
  return dtype == RECMV_F32 && d.stride[1] == 1 && d.size[1] % 4 == 0 && aligned16(p) &&
         d.stride[0] % 4 == 0 && d.stride[2] % 4 == 0 && d.stride[3] % 4 == 0 && d.stride[4] % 4 == 0;
}

int check_common(const char* who, const recmv_tensor5* in, const recmv_tensor5* gr, int interp, int pad,
                 int dtype) {
  RECMV_REQUIRE(in && gr, "%s: NULL descriptor", who);
  if (interp != 0) {
    set_error("grid_sampler(): only support Bilinear now");
    return RECMV_ERR_UNSUPPORTED;
  }
  if (pad != 1) {
    set_error("grid_sampler(): only support Border Padding now");
    return RECMV_ERR_UNSUPPORTED;
  }
  if (dtype != RECMV_F32 && dtype != RECMV_F64) {
    set_error("%s: dtype %d unsupported", who, dtype);
    return RECMV_ERR_UNSUPPORTED;
  }
  RECMV_REQUIRE(in->size[0] == gr->size[0],
                "grid_sampler(): expected grid and input to have same batch size");
  RECMV_REQUIRE(gr->size[4] == 3, "grid_sampler(): expected grid to have size 3 in last dimension");
  for (int i = 2; i < 5; ++i)
    RECMV_REQUIRE(in->size[i] > 0, "grid_sampler(): expected input to have non-empty spatial dimensions");
  return RECMV_OK;
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int recmv_grid_sample3d_forward(const void* input, const recmv_tensor5* input_desc,
                                           const void* grid, const recmv_tensor5* grid_desc,
                                           void* output, const recmv_tensor5* output_desc, int interp,
                                           int pad, int dtype, void* stream) {
  int rc = check_common("grid_sample3d_forward", input_desc, grid_desc, interp, pad, dtype);
  if (rc) return rc;
  RECMV_REQUIRE(output_desc, "grid_sample3d_forward: NULL output descriptor");
  const Desc5 in = to_desc(input_desc), gr = to_desc(grid_desc), out = to_desc(output_desc);
  const int64_t count = gr.size[0] * gr.size[1] * gr.size[2] * gr.size[3];
  if (count == 0 || in.size[1] == 0) return RECMV_OK;
  RECMV_REQUIRE(input && grid && output, "grid_sample3d_forward: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  const int g = stream_grid(count, kBlk);
  if (dtype == RECMV_F32) {
    // record-coalesced lanes (the kernels' comment; profiles/r03_kernel_only_sampler_v3.txt), bit-identical to the kernels below
    bool handled = false;
    if (count >= 2048 && rec_ok(input, in, gr, dtype) && fits32(out)) {
      handled = true;
      const int G = rec_lanes(in.size[1], count, 0);
      const RecGeom q = rec_geom(in, gr, out);
      const int64_t blocks = ceil_div(ceil_div(count, (int64_t)(kBlk / G)), (int64_t)kNumXCD) * kNumXCD, cap = (int64_t)kNumCU * 16;
      const dim3 gdim((unsigned)(blocks < cap ? blocks : cap));
#define RECMV_FWD_REC(GG, VV)                                                                                        \
  hipLaunchKernelGGL((gs3d_fwd_rec_kernel<GG, VV>), gdim, dim3(kBlk), 0, s, (int)count, q, (const float*)input,         \
                     (const float*)grid, (float*)output)
      RECMV_REC_DISPATCH(in.size[1], G, RECMV_FWD_REC)
#undef RECMV_FWD_REC
    }
    if (handled) {
    } else if (vec4_ok(input, in, dtype))
      hipLaunchKernelGGL((gs3d_fwd_kernel<float, 4>), dim3(g), dim3(kBlk), 0, s, count,
                         (const float*)input, in, (const float*)grid, gr, (float*)output, out);
    else
      hipLaunchKernelGGL((gs3d_fwd_kernel<float, 1>), dim3(g), dim3(kBlk), 0, s, count,
                         (const float*)input, in, (const float*)grid, gr, (float*)output, out);
  } else {
    hipLaunchKernelGGL((gs3d_fwd_kernel<double, 1>), dim3(g), dim3(kBlk), 0, s, count,
                       (const double*)input, in, (const double*)grid, gr, (double*)output, out);
  }
  return check_launch("grid_sample3d_forward");
}

template <typename T, int VEC>
static void launch_bwd(int g, hipStream_t s, int64_t count, const void* input, const Desc5& in,
                       const void* grid, const Desc5& gr, const void* gout, const Desc5& go, void* ginp,
                       const Desc5& gi, void* ggrid) {
  if (ginp)
    hipLaunchKernelGGL((gs3d_bwd_kernel<T, VEC, true>), dim3(g), dim3(kBlk), 0, s, count,
                       (const T*)input, in, (const T*)grid, gr, (const T*)gout, go, (T*)ginp, gi,
                       (T*)ggrid);
  else
    hipLaunchKernelGGL((gs3d_bwd_kernel<T, VEC, false>), dim3(g), dim3(kBlk), 0, s, count,
                       (const T*)input, in, (const T*)grid, gr, (const T*)gout, go, (T*)ginp, gi,
                       (T*)ggrid);
}

extern "C" int recmv_grid_sample3d_backward(const void* input, const recmv_tensor5* input_desc,
                                            const void* grid, const recmv_tensor5* grid_desc,
                                            const void* grad_output,
                                            const recmv_tensor5* grad_output_desc, void* grad_input,
                                            const recmv_tensor5* grad_input_desc, void* grad_grid,
                                            int interp, int pad, int dtype, void* stream) {
  int rc = check_common("grid_sample3d_backward", input_desc, grid_desc, interp, pad, dtype);
  if (rc) return rc;
  RECMV_REQUIRE(grad_output_desc, "grid_sample3d_backward: NULL grad_output descriptor");
  RECMV_REQUIRE(!grad_input || grad_input_desc, "grid_sample3d_backward: grad_input without descriptor");
  const Desc5 in = to_desc(input_desc), gr = to_desc(grid_desc), go = to_desc(grad_output_desc);
  const Desc5 gi = grad_input ? to_desc(grad_input_desc) : in;
  const int64_t count = gr.size[0] * gr.size[1] * gr.size[2] * gr.size[3];
  if (count == 0) return RECMV_OK;
  RECMV_REQUIRE(input && grid && grad_output && grad_grid, "grid_sample3d_backward: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  const int g = stream_grid(count, kBlk);
  if (!grad_input && rec_ok(input, in, gr, dtype) && fits32(go)) {
    bool handled = true;
    const int G = rec_lanes(in.size[1], count, 1);
    const RecGeom q = rec_geom(in, gr, go);
    const int64_t blocks = ceil_div(ceil_div(count, (int64_t)(kBlk / G)), (int64_t)kNumXCD) * kNumXCD, cap = (int64_t)kNumCU * 16;
    const dim3 gdim((unsigned)(blocks < cap ? blocks : cap));      // (a multiple of the XCD count: see the kernels' tile order)
#define RECMV_BWD_REC(GG, VV)                                                                                        \
  hipLaunchKernelGGL((gs3d_bwd_rec_kernel<GG, VV>), gdim, dim3(kBlk), 0, s, (int)count, q, (const float*)input,         \
                     (const float*)grid, (const float*)grad_output, (float*)grad_grid)
    RECMV_REC_DISPATCH(in.size[1], G, RECMV_BWD_REC)
#undef RECMV_BWD_REC
    if (handled) return check_launch("grid_sample3d_backward");
  }
  if (dtype == RECMV_F32) {
    if (vec4_ok(input, in, dtype))
      launch_bwd<float, 4>(g, s, count, input, in, grid, gr, grad_output, go, grad_input, gi, grad_grid);
    else
      launch_bwd<float, 1>(g, s, count, input, in, grid, gr, grad_output, go, grad_input, gi, grad_grid);
  } else {
    launch_bwd<double, 1>(g, s, count, input, in, grid, gr, grad_output, go, grad_input, gi, grad_grid);
  }
  return check_launch("grid_sample3d_backward");
}

template <typename T, int VEC>
static void launch_dbwd(int g, hipStream_t s, int64_t count, const void* ggI, const Desc5& gI,
                        const void* ggG, const Desc5& gG, const void* input, const Desc5& in,
                        const void* grid, const Desc5& gr, const void* gout, const Desc5& go, void* ginp,
                        const Desc5& gi, void* ggrid, void* ggout, const Desc5& ggo) {
#define RECMV_DBWD(A, B)                                                                              \
  hipLaunchKernelGGL((gs3d_dbwd_kernel<T, VEC, A, B>), dim3(g), dim3(kBlk), 0, s, count, (const T*)ggI, \
                     gI, (const T*)ggG, gG, (const T*)input, in, (const T*)grid, gr, (const T*)gout, go, \
                     (T*)ginp, gi, (T*)ggrid, (T*)ggout, ggo)
  if (ggI && ginp)
    RECMV_DBWD(true, true);
  else if (ggI)
    RECMV_DBWD(true, false);
  else if (ginp)
    RECMV_DBWD(false, true);
  else
    RECMV_DBWD(false, false);
#undef RECMV_DBWD
}

extern "C" int recmv_grid_sample3d_dbackward(
    const void* ggI, const recmv_tensor5* ggI_desc, const void* ggG, const recmv_tensor5* ggG_desc,
    const void* input, const recmv_tensor5* input_desc, const void* grid, const recmv_tensor5* grid_desc,
    const void* grad_output, const recmv_tensor5* grad_output_desc, void* grad_input,
    const recmv_tensor5* grad_input_desc, void* grad_grid, void* grad_grad_output,
    const recmv_tensor5* grad_grad_output_desc, int interp, int pad, int dtype, void* stream) {
  int rc = check_common("grid_sample3d_dbackward", input_desc, grid_desc, interp, pad, dtype);
  if (rc) return rc;
  RECMV_REQUIRE(ggG_desc && grad_output_desc && grad_grad_output_desc,
                "grid_sample3d_dbackward: NULL descriptor");
  RECMV_REQUIRE(!ggI || ggI_desc, "grid_sample3d_dbackward: ggI without descriptor");
  RECMV_REQUIRE(!grad_input || grad_input_desc, "grid_sample3d_dbackward: grad_input without descriptor");
  const Desc5 in = to_desc(input_desc), gr = to_desc(grid_desc), go = to_desc(grad_output_desc);
  const Desc5 gG = to_desc(ggG_desc), ggo = to_desc(grad_grad_output_desc);
  const Desc5 gI = ggI ? to_desc(ggI_desc) : in;
  const Desc5 gi = grad_input ? to_desc(grad_input_desc) : in;
  const int64_t count = gr.size[0] * gr.size[1] * gr.size[2] * gr.size[3];
  if (count == 0) return RECMV_OK;
  RECMV_REQUIRE(ggG && input && grid && grad_output && grad_grid && grad_grad_output,
                "grid_sample3d_dbackward: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  const int g = stream_grid(count, kBlk);
  if (!ggI && !grad_input && rec_ok(input, in, gr, dtype) && fits32(go) && fits32(gG) && fits32(ggo)) {
    bool handled = true;
    const int G = rec_lanes(in.size[1], count, 2);
    RecGeom q = rec_geom(in, gr, go);
    q.gg_n = (int)gG.stride[0], q.gg_p = (int)gG.stride[3], q.gg_c = (int)gG.stride[4];
    q.o_n = (int)ggo.stride[0], q.o_c = (int)ggo.stride[1], q.o_p = (int)ggo.stride[4];
    const int64_t blocks = ceil_div(ceil_div(count, (int64_t)(kBlk / G)), (int64_t)kNumXCD) * kNumXCD, cap = (int64_t)kNumCU * 16;
    const dim3 gdim((unsigned)(blocks < cap ? blocks : cap));      // (a multiple of the XCD count: see the kernels' tile order)
#define RECMV_DBWD_REC(GG, VV)                                                                                       \
  hipLaunchKernelGGL((gs3d_dbwd_rec_kernel<GG, VV>), gdim, dim3(kBlk), 0, s, (int)count, q, (const float*)ggG,          \
                     (const float*)input, (const float*)grid, (const float*)grad_output, (float*)grad_grid,          \
                     (float*)grad_grad_output)
    RECMV_REC_DISPATCH(in.size[1], G, RECMV_DBWD_REC)
#undef RECMV_DBWD_REC
    if (handled) return check_launch("grid_sample3d_dbackward");
  }
  if (dtype == RECMV_F32) {
    if (vec4_ok(input, in, dtype) && (!ggI || vec4_ok(ggI, gI, dtype)))
      launch_dbwd<float, 4>(g, s, count, ggI, gI, ggG, gG, input, in, grid, gr, grad_output, go,
                            grad_input, gi, grad_grid, grad_grad_output, ggo);
    else
      launch_dbwd<float, 1>(g, s, count, ggI, gI, ggG, gG, input, in, grid, gr, grad_output, go,
                            grad_input, gi, grad_grid, grad_grad_output, ggo);
  } else {
    launch_dbwd<double, 1>(g, s, count, ggI, gI, ggG, gG, input, in, grid, gr, grad_output, go,
                           grad_input, gi, grad_grid, grad_grad_output, ggo);
  }
  return check_launch("grid_sample3d_dbackward");
}

extern "C" int recmv_set_sampler_mode(int mode) {
  const int prev = g_sampler_mode;
  if (mode >= 0 && mode <= 4) g_sampler_mode = mode;     // (2, 3, 4: forced lane splits, measurement only — tools/kernel_only.py)
  return prev;
}

extern "C" int recmv_get_sampler_mode(void) { return g_sampler_mode; }
