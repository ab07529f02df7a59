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

// Forward for a channels-last f32 volume whose corner record is G float4 (C = 4 G channels): lane t of a 64 G-thread workgroup
// serves point t / G, channel group t % G, so the G lanes of a point read ONE corner record (16 G contiguous bytes, one or two
// cache lines) with one instruction — a quarter of the cache lines per instruction that the wave-per-channel-group mapping
// touches, which is what scattered points are bound by.  Every lane stores its own four channels (runs of 64 / G points per
// channel row; the L2 merges them).  Arithmetic per output element: the fma chain of gs3d_fwd_kernel (bit-identical).
template <int G>
__global__ __launch_bounds__(64 * G) void gs3d_fwd_rec_kernel(int64_t nthreads, const float* __restrict__ input, Desc5 in,
                                                             const float* __restrict__ grid, Desc5 gr,
                                                             float* __restrict__ output, Desc5 out) {
  const int64_t D = in.size[2], H = in.size[3], W = in.size[4];
  const int64_t oD = gr.size[1], oH = gr.size[2], oW = gr.size[3];
  const int t = threadIdx.x, pt = t / G, ch = (t - pt * G) * 4;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < nthreads; base += (int64_t)gridDim.x * 64) {
    const int64_t index = base + pt;
    if (index >= nthreads) continue;
    int64_t w, h, d, n;
    if (oD * oH == 1 && nthreads < (1ll << 31)) {      // a list of points per batch item: 32-bit index arithmetic
      const unsigned iw = (unsigned)index, uw = (unsigned)oW;
      n = iw / uw;
      w = iw - (unsigned)n * uw;
      h = d = 0;
    } else {
      w = index % oW, h = (index / oW) % oH, d = (index / (oH * oW)) % oD, n = index / (oD * oH * oW);
    }
    const float* g = grid + n * gr.stride[0] + d * gr.stride[1] + h * gr.stride[2] + w * gr.stride[3];
    const Cell<float> c = make_cell<float>(g[0], g[gr.stride[4]], g[2 * gr.stride[4]], W, H, D);
    const float* inp = input + n * in.stride[0] + ch;
    float4 val[8];
    float wgt[8];
    bool inb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      RECMV_CORNER_BITS(k);
      wgt[k] = c.fx[bx] * c.fy[by] * c.fz[bz];
      inb[k] = c.in_x[bx] && c.in_y[by] && c.in_z[bz];
      const int64_t off = (int64_t)(c.z0 + bz) * in.stride[2] + (int64_t)(c.y0 + by) * in.stride[3] +
                          (int64_t)(c.x0 + bx) * in.stride[4];
      if (inb[k]) val[k] = *reinterpret_cast<const float4*>(inp + off);
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (inb[k]) {
        a0 = fma(val[k].x, wgt[k], a0);
        a1 = fma(val[k].y, wgt[k], a1);
        a2 = fma(val[k].z, wgt[k], a2);
        a3 = fma(val[k].w, wgt[k], a3);
      }
    float* o = output + n * out.stride[0] + d * out.stride[2] + h * out.stride[3] + w * out.stride[4] + ch * out.stride[1];
    o[0] = a0;
    o[out.stride[1]] = a1;
    o[2 * out.stride[1]] = a2;
    o[3 * out.stride[1]] = a3;
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
    // record-coalesced lanes while the launch is latency-bound: at 153 k points 16.8 us on surface-coherent points and 28.6 us
    // on random ones, against 24 / 111 us for one lane per point; past ~3e5 points the one-lane-per-point kernel amortises the
    // per-point geometry better on coherent points (30.7 vs 39.5 us at 461 k) — though not on random ones (346 vs 85 us) —
    // profiles/r02_sampler_forward_ab.txt
    const int G = (int)(in.size[1] / 4);
    if (vec4_ok(input, in, dtype) && in.stride[4] == in.size[1] && count >= 4096 && count <= 300000 &&
        (G == 1 || G == 2 || G == 3 || G == 4 || G == 6 || G == 8)) {
      const int64_t blocks = ceil_div(count, (int64_t)64);
      const int64_t cap = (int64_t)kNumCU * 10;
      const dim3 gdim((unsigned)(blocks < cap ? blocks : cap));
#define RECMV_FWD_REC(GG)                                                                                          \
  hipLaunchKernelGGL(gs3d_fwd_rec_kernel<GG>, gdim, dim3(64 * GG), 0, s, count, (const float*)input, in, (const float*)grid, \
                     gr, (float*)output, out)
      switch (G) {
        case 1: RECMV_FWD_REC(1); break;
        case 2: RECMV_FWD_REC(2); break;
        case 3: RECMV_FWD_REC(3); break;
        case 4: RECMV_FWD_REC(4); break;
        case 6: RECMV_FWD_REC(6); break;
        default: RECMV_FWD_REC(8); break;
      }
#undef RECMV_FWD_REC
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
