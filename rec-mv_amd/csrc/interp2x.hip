// 2x-1 trilinear upsample with boundary mask (forward) and its transpose (backward) — gfx950.
//
// Semantics follow MCAcc/cuda/interp2x_boundary3d_kernel.cu:11-239 of the reference: an output voxel
// with all-even coordinates copies its source; otherwise it is the mean of the 2/4/8 sources that
// bracket it along its odd axes, and is flagged "boundary" when those sources disagree on
// (v > balance).  Equivalent to F.interpolate(trilinear, align_corners=True) + (0 < valid < 1)
// (MCAcc/seg3d_lossless.py:273-282).
//
// Design: forward = one lane per source cell, lanes along x (8 coalesced row loads -> up to 8 outputs, see the
// kernel); backward = one lane per input voxel gathering its 27 output neighbours; both on the caller's stream
// (the reference launches 1024-thread blocks of single voxels on the default stream).  The sources
// are summed in the reference's order (v1..v8) so results match its CUDA build bit for bit.
//
// Algorithmic bytes: 4 n^3 in + 5 (2n-1)^3 out (forward); 4 (2n-1)^3 in + 4 n^3 out (backward).
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;
#pragma clang fp contract(off)

// One lane per SOURCE CELL (t, u, v): its 8 corner values c[z][y][x] are loaded once (coalesced along x) and produce
// the up to 8 outputs (2t+a, 2u+b, 2v+c), a, b, c in {0,1}, that lie inside the cell: no per-output index
// arithmetic, no parity divergence, every source value is fetched once per neighbouring cell instead of once per
// output, and the two x-parities of a lane are adjacent in memory.  The sums keep the reference's source order (x
// fastest, then y, then z; the skip_x / skip_y 4-point cases z fastest) and the mean is a multiplication by the
// exact power of two 1/2, 1/4, 1/8 — bit-identical to `(sum)/2.`, `/4.0`, `/8.0` in double + store rounding.
template <typename T>
__device__ __forceinline__ void put(T* __restrict__ out, uint8_t* __restrict__ bnd, int64_t o, T s, T scale,
                                    bool differ) {
  out[o] = s * scale;
  bnd[o] = differ ? 1 : 0;
}

// The two x-parities of a cell are adjacent in the output row: ONE 8-byte value store and ONE 2-byte flag store per (y, z)
// parity instead of two of each (the rows of a (2n-1)-wide volume start at any 4-byte / 1-byte offset, so the types below
// only claim that alignment; gfx950 global stores take it).  Half the store instructions and L2 write transactions.
template <typename T>
struct alignas(alignof(T)) Pair {
  T a, b;
};
struct alignas(1) FlagPair {
  uint8_t a, b;
};
template <typename T>
__device__ __forceinline__ void put2(T* __restrict__ out, uint8_t* __restrict__ bnd, int64_t o, T s0, bool d0, T s1, T scale1,
                                     bool d1) {
  Pair<T> v;
  v.a = s0;
  v.b = s1 * scale1;
  *reinterpret_cast<Pair<T>*>(out + o) = v;
  FlagPair f;
  f.a = d0 ? 1 : 0;
  f.b = d1 ? 1 : 0;
  *reinterpret_cast<FlagPair*>(bnd + o) = f;
}

// R consecutive cell rows (same z, y = u0 .. u0+R-1) per workgroup trip: their 2 (R+1) source rows are requested up front —
// neighbouring cell rows share a source row, and a lane has 4 (R+1) loads in flight instead of 8 (one cell row per trip was bound
// by the latency of its 8 loads: 26 us at 129^3 -> 257^3 whatever the store width, profiles/r03_kernel_only_interp_v1.txt).
template <typename T, int R>
__global__ __launch_bounds__(256) void interp2x_fwd_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                            uint8_t* __restrict__ bnd, int64_t groups, int hg,
                                                            int d, int h, int w, float balance) {
  const int D = 2 * d - 1, H = 2 * h - 1, W = 2 * w - 1;
  const T bal = (T)balance;
  for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int u0 = (int)(grp % hg) * R;                  // first cell row of the group
    const int t = (int)((grp / hg) % d);
    const int64_t b = grp / ((int64_t)hg * d);
    const bool hz = t < d - 1;                           // odd z outputs exist for this cell layer
    const int t1 = hz ? t + 1 : t;
    const T* src = in + b * (int64_t)d * h * w;
    const T* lo = src + (int64_t)t * h * w;              // source layer z-
    const T* hi = src + (int64_t)t1 * h * w;             // source layer z+
    for (int v = threadIdx.x; v < w; v += blockDim.x) {
      const bool hx = v < w - 1;
      const int v1 = hx ? v + 1 : v;
      T a[R + 1][2], c[R + 1][2];                        // [source row][x parity] of the two layers
#pragma unroll
      for (int r = 0; r <= R; ++r) {
        const int y = min(u0 + r, h - 1);
        a[r][0] = lo[(int64_t)y * w + v];
        a[r][1] = lo[(int64_t)y * w + v1];
        c[r][0] = hi[(int64_t)y * w + v];
        c[r][1] = hi[(int64_t)y * w + v1];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int u = u0 + r;
        if (u >= h) break;
        const bool hy = u < h - 1;                       // odd y outputs exist for this cell row
        const T c000 = a[r][0], c001 = a[r][1], c010 = a[r + 1][0], c011 = a[r + 1][1];
        const T c100 = c[r][0], c101 = c[r][1], c110 = c[r + 1][0], c111 = c[r + 1][1];
        const int64_t o00 = ((b * D + 2 * t) * H + 2 * u) * (int64_t)W;     // output rows (z, y) of this cell row
        const int64_t o01 = o00 + W, o10 = o00 + (int64_t)H * W, o11 = o10 + W;
        const bool f = c000 > bal;
        const bool g001 = (c001 > bal) != f, g010 = (c010 > bal) != f, g011 = (c011 > bal) != f;
        const bool g100 = (c100 > bal) != f, g101 = (c101 > bal) != f, g110 = (c110 > bal) != f;
        const bool g111 = (c111 > bal) != f;
        const int x = 2 * v;
        if (hx) {       // every cell but the last of a row: both x-parities, stored as pairs
          put2(out, bnd, o00 + x, c000, false, c000 + c001, (T)0.5, g001);
          if (hy) put2(out, bnd, o01 + x, (c000 + c010) * (T)0.5, g010, ((c000 + c001) + c010) + c011, (T)0.25, g001 || g010 || g011);
          if (hz) {
            put2(out, bnd, o10 + x, (c000 + c100) * (T)0.5, g100, ((c000 + c100) + c001) + c101, (T)0.25, g100 || g001 || g101);   // skip_y
            if (hy)
              put2(out, bnd, o11 + x, (((c000 + c100) + c010) + c110) * (T)0.25, g100 || g010 || g110,                       // skip_x
                   ((((((c000 + c001) + c010) + c011) + c100) + c101) + c110) + c111, (T)0.125,
                   g001 || g010 || g011 || g100 || g101 || g110 || g111);
          }
        } else {
          out[o00 + x] = c000;
          bnd[o00 + x] = 0;
          if (hy) put(out, bnd, o01 + x, c000 + c010, (T)0.5, g010);
          if (hz) {
            put(out, bnd, o10 + x, c000 + c100, (T)0.5, g100);
            if (hy) put(out, bnd, o11 + x, ((c000 + c100) + c010) + c110, (T)0.25, g100 || g010 || g110);   // skip_x
          }
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlk) void interp2x_bwd_kernel(const T* __restrict__ go, T* __restrict__ gi,
                                                            int64_t bc, int D, int H, int W) {
  const int d = (D + 1) / 2, h = (H + 1) / 2, w = (W + 1) / 2;
  const int64_t total = bc * d * h * w;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlk) {
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const int z = (int)((i / ((int64_t)h * w)) % d);
    const int64_t b = i / ((int64_t)d * h * w);
    const T* g = go + b * (int64_t)D * H * W;
    auto at = [&](int zz, int yy, int xx) { return g[((int64_t)zz * H + yy) * W + xx]; };
    const bool xm = x > 0, xp = x < w - 1, ym = y > 0, yp = y < h - 1, zm = z > 0, zp = z < d - 1;
    const int X = 2 * x, Y = 2 * y, Z = 2 * z;
    T grad = at(Z, Y, X);
    // 6 edge neighbours (weight 1/2), order of interp2x_boundary3d_kernel.cu:180-191
    if (xm) grad = (T)((double)grad + (double)at(Z, Y, X - 1) * 0.5);
    if (xp) grad = (T)((double)grad + (double)at(Z, Y, X + 1) * 0.5);
    if (ym) grad = (T)((double)grad + (double)at(Z, Y - 1, X) * 0.5);
    if (yp) grad = (T)((double)grad + (double)at(Z, Y + 1, X) * 0.5);
    if (zm) grad = (T)((double)grad + (double)at(Z - 1, Y, X) * 0.5);
    if (zp) grad = (T)((double)grad + (double)at(Z + 1, Y, X) * 0.5);
    // 12 face neighbours (weight 1/4): xy, xz, yz  (:194-219)
    if (xm && ym) grad = (T)((double)grad + (double)at(Z, Y - 1, X - 1) * 0.25);
    if (xp && ym) grad = (T)((double)grad + (double)at(Z, Y - 1, X + 1) * 0.25);
    if (xm && yp) grad = (T)((double)grad + (double)at(Z, Y + 1, X - 1) * 0.25);
    if (xp && yp) grad = (T)((double)grad + (double)at(Z, Y + 1, X + 1) * 0.25);
    if (xm && zm) grad = (T)((double)grad + (double)at(Z - 1, Y, X - 1) * 0.25);
    if (xp && zm) grad = (T)((double)grad + (double)at(Z - 1, Y, X + 1) * 0.25);
    if (xm && zp) grad = (T)((double)grad + (double)at(Z + 1, Y, X - 1) * 0.25);
    if (xp && zp) grad = (T)((double)grad + (double)at(Z + 1, Y, X + 1) * 0.25);
    if (ym && zm) grad = (T)((double)grad + (double)at(Z - 1, Y - 1, X) * 0.25);
    if (yp && zm) grad = (T)((double)grad + (double)at(Z - 1, Y + 1, X) * 0.25);
    if (ym && zp) grad = (T)((double)grad + (double)at(Z + 1, Y - 1, X) * 0.25);
    if (yp && zp) grad = (T)((double)grad + (double)at(Z + 1, Y + 1, X) * 0.25);
    // 8 corner neighbours (weight 1/8)  (:222-237)
    if (xm && ym && zm) grad = (T)((double)grad + (double)at(Z - 1, Y - 1, X - 1) * 0.125);
    if (xp && ym && zm) grad = (T)((double)grad + (double)at(Z - 1, Y - 1, X + 1) * 0.125);
    if (xm && yp && zm) grad = (T)((double)grad + (double)at(Z - 1, Y + 1, X - 1) * 0.125);
    if (xp && yp && zm) grad = (T)((double)grad + (double)at(Z - 1, Y + 1, X + 1) * 0.125);
    if (xm && ym && zp) grad = (T)((double)grad + (double)at(Z + 1, Y - 1, X - 1) * 0.125);
    if (xp && ym && zp) grad = (T)((double)grad + (double)at(Z + 1, Y - 1, X + 1) * 0.125);
    if (xm && yp && zp) grad = (T)((double)grad + (double)at(Z + 1, Y + 1, X - 1) * 0.125);
    if (xp && yp && zp) grad = (T)((double)grad + (double)at(Z + 1, Y + 1, X + 1) * 0.125);
    gi[i] = grad;
  }
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int recmv_interp2x_boundary3d_forward(const void* input, void* output, uint8_t* is_boundary,
                                                 int64_t bc, int64_t d, int64_t h, int64_t w,
                                                 float balance_value, int dtype, void* stream) {
  RECMV_REQUIRE(bc >= 0 && d >= 0 && h >= 0 && w >= 0, "interp2x_forward: negative size");
  if (bc == 0 || d == 0 || h == 0 || w == 0) return RECMV_OK;
  RECMV_REQUIRE(input && output && is_boundary, "interp2x_forward: NULL pointer");
  RECMV_REQUIRE(d < (1 << 30) && h < (1 << 30) && w < (1 << 30), "interp2x_forward: size too large");
  constexpr int R = 4;                                   // cell rows per workgroup trip
  const int hg = (int)ceil_div(h, (int64_t)R);
  const int64_t groups = bc * d * hg;
  hipStream_t s = (hipStream_t)stream;
  const int blk = (int)(w >= 256 ? 256 : ceil_div(w, kWave) * kWave);
  const unsigned g = (unsigned)(groups < (1ll << 30) ? groups : (1ll << 30));
  if (dtype == RECMV_F32)
    hipLaunchKernelGGL((interp2x_fwd_kernel<float, R>), dim3(g), dim3(blk), 0, s, (const float*)input,
                       (float*)output, is_boundary, groups, hg, (int)d, (int)h, (int)w, balance_value);
  else if (dtype == RECMV_F64)
    hipLaunchKernelGGL((interp2x_fwd_kernel<double, R>), dim3(g), dim3(blk), 0, s, (const double*)input,
                       (double*)output, is_boundary, groups, hg, (int)d, (int)h, (int)w, balance_value);
  else {
    set_error("interp2x_forward: dtype %d unsupported", dtype);
    return RECMV_ERR_UNSUPPORTED;
  }
  return check_launch("interp2x_forward");
}

extern "C" int recmv_interp2x_boundary3d_backward(const void* grad_output, void* grad_input, int64_t bc,
                                                  int64_t D, int64_t H, int64_t W, int dtype,
                                                  void* stream) {
  RECMV_REQUIRE(bc >= 0 && D >= 0 && H >= 0 && W >= 0, "interp2x_backward: negative size");
  if (bc == 0 || D == 0 || H == 0 || W == 0) return RECMV_OK;
  RECMV_REQUIRE(grad_output && grad_input, "interp2x_backward: NULL pointer");
  const int64_t total = bc * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2);
  hipStream_t s = (hipStream_t)stream;
  const int g = stream_grid(total, kBlk);
  if (dtype == RECMV_F32)
    hipLaunchKernelGGL(interp2x_bwd_kernel<float>, dim3(g), dim3(kBlk), 0, s, (const float*)grad_output,
                       (float*)grad_input, bc, (int)D, (int)H, (int)W);
  else if (dtype == RECMV_F64)
    hipLaunchKernelGGL(interp2x_bwd_kernel<double>, dim3(g), dim3(kBlk), 0, s, (const double*)grad_output,
                       (double*)grad_input, bc, (int)D, (int)H, (int)W);
  else {
    set_error("interp2x_backward: dtype %d unsupported", dtype);
    return RECMV_ERR_UNSUPPORTED;
  }
  return check_launch("interp2x_backward");
}
