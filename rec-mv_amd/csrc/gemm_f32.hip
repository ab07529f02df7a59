// f32 MFMA contractions for the three MLPs of the hot path (SDF, deformer, colour) — gfx950.
//
// The reference runs these as torch.nn.Linear -> cuBLAS sgemm plus separate bias / activation kernels
// (model/network.py:98-111, model/Deformer.py:194-199, model/RenderNet.py:83-94).  Here each layer is ONE
// kernel: v_mfma_f32_32x32x2_f32 (exact f32 multiply-accumulate, 157 TFLOP/s dense peak on MI355X) over
// LDS-staged 128x128x32 tiles, with bias + activation (+ the 1/sqrt(2) skip scale) fused in the epilogue
// so activations make one HBM round trip per layer.
//
//   gemm_nt : C[M,N] = act(A[M,K] . B[N,K]^T + bias) * out_scale      forward  (A = activations, B = W)
//                                                                     and dX = dZ . W^T^T (B = W^T copy)
//   gemm_tn : C[M,N] = A[K,M]^T . B[K,N]                              dW = dZ^T . X  (K = #points), split-K
//
// Tiling (wave64): 256 threads = 4 waves in 2x2, each wave owns a 64x64 sub-tile = 2x2 MFMA tiles
// (64 accumulator VGPRs).  gemm_nt keeps both operands k-contiguous in LDS (row stride 36 floats: the
// 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots) and feeds 4 MFMAs per 16-byte read;
// gemm_tn keeps them k-major and reads one dword per MFMA operand (conflict-free, 2 x 32-lane groups).
// Global->LDS goes through registers one K-tile ahead (one barrier per K-tile, 2 LDS buffers).
// Workgroup ids are remapped so the column tiles that share an A row-panel run on the same XCD/L2.
//
// FLOPs: 2*M*N*K.  With K=N=512 the arithmetic intensity is 128 FLOP/B >> 157e12/8e12, so every layer
// is MFMA-bound, not HBM-bound.
#include <mutex>
#include <unordered_map>
#include <type_traits>
#include <vector>
#include "common.h"
// RECMV_LIBM_SOFTPLUS (an experiment build of tools/trajectory_seeds.py, never the product's): the activation through the
// correctly-rounded-to-an-ulp library functions instead of the hardware exp2 / log2 units.
#ifdef RECMV_LIBM_SOFTPLUS
#define RECMV_EXPF(x) expf(x)
#define RECMV_LOG1PF(t) log1pf(t)
#else
#define RECMV_EXPF(x) __expf(x)
#define RECMV_LOG1PF(t) __logf(1.f + (t))
#endif

#include <algorithm>

namespace recmv {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- 3-way bf16 split of f32 operands (optional matrix mode "bf16x6") ------------------------------------------
// x = h + m + l exactly, each piece a bf16 (round-to-nearest at every step: |m| <= 2^-9 |x|, |l| <= 2^-17 |x|).
// A product x*y is then formed from the six piece products of weight >= 2^-18 (hh, hm, mh, hl, lh, mm) on the bf16
// matrix pipe (16x the f32 matrix rate) with f32 accumulation; the dropped products are <= 2^-25 relative, below
// f32 rounding.  Each piece product is exact in f32, so the result differs from the f32 MFMA only by the order of
// the f32 accumulation.
struct Pieces {
  bf16x8 h, m, l;
};
// Two f32 -> one packed pair of bf16, round to nearest even.  Default: the hardware conversion (v_cvt_pk_bf16_f32, new in gfx950).
// -DRECMV_SPLIT_INT: the same rounding in integer arithmetic (finite operands; the A/B build of tools/def_regu_stress.py).
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
#ifdef RECMV_SPLIT_INT
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u);
  ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
#else
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
#endif
}
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = pack_bf16(x0, x1);
  const f32x2 r = {x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xffff0000u)};
  m = pack_bf16(r.x, r.y);
  const f32x2 q = {r.x - __uint_as_float(m << 16), r.y - __uint_as_float(m & 0xffff0000u)};
  l = pack_bf16(q.x, q.y);
}
__device__ __forceinline__ Pieces split8(float4 a, float4 b) {
  unsigned h[4], m[4], l[4];
  split2(a.x, a.y, h[0], m[0], l[0]);
  split2(a.z, a.w, h[1], m[1], l[1]);
  split2(b.x, b.y, h[2], m[2], l[2]);
  split2(b.z, b.w, h[3], m[3], l[3]);
  Pieces p;
  p.h = __builtin_bit_cast(bf16x8, (u32x4){h[0], h[1], h[2], h[3]});
  p.m = __builtin_bit_cast(bf16x8, (u32x4){m[0], m[1], m[2], m[3]});
  p.l = __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]});
  return p;
}

constexpr int BM = 128, BN = 128, BK = 32;   // large tile (and the TN kernel's tile)
constexpr int kBlk = 256;
constexpr int LDK = BK + 4;    // NT: padded k stride (floats) of an LDS row
constexpr int LDM = BM + 4;    // TN: padded m stride (floats) of an LDS k-row

__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t nb) {
  const int64_t per = nb / kNumXCD;
  if (b >= per * kNumXCD) return b;
  return (b % kNumXCD) * per + b / kNumXCD;
}

// Activations of the fused epilogue.  ACT is a compile-time constant inside the epilogue loops.
//   softplus(beta): torch semantics (x*beta > 20 ? x : log1p(exp(x*beta))/beta), evaluated as
//   (max(zb,0) + log1p(exp(-|zb|))) / beta with the hardware exp2/log2 units: exp(-|zb|) = t in (0,1];
//   log1p(t) by its alternating series below 2^-6 (keeps the relative accuracy of tiny outputs, which the
//   backward recovers sigmoid(zb) = 1 - exp(-beta*y) from) and log(1+t) above.  |error| < 2e-7/beta.
template <int ACT>
__device__ __forceinline__ float apply_act(float z, float p, float inv_p) {
  if (ACT == RECMV_ACT_RELU) return z > 0.f ? z : 0.f;
  if (ACT == RECMV_ACT_SOFTPLUS) {
    const float zb = z * p;
    const float t = RECMV_EXPF(-fabsf(zb));
    const float series = t * (1.f - t * (0.5f - t * (0.33333334f - 0.25f * t)));
    const float l = t < 0.015625f ? series : RECMV_LOG1PF(t);
    const float y = (fmaxf(zb, 0.f) + l) * inv_p;
    return zb > 20.f ? z : y;
  }
  if (ACT == RECMV_ACT_TANH) return tanhf(z);
  return z;
}

// act'(z) expressed through y = act(z) (elementwise.hip has the same formulas)
__device__ __forceinline__ float dact_y(float y, int act, float p) {
  switch (act) {
    case RECMV_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case RECMV_ACT_SOFTPLUS: {
      // sigmoid(beta z) = 1 - exp(-beta y).  This runs on the operand-staging path of an MFMA kernel, so it uses
      // the hardware exp2 unit; below t = 1/64 the alternating series keeps the relative accuracy expm1 would give.
      const float t = p * y;
      const float series = t * (1.f - t * (0.5f - t * (0.16666667f - 0.041666668f * t)));
      return t < 0.015625f ? series : 1.f - RECMV_EXPF(-t);
    }
    case RECMV_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// Optional transform of the A operand while it is staged: A_eff = A (.) act'(y_scale * Y) * a_scale — the
// dZ = dY (.) act'(Z) step of a layer's backward fused into the dX = dZ W product (no dZ round trip through HBM).
struct AMul {
  const float* Y;
  int64_t ldy;
  int act;
  float param, y_scale, a_scale;
  // row-segmented weights (recmv_gemm_nt_seg): tiles whose first row is >= split multiply by B2 (+ bias2) instead of B (+ bias) —
  // two nets of one shape over one concatenated row block in one launch; split is a multiple of every tile height (128)
  const float* B2;
  const float* bias2;
  int split;
};

__device__ __forceinline__ float4 amul4(float4 a, float4 y, const AMul& m) {
  a.x *= dact_y(y.x * m.y_scale, m.act, m.param) * m.a_scale;
  a.y *= dact_y(y.y * m.y_scale, m.act, m.param) * m.a_scale;
  a.z *= dact_y(y.z * m.y_scale, m.act, m.param) * m.a_scale;
  a.w *= dact_y(y.w * m.y_scale, m.act, m.param) * m.a_scale;
  return a;
}

// 4 consecutive floats of a row, zero-filled past `limit` (elements left in the row).
__device__ __forceinline__ float4 load4_guard(const float* __restrict__ p, int64_t limit, bool vec_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (limit >= 4 && vec_ok) {
    v = *reinterpret_cast<const float4*>(p);
  } else {
    if (limit > 0) v.x = p[0];
    if (limit > 1) v.y = p[1];
    if (limit > 2) v.z = p[2];
    if (limit > 3) v.w = p[3];
  }
  return v;
}

// ------------------------------------------------------------------------------------------ NT
// T = MFMA tiles per wave along each dimension: T=2 -> 128x128 workgroup tile (large M), T=1 -> 64x64
// (small M: the ray path launches 3k-6k rows, 128-tiles would leave most of the 256 CUs idle).
// FAST: both operands 16-byte aligned with row strides and K multiples of 4 -> every staging load is one
// unconditional-address global_load_dwordx4 (rows clamped to the last valid row, k tail zero-filled per
// float4); otherwise the element-guarded loader.
//
// Epilogue: the accumulators go through LDS (the operand buffers are dead by then) and are written out by a
// compact loop — each thread owns a fixed 4-column strip, adds its bias float4, applies the activation and
// issues 16-byte row-contiguous stores.  (A fully unrolled per-accumulator-register epilogue costs 64 copies
// of the activation per thread: more instruction bytes than the instruction cache holds, and 4-byte stores.)
// Optional OUTPUT transform of the kernels without the operand transform: C (.)= act'(y_scale * Y[row, col]) * a_scale —
// the dZ = dY (.) act'(Z) step of the NEXT layer of a backward chain applied where dY is produced (each element once;
// the operand transform above recomputes it in every column tile that reads the element).
__device__ __forceinline__ float4 emul4(float4 v, const AMul& m, int gm, int gn, int N) {
  const float* y = m.Y + (int64_t)gm * m.ldy + gn;
  float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (gn + 4 <= N && (m.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(m.Y) & 15) == 0) {
    yv = *reinterpret_cast<const float4*>(y);
  } else {
    yv.x = y[0];
    if (gn + 1 < N) yv.y = y[1];
    if (gn + 2 < N) yv.z = y[2];
    if (gn + 3 < N) yv.w = y[3];
  }
  v.x *= dact_y(yv.x * m.y_scale, m.act, m.param) * m.a_scale;
  v.y *= dact_y(yv.y * m.y_scale, m.act, m.param) * m.a_scale;
  v.z *= dact_y(yv.z * m.y_scale, m.act, m.param) * m.a_scale;
  v.w *= dact_y(yv.w * m.y_scale, m.act, m.param) * m.a_scale;
  return v;
}

template <int TBM, int TBN, int ACT>
__device__ __forceinline__ void nt_epilogue_rows(const float* __restrict__ Cs, const float* __restrict__ bias,
                                                 float* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0,
                                                 float act_param, float out_scale, bool c_vec, const AMul& em) {
  constexpr int LDC = TBN + 4;
  constexpr int C4 = TBN / 4;                 // float4 strips per tile row (32 or 16): divides kBlk
  constexpr int ROWS_PER_PASS = kBlk / C4;    // 8 or 16
  const int tid = threadIdx.x;
  const int c4 = tid % C4, r0 = tid / C4;
  const int gn = n0 + c4 * 4;
  if (gn >= N) return;
  const float inv_p = act_param != 0.f ? 1.f / act_param : 0.f;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) {
    bv.x = bias[gn];
    if (gn + 1 < N) bv.y = bias[gn + 1];
    if (gn + 2 < N) bv.z = bias[gn + 2];
    if (gn + 3 < N) bv.w = bias[gn + 3];
  }
  const bool full4 = c_vec && gn + 4 <= N;
#pragma unroll 2
  for (int row = r0; row < TBM; row += ROWS_PER_PASS) {
    const int gm = m0 + row;
    if (gm >= M) break;
    float4 v = *reinterpret_cast<const float4*>(Cs + row * LDC + c4 * 4);
    v.x = apply_act<ACT>(v.x + bv.x, act_param, inv_p) * out_scale;
    v.y = apply_act<ACT>(v.y + bv.y, act_param, inv_p) * out_scale;
    v.z = apply_act<ACT>(v.z + bv.z, act_param, inv_p) * out_scale;
    v.w = apply_act<ACT>(v.w + bv.w, act_param, inv_p) * out_scale;
    if (em.Y) v = emul4(v, em, gm, gn, N);
    float* dst = C + (int64_t)gm * ldc + gn;
    if (full4) {
      *reinterpret_cast<float4*>(dst) = v;
    } else {
      dst[0] = v.x;
      if (gn + 1 < N) dst[1] = v.y;
      if (gn + 2 < N) dst[2] = v.z;
      if (gn + 3 < N) dst[3] = v.w;
    }
  }
}

template <int T, int ACT>
__device__ __forceinline__ void nt_epilogue(const float* __restrict__ Cs, const float* __restrict__ bias,
                                            float* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0,
                                            float act_param, float out_scale, bool c_vec, const AMul& em) {
  nt_epilogue_rows<64 * T, 64 * T, ACT>(Cs, bias, C, ldc, M, N, m0, n0, act_param, out_scale, c_vec, em);
}

template <int T, bool FAST, bool AMUL, bool BF3>
__global__ __launch_bounds__(kBlk) void gemm_nt_kernel(const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ B, int64_t ldb,
                                                       const float* __restrict__ bias, float* __restrict__ C,
                                                       int64_t ldc, int M, int N, int K, int act,
                                                       float act_param, float out_scale, int nbm, int nbn,
                                                       bool a_vec, bool b_vec, bool c_vec, AMul am) {
  constexpr int TBM = 64 * T, TBN = 64 * T;     // workgroup tile
  constexpr int WT = 32 * T;                    // wave tile edge
  constexpr int NLD = TBM * 8 / kBlk;           // float4 per thread per operand tile (4 or 2)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [2][TBM][LDK]
  float* Bs = smem + 2 * TBM * LDK;        // [2][TBN][LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;      // 2x2 waves
  const int64_t logical = xcd_remap(blockIdx.x, (int64_t)nbm * nbn);
  const int tile_m = (int)(logical / nbn), tile_n = (int)(logical % nbn);
  const int m0 = tile_m * TBM, n0 = tile_n * TBN;
  if (am.B2 && m0 >= am.split) {              // second weight set for the rows of the second net (see AMul)
    B = am.B2;
    bias = am.bias2;
  }

  f32x16 acc[T][T];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging map: TBM*8 float4 per operand tile; row = idx/8, c4 = idx%8
  float4 ra[NLD], rb[NLD];
  const float* pa[NLD];
  const float* pb[NLD];
  const float* py[NLD];
  if (FAST) {
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int idx = tid + kBlk * r;
      const int row = idx >> 3, c4 = idx & 7;
      int gm = m0 + row, gn = n0 + row;
      gm = gm < M ? gm : M - 1;                 // clamped rows are computed and never stored
      gn = gn < N ? gn : N - 1;
      pa[r] = A + (int64_t)gm * lda + c4 * 4;
      pb[r] = B + (int64_t)gn * ldb + c4 * 4;
      if (AMUL) py[r] = am.Y + (int64_t)gm * am.ldy + c4 * 4;
    }
  }
  auto gload = [&](int k0) {
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int idx = tid + kBlk * r;
      const int row = idx >> 3, c4 = idx & 7;
      const int k = k0 + c4 * 4;
      if (FAST) {
        const bool in = k < K;                  // K % 4 == 0: a float4 is entirely inside or outside
        ra[r] = in ? *reinterpret_cast<const float4*>(pa[r] + k0) : make_float4(0, 0, 0, 0);
        rb[r] = in ? *reinterpret_cast<const float4*>(pb[r] + k0) : make_float4(0, 0, 0, 0);
        if (AMUL && in) ra[r] = amul4(ra[r], *reinterpret_cast<const float4*>(py[r] + k0), am);
      } else {
        const int gm = m0 + row, gn = n0 + row;
        ra[r] = (gm < M) ? load4_guard(A + (int64_t)gm * lda + k, K - k, a_vec) : make_float4(0, 0, 0, 0);
        rb[r] = (gn < N) ? load4_guard(B + (int64_t)gn * ldb + k, K - k, b_vec) : make_float4(0, 0, 0, 0);
        if (AMUL && gm < M) ra[r] = amul4(ra[r], load4_guard(am.Y + (int64_t)gm * am.ldy + k, K - k, false), am);
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int idx = tid + kBlk * r;
      const int row = idx >> 3, c4 = idx & 7;
      *reinterpret_cast<float4*>(As + (buf * TBM + row) * LDK + c4 * 4) = ra[r];
      *reinterpret_cast<float4*>(Bs + (buf * TBN + row) * LDK + c4 * 4) = rb[r];
    }
  };

  const int nk = (K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int arow = wm * WT + (lane & 31), brow = wn * WT + (lane & 31), khalf = (lane >> 5) * 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    if (BF3) {
      // bf16x6: per 16-column step every lane converts its 8 consecutive k of each operand row into three bf16
      // pieces (registers only; LDS still holds f32) and issues six bf16 MFMAs per 32x32 tile, small terms first
      const float* as = As + (buf * TBM + arow) * LDK + 2 * khalf;
      const float* bs = Bs + (buf * TBN + brow) * LDK + 2 * khalf;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        Pieces pa[T], pb[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {
          const float* ap = as + i * 32 * LDK + ks * 16;
          const float* bp = bs + i * 32 * LDK + ks * 16;
          pa[i] = split8(*reinterpret_cast<const float4*>(ap), *reinterpret_cast<const float4*>(ap + 4));
          pb[i] = split8(*reinterpret_cast<const float4*>(bp), *reinterpret_cast<const float4*>(bp + 4));
        }
#pragma unroll
        for (int mi = 0; mi < T; ++mi)
#pragma unroll
          for (int ni = 0; ni < T; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].h, pb[ni].l, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].l, pb[ni].h, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].m, pb[ni].m, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].h, pb[ni].m, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].m, pb[ni].h, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].h, pb[ni].h, acc[mi][ni], 0, 0, 0);
          }
      }
    } else {
    const float* as = As + (buf * TBM + arow) * LDK + khalf;
    const float* bs = Bs + (buf * TBN + brow) * LDK + khalf;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 a[T], b[T];
#pragma unroll
      for (int i = 0; i < T; ++i) {
        a[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDK + kk * 8);
        b[i] = *reinterpret_cast<const float4*>(bs + i * 32 * LDK + kk * 8);
      }
#pragma unroll
      for (int mi = 0; mi < T; ++mi)
#pragma unroll
        for (int ni = 0; ni < T; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // accumulators -> LDS: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  constexpr int LDC = TBN + 4;
  float* Cs = smem;                        // [TBM][LDC] <= the operand buffers
#pragma unroll
  for (int mi = 0; mi < T; ++mi)
#pragma unroll
    for (int ni = 0; ni < T; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WT + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wn * WT + ni * 32 + (lane & 31);
        Cs[row * LDC + col] = acc[mi][ni][r];
      }
  __syncthreads();
  switch (act) {
    case RECMV_ACT_RELU:
      nt_epilogue<T, RECMV_ACT_RELU>(Cs, bias, C, ldc, M, N, m0, n0, act_param, out_scale, c_vec, AMUL ? AMul{nullptr, 0, 0, 0.f, 1.f, 1.f} : am);
      break;
    case RECMV_ACT_SOFTPLUS:
      nt_epilogue<T, RECMV_ACT_SOFTPLUS>(Cs, bias, C, ldc, M, N, m0, n0, act_param, out_scale, c_vec, AMUL ? AMul{nullptr, 0, 0, 0.f, 1.f, 1.f} : am);
      break;
    case RECMV_ACT_TANH:
      nt_epilogue<T, RECMV_ACT_TANH>(Cs, bias, C, ldc, M, N, m0, n0, act_param, out_scale, c_vec, AMUL ? AMul{nullptr, 0, 0, 0.f, 1.f, 1.f} : am);
      break;
    default:
      nt_epilogue<T, RECMV_ACT_NONE>(Cs, bias, C, ldc, M, N, m0, n0, act_param, out_scale, c_vec, AMUL ? AMul{nullptr, 0, 0, 0.f, 1.f, 1.f} : am);
  }
}

// ------------------------------------------------------------------------------------------ NT 128x128, three workgroups per CU
// The 128x128 f32 kernel above keeps two K-tiles of both operands in LDS (72 KB): two workgroups per CU, i.e. two waves per SIMD, and
// with K = 512 a workgroup lives for only 16 K-tiles — its prologue (first operand tile from HBM) and its epilogue (accumulators ->
// LDS -> activation -> HBM) leave its SIMD partner alone with the matrix pipe for ~15 % of its life, and one wave alone does not
// keep the pipe busy across its own barriers.  This variant holds LESS in LDS so that THREE workgroups fit a CU (registers allow
// exactly three: 94 + 64 accumulators): either one K-tile of 32 columns (SINGLE: the next tile waits in registers, two barriers per
// tile) or two K-tiles of 16 columns; the epilogue goes through LDS in two halves of 64 rows.  Same products in the same order
// as gemm_nt_kernel<2, true, ...>: bit-identical results.
template <bool AMUL, int BKT, bool SINGLE, int MI, int NI>
__global__ __launch_bounds__(kBlk, (MI * NI == 4 ? (BKT <= 16 ? 4 : 3) : 5))
void gemm_nt_occ_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                        const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int M, int N, int K, int act,
                        float act_param, float out_scale, int nbm, int nbn, bool c_vec, AMul am) {
  constexpr int TBM = 64 * MI, TBN = 64 * NI;   // workgroup tile; 2 x 2 waves of (32 MI) x (32 NI)
  constexpr int LDKT = BKT + 4;                 // padded k stride (floats) of an LDS row
  constexpr int C4R = BKT / 4;                  // float4 per operand row and K-tile
  constexpr int NLA = TBM * C4R / kBlk, NLB = TBN * C4R / kBlk;   // float4 per thread per operand tile
  constexpr int NBUF = SINGLE ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                             // [NBUF][TBM][LDKT]
  float* Bs = smem + NBUF * TBM * LDKT;         // [NBUF][TBN][LDKT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t logical = xcd_remap(blockIdx.x, (int64_t)nbm * nbn);
  const int tile_m = (int)(logical / nbn), tile_n = (int)(logical % nbn);
  const int m0 = tile_m * TBM, n0 = tile_n * TBN;
  if (am.B2 && m0 >= am.split) {
    B = am.B2;
    bias = am.bias2;
  }
  f32x16 acc[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[NLA], rb[NLB];
  const float* pa[NLA];
  const float* pb[NLB];
  const float* py[NLA];
#pragma unroll
  for (int r = 0; r < NLA; ++r) {
    const int idx = tid + kBlk * r;
    int gm = m0 + idx / C4R;
    gm = gm < M ? gm : M - 1;                   // clamped rows are computed and never stored
    pa[r] = A + (int64_t)gm * lda + (idx % C4R) * 4;
    if (AMUL) py[r] = am.Y + (int64_t)gm * am.ldy + (idx % C4R) * 4;
  }
#pragma unroll
  for (int r = 0; r < NLB; ++r) {
    const int idx = tid + kBlk * r;
    int gn = n0 + idx / C4R;
    gn = gn < N ? gn : N - 1;
    pb[r] = B + (int64_t)gn * ldb + (idx % C4R) * 4;
  }
  auto gload = [&](int k0) {
#pragma unroll
    for (int r = 0; r < NLA; ++r) {
      const bool in = k0 + ((tid + kBlk * r) % C4R) * 4 < K;       // K % 4 == 0: a float4 is entirely inside or outside
      ra[r] = in ? *reinterpret_cast<const float4*>(pa[r] + k0) : make_float4(0, 0, 0, 0);
      if (AMUL && in) ra[r] = amul4(ra[r], *reinterpret_cast<const float4*>(py[r] + k0), am);
    }
#pragma unroll
    for (int r = 0; r < NLB; ++r) {
      const bool in = k0 + ((tid + kBlk * r) % C4R) * 4 < K;
      rb[r] = in ? *reinterpret_cast<const float4*>(pb[r] + k0) : make_float4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < NLA; ++r) {
      const int idx = tid + kBlk * r;
      *reinterpret_cast<float4*>(As + (buf * TBM + idx / C4R) * LDKT + (idx % C4R) * 4) = ra[r];
    }
#pragma unroll
    for (int r = 0; r < NLB; ++r) {
      const int idx = tid + kBlk * r;
      *reinterpret_cast<float4*>(Bs + (buf * TBN + idx / C4R) * LDKT + (idx % C4R) * 4) = rb[r];
    }
  };

  const int nk = (K + BKT - 1) / BKT;
  gload(0);
  lstore(0);
  __syncthreads();
  const int arow = wm * 32 * MI + (lane & 31), brow = wn * 32 * NI + (lane & 31), khalf = (lane >> 5) * 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = SINGLE ? 0 : (kt & 1);
    if (kt + 1 < nk) gload((kt + 1) * BKT);
    const float* as = As + (buf * TBM + arow) * LDKT + khalf;
    const float* bs = Bs + (buf * TBN + brow) * LDKT + khalf;
#pragma unroll
    for (int kk = 0; kk < BKT / 8; ++kk) {
      float4 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDKT + kk * 8);
#pragma unroll
      for (int i = 0; i < NI; ++i) b[i] = *reinterpret_cast<const float4*>(bs + i * 32 * LDKT + kk * 8);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
    if (SINGLE) __syncthreads();                 // every wave has read the tile: the buffer may be overwritten
    if (kt + 1 < nk) lstore(SINGLE ? 0 : (buf ^ 1));
    __syncthreads();
  }

  // accumulators -> LDS -> HBM in two passes: pass h takes the rows of the waves with wm == h (the half-tile fits the LDS the
  // occupancy allows)
  constexpr int HR = 32 * MI, LDC = TBN + 4;
  float* Cs = smem;                              // [HR][LDC]
  const AMul em = AMUL ? AMul{nullptr, 0, 0, 0.f, 1.f, 1.f} : am;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
    if (wm == h) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = wn * 32 * NI + ni * 32 + (lane & 31);
            Cs[row * LDC + col] = acc[mi][ni][r];
          }
    }
    __syncthreads();
    const int mh = m0 + HR * h;
    switch (act) {
      case RECMV_ACT_RELU:
        nt_epilogue_rows<HR, TBN, RECMV_ACT_RELU>(Cs, bias, C, ldc, M, N, mh, n0, act_param, out_scale, c_vec, em);
        break;
      case RECMV_ACT_SOFTPLUS:
        nt_epilogue_rows<HR, TBN, RECMV_ACT_SOFTPLUS>(Cs, bias, C, ldc, M, N, mh, n0, act_param, out_scale, c_vec, em);
        break;
      case RECMV_ACT_TANH:
        nt_epilogue_rows<HR, TBN, RECMV_ACT_TANH>(Cs, bias, C, ldc, M, N, mh, n0, act_param, out_scale, c_vec, em);
        break;
      default:
        nt_epilogue_rows<HR, TBN, RECMV_ACT_NONE>(Cs, bias, C, ldc, M, N, mh, n0, act_param, out_scale, c_vec, em);
    }
  }
}

// ------------------------------------------------------------------------------------------ NT, bf16x6 staged
// The bf16x6 mode with the 3-way split done ONCE per operand element, on the global -> LDS staging path (8 elements
// per thread and item: two 16-byte loads, 44 VALU, three 16-byte LDS stores), instead of once per wave that reads the
// element as an MFMA fragment: the main loop is then 16-byte fragment reads + bf16 MFMAs, with the split's VALU issued
// in the shadow of the MFMAs (about four per MFMA).
// LDS image of an operand tile: three planes (h, m, l) of [rows][32 k] bf16, row = 64 bytes = four 16-byte chunks,
// chunk c of row r stored at chunk position c ^ ((r >> 2) & 3): the four 16-lane groups of a ds_read_b128 fragment
// read (rows {0-3,12-15,20-27}+.. at one chunk) and the 8-lane groups of the ds_write_b128 stores (two rows x four
// chunks) each cover distinct 16-byte bank slots, without padding.  One LDS buffer (48 KB of operands for 128x128; two
// workgroups per CU, each covering the other's barriers and epilogue).
// Pipeline per K-tile kt (raw = f32 as loaded, pieces = packed bf16 planes, both in registers): the raw registers hold
// tile kt+1 (requested during the previous iteration); it is split while the first 16 columns of tile kt are multiplied
// (each MFMA followed by its share of the split's VALU), tile kt+2 is requested as soon as the raw registers are free,
// the second 16 columns are multiplied, then barrier, pieces -> LDS, barrier.
// Measured alternatives that were slower (profiles/r02_gemm_staged_split.txt): two raw A sets with the A request a whole
// K-tile ahead (-6 %), accumulators taking turns between consecutive MFMAs (-5 %): the loop is bound by the SIMD's issue
// slots (about 6 non-MFMA instructions per MFMA across the two co-resident waves), not by latency.
__device__ __forceinline__ int b3_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// PRE: the B operand (a weight matrix, re-read by every row tile of every launch of an optimiser step) arrives already split —
// three bf16 planes [3][rows][Kp] (Kp = K rounded up to a K-tile, zero-padded) written once per weight version by
// recmv_b3_split with the SAME split8 as the staging path, so the pieces, the products and their order are those of the in-loop
// split (bit-identical results).  The B half of the split's VALU (44 per 8 elements) leaves the loop, which is bound by its issue
// slots; a thread's three 16-byte plane chunks of K-tile kt + 2 are requested behind the LDS stores of tile kt + 1 and stored one
// K-tile later.
struct B3Pre {
  const char* planes;        // first weight set
  const char* planes2;       // second weight set (row-segmented launches), or NULL
  int64_t plane_bytes, plane_bytes2;
  int Kp;
};

template <int T, bool AMUL, bool PRE>
__global__ __launch_bounds__(kBlk, 2) void gemm_nt_b3_kernel(const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ B, int64_t ldb,
                                                             const float* __restrict__ bias, float* __restrict__ C,
                                                             int64_t ldc, int M, int N, int K, int act,
                                                             float act_param, float out_scale, int nbm, int nbn,
                                                             bool c_vec, AMul am, B3Pre pre) {
  constexpr int TBM = 64 * T, TBN = 64 * T, WT = 32 * T;
  constexpr int NI = TBM * 4 / kBlk;            // (row, 8-k chunk) items per thread and operand: 2 or 1
  constexpr int PLANE = TBM * 64;               // bytes of one plane of one operand tile
  constexpr int NSET = 1;                       // raw A register sets
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* As = reinterpret_cast<char*>(smem);     // [3][TBM][64 B]
  char* Bs = As + 3 * PLANE;                    // [3][TBN][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t logical = xcd_remap(blockIdx.x, (int64_t)nbm * nbn);
  const int tile_m = (int)(logical / nbn), tile_n = (int)(logical % nbn);
  const int m0 = tile_m * TBM, n0 = tile_n * TBN;
  const char* planes = pre.planes;
  int64_t plane_bytes = pre.plane_bytes;
  if (am.B2 && m0 >= am.split) {              // second weight set for the rows of the second net (see AMul)
    B = am.B2;
    bias = am.bias2;
    planes = pre.planes2;
    plane_bytes = pre.plane_bytes2;
  }

  f32x16 acc[T][T];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[NSET][NI][2], rb[NI][2], ry[AMUL ? NI : 1][2];
  u32x4 sa[NI][3], sb[NI][3];
  const float* pa[NI];
  const float* pb[NI];
  const char* pq[NI];
  const float* py[NI];
  int soff[NI];
#pragma unroll
  for (int r = 0; r < NI; ++r) {
    const int item = tid + kBlk * r;
    const int row = item >> 2, ch = item & 3;
    int gm = m0 + row, gn = n0 + row;
    gm = gm < M ? gm : M - 1;                   // clamped rows are computed and never stored
    gn = gn < N ? gn : N - 1;
    pa[r] = A + (int64_t)gm * lda + ch * 8;
    pb[r] = B + (int64_t)gn * ldb + ch * 8;
    if (PRE) pq[r] = planes + ((int64_t)gn * pre.Kp + ch * 8) * 2;
    if (AMUL) py[r] = am.Y + (int64_t)gm * am.ldy + ch * 8;
    soff[r] = b3_off(row, ch);
  }
  const int kch = (tid & 3) * 8;
  const int nk = (K + BK - 1) / BK;
  // Offsets of a thread's two float4 of K-tile t, and whether they are inside K (K % 4 == 0: a float4 is entirely
  // inside or outside; an outside one reads the row's last float4 and is zeroed).  Whole tiles take the plain path.
#define RECMV_KEEP4(v, c) v = make_float4((c) ? v.x : 0.f, (c) ? v.y : 0.f, (c) ? v.z : 0.f, (c) ? v.w : 0.f)
  auto load_a = [&](auto set_c, auto whole_c, int t) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    const int k0 = t * BK, k = k0 + kch;
    const bool in0 = decltype(whole_c)::value || k < K, in1 = decltype(whole_c)::value || k + 4 < K;
    const int o0 = in0 ? k0 : K - 4 - kch, o1 = in1 ? k0 + 4 : K - 4 - kch;
#pragma unroll
    for (int r = 0; r < NI; ++r) {
      ra[S][r][0] = *reinterpret_cast<const float4*>(pa[r] + o0);
      ra[S][r][1] = *reinterpret_cast<const float4*>(pa[r] + o1);
      if (AMUL) {
        ry[r][0] = *reinterpret_cast<const float4*>(py[r] + o0);
        ry[r][1] = *reinterpret_cast<const float4*>(py[r] + o1);
      }
      if (!decltype(whole_c)::value) {
        RECMV_KEEP4(ra[S][r][0], in0);
        RECMV_KEEP4(ra[S][r][1], in1);
      }
    }
  };
  auto load_b = [&](auto whole_c, int t) __attribute__((always_inline)) {
    const int k0 = t * BK, k = k0 + kch;
    const bool in0 = decltype(whole_c)::value || k < K, in1 = decltype(whole_c)::value || k + 4 < K;
    const int o0 = in0 ? k0 : K - 4 - kch, o1 = in1 ? k0 + 4 : K - 4 - kch;
#pragma unroll
    for (int r = 0; r < NI; ++r) {
      rb[r][0] = *reinterpret_cast<const float4*>(pb[r] + o0);
      rb[r][1] = *reinterpret_cast<const float4*>(pb[r] + o1);
      if (!decltype(whole_c)::value) {
        RECMV_KEEP4(rb[r][0], in0);
        RECMV_KEEP4(rb[r][1], in1);
      }
    }
  };
#undef RECMV_KEEP4
  auto load_b_pre = [&](int t) __attribute__((always_inline)) {      // pieces of K-tile t straight into the store registers
#pragma unroll
    for (int r = 0; r < NI; ++r)
#pragma unroll
      for (int p = 0; p < 3; ++p) sb[r][p] = *reinterpret_cast<const u32x4*>(pq[r] + p * plane_bytes + (int64_t)t * (BK * 2));
  };
  auto split_a = [&](auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int r = 0; r < NI; ++r) {
      if (AMUL) {
        ra[S][r][0] = amul4(ra[S][r][0], ry[r][0], am);
        ra[S][r][1] = amul4(ra[S][r][1], ry[r][1], am);
      }
      const Pieces a = split8(ra[S][r][0], ra[S][r][1]);
      sa[r][0] = __builtin_bit_cast(u32x4, a.h);
      sa[r][1] = __builtin_bit_cast(u32x4, a.m);
      sa[r][2] = __builtin_bit_cast(u32x4, a.l);
    }
  };
  // keeps everything computed from `v` behind this point of the instruction stream (the split is pure arithmetic: without
  // it the compiler may start it ahead of the scheduling barrier, right behind the loads it waits for)
  auto pin4 = [&](float4& v) __attribute__((always_inline)) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
  };
  auto split_b = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < NI; ++r) {
      pin4(rb[r][0]);
      pin4(rb[r][1]);
      const Pieces b = split8(rb[r][0], rb[r][1]);
      sb[r][0] = __builtin_bit_cast(u32x4, b.h);
      sb[r][1] = __builtin_bit_cast(u32x4, b.m);
      sb[r][2] = __builtin_bit_cast(u32x4, b.l);
    }
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < NI; ++r)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        *reinterpret_cast<u32x4*>(As + p * PLANE + soff[r]) = sa[r][p];
        *reinterpret_cast<u32x4*>(Bs + p * PLANE + soff[r]) = sb[r][p];
      }
  };
  const int arow = wm * WT + (lane & 31), brow = wn * WT + (lane & 31), kh = lane >> 5;
  const int aswz = (arow >> 2) & 3, bswz = (brow >> 2) & 3;   // (+32 rows leaves the swizzle unchanged)
  bf16x8 fa[T][3], fb[T][3];
  auto reads = [&](int ks) __attribute__((always_inline)) {
    const char* ap = As + arow * 64 + (((ks * 2 + kh) ^ aswz) << 4);
    const char* bp = Bs + brow * 64 + (((ks * 2 + kh) ^ bswz) << 4);
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[i][p] = *reinterpret_cast<const bf16x8*>(ap + p * PLANE + i * 32 * 64);
        fb[i][p] = *reinterpret_cast<const bf16x8*>(bp + p * PLANE + i * 32 * 64);
      }
  };
  auto mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < T; ++mi)
#pragma unroll
      for (int ni = 0; ni < T; ++ni) {     // piece products hl, lh, mm, hm, mh, hh: small terms first
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][2], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][2], fb[ni][0], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][1], fb[ni][1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][1], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][1], fb[ni][0], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][0], acc[mi][ni], 0, 0, 0);
      }
  };
  // the scheduling pattern of half a K-tile: its fragment reads, then every MFMA followed by VALU of the split
  auto pattern = [&](auto valu_c) __attribute__((always_inline)) {
    constexpr int valu_per_mfma = decltype(valu_c)::value;
    __builtin_amdgcn_sched_group_barrier(0x100, 6 * T, 0);
#pragma unroll
    for (int i = 0; i < 6 * T * T; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, valu_per_mfma, 0);
    }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  using S0 = std::integral_constant<int, 0>;

  {
    auto ktile = [&](auto whole_c, int kt) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      reads(0);
      split_a(S0{});
      if (!PRE) split_b();
      if (decltype(whole_c)::value || kt + 2 < nk) {
        load_a(S0{}, whole_c, kt + 2);
        if (!PRE) load_b(whole_c, kt + 2);
      }
      mfmas();
      if (decltype(whole_c)::value) {
        pattern(std::integral_constant<int, PRE ? 4 : 8>{});
        __builtin_amdgcn_sched_group_barrier(0x020, ((AMUL ? 4 : 2) + (PRE ? 0 : 2)) * NI, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      reads(1);
      mfmas();
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      lstore();
      __syncthreads();
      if (PRE && (decltype(whole_c)::value || kt + 2 < nk)) load_b_pre(kt + 2);
    };
    load_a(S0{}, No{}, 0);
    if (PRE) load_b_pre(0);
    else load_b(No{}, 0);
    split_a(S0{});
    if (!PRE) split_b();
    lstore();
    if (nk > 1) {
      load_a(S0{}, No{}, 1);
      if (PRE) load_b_pre(1);
      else load_b(No{}, 1);
    }
    __syncthreads();
    int kt = 0;
    const int nwhole = K / BK;
    for (; kt + 2 < nwhole; ++kt) ktile(Yes{}, kt);
    for (; kt + 1 < nk; ++kt) ktile(No{}, kt);
  }
  reads(0);
  mfmas();
  reads(1);
  mfmas();
  __syncthreads();

  constexpr int LDC = TBN + 4;
  float* Cs = smem;                        // [TBM][LDC]
#pragma unroll
  for (int mi = 0; mi < T; ++mi)
#pragma unroll
    for (int ni = 0; ni < T; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WT + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wn * WT + ni * 32 + (lane & 31);
        Cs[row * LDC + col] = acc[mi][ni][r];
      }
  __syncthreads();
  switch (act) {
    case RECMV_ACT_RELU:
      nt_epilogue<T, RECMV_ACT_RELU>(Cs, bias, C, ldc, M, N, m0, n0, act_param, out_scale, c_vec, AMUL ? AMul{nullptr, 0, 0, 0.f, 1.f, 1.f} : am);
      break;
    case RECMV_ACT_SOFTPLUS:
      nt_epilogue<T, RECMV_ACT_SOFTPLUS>(Cs, bias, C, ldc, M, N, m0, n0, act_param, out_scale, c_vec, AMUL ? AMul{nullptr, 0, 0, 0.f, 1.f, 1.f} : am);
      break;
    case RECMV_ACT_TANH:
      nt_epilogue<T, RECMV_ACT_TANH>(Cs, bias, C, ldc, M, N, m0, n0, act_param, out_scale, c_vec, AMUL ? AMul{nullptr, 0, 0, 0.f, 1.f, 1.f} : am);
      break;
    default:
      nt_epilogue<T, RECMV_ACT_NONE>(Cs, bias, C, ldc, M, N, m0, n0, act_param, out_scale, c_vec, AMUL ? AMul{nullptr, 0, 0, 0.f, 1.f, 1.f} : am);
  }
}

// B [N][K] f32 (row stride ldb, K % 8 == 0) -> planes [3][N][Kp] bf16 (h, m, l of split8), zero beyond K.
__global__ __launch_bounds__(kBlk) void b3_split_kernel(const float* __restrict__ B, int64_t ldb, int64_t N, int K, int Kp,
                                                        char* __restrict__ planes) {
  const int c8 = Kp / 8;
  const int64_t items = N * c8, plane_bytes = N * (int64_t)Kp * 2;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < items; e += (int64_t)gridDim.x * kBlk) {
    const int64_t row = e / c8;
    const int k = (int)(e - row * c8) * 8;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (k < K) {                                    // (K % 8 == 0: a chunk is entirely inside or outside)
      a = *reinterpret_cast<const float4*>(B + row * ldb + k);
      b = *reinterpret_cast<const float4*>(B + row * ldb + k + 4);
    }
    const Pieces p = split8(a, b);
    char* o = planes + (row * Kp + k) * 2;
    *reinterpret_cast<u32x4*>(o) = __builtin_bit_cast(u32x4, p.h);
    *reinterpret_cast<u32x4*>(o + plane_bytes) = __builtin_bit_cast(u32x4, p.m);
    *reinterpret_cast<u32x4*>(o + 2 * plane_bytes) = __builtin_bit_cast(u32x4, p.l);
  }
}

// ------------------------------------------------------------------------------------------ NT, narrow tile
// 64x32 workgroup tile for launches too small to give every CU three 64x64 workgroups (the ray path: M = 3072 rows
// against N = 512 is 384 tiles of 64x64 — 1.5 per CU, so half the CUs carry twice the work of the others — but 768
// tiles of 64x32 = 3 per CU).  The 4 waves are 2 (row halves) x 2 (halves of every K-tile): each wave owns one 32x32
// MFMA tile over half of K, the two K-halves are summed through LDS before the epilogue.
template <bool FAST, bool AMUL, bool BF3>
__global__ __launch_bounds__(kBlk) void gemm_nt_narrow_kernel(const float* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ B, int64_t ldb,
                                                              const float* __restrict__ bias, float* __restrict__ C,
                                                              int64_t ldc, int M, int N, int K, int act,
                                                              float act_param, float out_scale, int nbm, int nbn,
                                                              bool a_vec, bool b_vec, bool c_vec, AMul am) {
  constexpr int TBM = 64, TBN = 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [2][64][LDK]
  float* Bs = smem + 2 * TBM * LDK;        // [2][32][LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wk = wave >> 1;
  const int64_t logical = xcd_remap(blockIdx.x, (int64_t)nbm * nbn);
  const int tile_m = (int)(logical / nbn), tile_n = (int)(logical % nbn);
  const int m0 = tile_m * TBM, n0 = tile_n * TBN;
  if (am.B2 && m0 >= am.split) {              // second weight set for the rows of the second net (see AMul)
    B = am.B2;
    bias = am.bias2;
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // staging: A 64 rows x 8 float4 (2 per thread), B 32 rows x 8 float4 (1 per thread)
  float4 ra[2], rb;
  const float* pa[2];
  const float* pb;
  const float* py[2];
  const int brow_s = tid >> 3, c4s = tid & 7;
  if (FAST) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      int gm = m0 + brow_s + 32 * r;
      gm = gm < M ? gm : M - 1;
      pa[r] = A + (int64_t)gm * lda + c4s * 4;
      if (AMUL) py[r] = am.Y + (int64_t)gm * am.ldy + c4s * 4;
    }
    int gn = n0 + brow_s;
    gn = gn < N ? gn : N - 1;
    pb = B + (int64_t)gn * ldb + c4s * 4;
  }
  auto gload = [&](int k0) {
    const int k = k0 + c4s * 4;
    if (FAST) {
      const bool in = k < K;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        ra[r] = in ? *reinterpret_cast<const float4*>(pa[r] + k0) : make_float4(0, 0, 0, 0);
        if (AMUL && in) ra[r] = amul4(ra[r], *reinterpret_cast<const float4*>(py[r] + k0), am);
      }
      rb = in ? *reinterpret_cast<const float4*>(pb + k0) : make_float4(0, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int gm = m0 + brow_s + 32 * r;
        ra[r] = (gm < M) ? load4_guard(A + (int64_t)gm * lda + k, K - k, a_vec) : make_float4(0, 0, 0, 0);
        if (AMUL && gm < M) ra[r] = amul4(ra[r], load4_guard(am.Y + (int64_t)gm * am.ldy + k, K - k, false), am);
      }
      const int gn = n0 + brow_s;
      rb = (gn < N) ? load4_guard(B + (int64_t)gn * ldb + k, K - k, b_vec) : make_float4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
      *reinterpret_cast<float4*>(As + (buf * TBM + brow_s + 32 * r) * LDK + c4s * 4) = ra[r];
    *reinterpret_cast<float4*>(Bs + (buf * TBN + brow_s) * LDK + c4s * 4) = rb;
  };

  const int nk = (K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int arow = wm * 32 + (lane & 31), brow = lane & 31, khalf = (lane >> 5) * 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    if (BF3) {
      const float* ap = As + (buf * TBM + arow) * LDK + 2 * khalf + wk * 16;
      const float* bp = Bs + (buf * TBN + brow) * LDK + 2 * khalf + wk * 16;
      const Pieces a = split8(*reinterpret_cast<const float4*>(ap), *reinterpret_cast<const float4*>(ap + 4));
      const Pieces b = split8(*reinterpret_cast<const float4*>(bp), *reinterpret_cast<const float4*>(bp + 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
    } else {
      const float* as = As + (buf * TBM + arow) * LDK + khalf + wk * 16;
      const float* bs = Bs + (buf * TBN + brow) * LDK + khalf + wk * 16;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const float4 a = *reinterpret_cast<const float4*>(as + kk * 8);
        const float4 b = *reinterpret_cast<const float4*>(bs + kk * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // the two K-halves -> two LDS planes [64][LDC]; the epilogue adds them
  constexpr int LDC = TBN + 4;
  float* Cs = smem;                        // [2][64][LDC] = 18.4 KB <= the operand buffers
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    Cs[(wk * TBM + row) * LDC + (lane & 31)] = acc[r];
  }
  __syncthreads();
  const int c4 = tid & 7, r0 = tid >> 3;   // 8 float4 strips per row, 32 rows per pass
  const int gn = n0 + c4 * 4;
  if (gn >= N) return;
  const float inv_p = act_param != 0.f ? 1.f / act_param : 0.f;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) {
    bv.x = bias[gn];
    if (gn + 1 < N) bv.y = bias[gn + 1];
    if (gn + 2 < N) bv.z = bias[gn + 2];
    if (gn + 3 < N) bv.w = bias[gn + 3];
  }
  const bool full4 = c_vec && gn + 4 <= N;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = r0 + 32 * pass;
    const int gm = m0 + row;
    if (gm >= M) break;
    const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * LDC + c4 * 4);
    const float4 v1 = *reinterpret_cast<const float4*>(Cs + (TBM + row) * LDC + c4 * 4);
    float4 v = make_float4(v0.x + v1.x + bv.x, v0.y + v1.y + bv.y, v0.z + v1.z + bv.z, v0.w + v1.w + bv.w);
    switch (act) {
      case RECMV_ACT_RELU:
        v.x = apply_act<RECMV_ACT_RELU>(v.x, act_param, inv_p);
        v.y = apply_act<RECMV_ACT_RELU>(v.y, act_param, inv_p);
        v.z = apply_act<RECMV_ACT_RELU>(v.z, act_param, inv_p);
        v.w = apply_act<RECMV_ACT_RELU>(v.w, act_param, inv_p);
        break;
      case RECMV_ACT_SOFTPLUS:
        v.x = apply_act<RECMV_ACT_SOFTPLUS>(v.x, act_param, inv_p);
        v.y = apply_act<RECMV_ACT_SOFTPLUS>(v.y, act_param, inv_p);
        v.z = apply_act<RECMV_ACT_SOFTPLUS>(v.z, act_param, inv_p);
        v.w = apply_act<RECMV_ACT_SOFTPLUS>(v.w, act_param, inv_p);
        break;
      case RECMV_ACT_TANH:
        v.x = apply_act<RECMV_ACT_TANH>(v.x, act_param, inv_p);
        v.y = apply_act<RECMV_ACT_TANH>(v.y, act_param, inv_p);
        v.z = apply_act<RECMV_ACT_TANH>(v.z, act_param, inv_p);
        v.w = apply_act<RECMV_ACT_TANH>(v.w, act_param, inv_p);
        break;
      default:
        break;
    }
    v.x *= out_scale;
    v.y *= out_scale;
    v.z *= out_scale;
    v.w *= out_scale;
    if (!AMUL && am.Y) v = emul4(v, am, gm, gn, N);
    float* dst = C + (int64_t)gm * ldc + gn;
    if (full4) {
      *reinterpret_cast<float4*>(dst) = v;
    } else {
      dst[0] = v.x;
      if (gn + 1 < N) dst[1] = v.y;
      if (gn + 2 < N) dst[2] = v.z;
      if (gn + 3 < N) dst[3] = v.w;
    }
  }
}

// ------------------------------------------------------------------------------------------ TN
// partial[split][M][N] = sum over k in the split's range of A[k][m]*B[k][n]
template <bool BF3>
__global__ __launch_bounds__(kBlk) void gemm_tn_kernel(const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ B, int64_t ldb,
                                                       float* __restrict__ P, int M, int N, int64_t K, int nbm,
                                                       int nbn, int64_t kchunk, bool a_vec, bool b_vec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][BK][LDM]
  float* Bs = smem + 2 * BK * LDM;        // [2][BK][LDM]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles = nbm * nbn;
  const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int tile_m = tile / nbn, tile_n = tile % nbn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int64_t kbeg = (int64_t)split * kchunk;
  int64_t kend = kbeg + kchunk;
  if (kend > K) kend = K;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging: BK rows x 128 floats = 1024 float4 per operand; krow = idx/32, c4 = idx%32
  float4 ra[4], rb[4];
  // `fast` (uniform): 16-byte aligned operands whose widths are multiples of 4 — a float4 of a row is entirely inside or
  // outside the matrix, so every staging load is one unconditional 16-byte load (an outside one reads a clamped address and
  // is zeroed); whole tiles (the common case: M, N multiples of 128, K-tile inside the split) skip the zeroing too.
  // Otherwise the element-guarded loader, whose per-lane branches keep the eight loads of a K-tile from overlapping.
  const bool fast = a_vec && b_vec && (M & 3) == 0 && (N & 3) == 0 && M >= 4 && N >= 4;
  const bool whole_mn = m0 + BM <= M && n0 + BN <= N;
  auto gload = [&](int64_t k0) {
    if (fast) {
      const bool whole = whole_mn && k0 + BK <= kend;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = tid + kBlk * r;
        const int krow = idx >> 5, c4 = idx & 31;
        const int64_t k = k0 + krow;
        const int cm = m0 + c4 * 4, cn = n0 + c4 * 4;
        if (whole) {
          ra[r] = *reinterpret_cast<const float4*>(A + k * lda + cm);
          rb[r] = *reinterpret_cast<const float4*>(B + k * ldb + cn);
        } else {
          const bool kin = k < kend, ain = kin && cm < M, bin = kin && cn < N;
          const int64_t kc = kin ? k : kend - 1;
          float4 a = *reinterpret_cast<const float4*>(A + kc * lda + (cm < M ? cm : M - 4));
          float4 b = *reinterpret_cast<const float4*>(B + kc * ldb + (cn < N ? cn : N - 4));
          ra[r] = make_float4(ain ? a.x : 0.f, ain ? a.y : 0.f, ain ? a.z : 0.f, ain ? a.w : 0.f);
          rb[r] = make_float4(bin ? b.x : 0.f, bin ? b.y : 0.f, bin ? b.z : 0.f, bin ? b.w : 0.f);
        }
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + kBlk * r;
      const int krow = idx >> 5, c4 = idx & 31;
      const int64_t k = k0 + krow;
      const int cm = m0 + c4 * 4, cn = n0 + c4 * 4;
      ra[r] = (k < kend) ? load4_guard(A + k * lda + cm, M - cm, a_vec) : make_float4(0, 0, 0, 0);
      rb[r] = (k < kend) ? load4_guard(B + k * ldb + cn, N - cn, b_vec) : make_float4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + kBlk * r;
      const int krow = idx >> 5, c4 = idx & 31;
      *reinterpret_cast<float4*>(As + (buf * BK + krow) * LDM + c4 * 4) = ra[r];
      *reinterpret_cast<float4*>(Bs + (buf * BK + krow) * LDM + c4 * 4) = rb[r];
    }
  };

  const int nk = (int)((kend - kbeg + BK - 1) / BK);
  if (nk > 0) {
    gload(kbeg);
    lstore(0);
  }
  __syncthreads();
  const int acol = wm * 64 + (lane & 31), bcol = wn * 64 + (lane & 31), kh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kbeg + (int64_t)(kt + 1) * BK);
    if (BF3) {
      // bf16x6 (see split2): a lane's 8 consecutive k of column m are 8 rows of the k-major LDS tile
      const float* as = As + (buf * BK + 8 * kh) * LDM + acol;
      const float* bs = Bs + (buf * BK + 8 * kh) * LDM + bcol;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        Pieces pa[2], pb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float* ap = as + ks * 16 * LDM + 32 * i;
          const float* bp = bs + ks * 16 * LDM + 32 * i;
          pa[i] = split8(make_float4(ap[0], ap[LDM], ap[2 * LDM], ap[3 * LDM]),
                         make_float4(ap[4 * LDM], ap[5 * LDM], ap[6 * LDM], ap[7 * LDM]));
          pb[i] = split8(make_float4(bp[0], bp[LDM], bp[2 * LDM], bp[3 * LDM]),
                         make_float4(bp[4 * LDM], bp[5 * LDM], bp[6 * LDM], bp[7 * LDM]));
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].h, pb[ni].l, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].l, pb[ni].h, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].m, pb[ni].m, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].h, pb[ni].m, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].m, pb[ni].h, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mi].h, pb[ni].h, acc[mi][ni], 0, 0, 0);
          }
      }
    } else {
    const float* as = As + (buf * BK + kh) * LDM + acol;
    const float* bs = Bs + (buf * BK + kh) * LDM + bcol;
#pragma unroll
    for (int k2 = 0; k2 < BK / 2; ++k2) {
      const float a0 = as[k2 * 2 * LDM], a1 = as[k2 * 2 * LDM + 32];
      const float b0 = bs[k2 * 2 * LDM], b1 = bs[k2 * 2 * LDM + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  float* Ps = P + (int64_t)split * M * N;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int gn = n0 + wn * 64 + ni * 32 + (lane & 31);
    if (gn >= N) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm < M) Ps[(int64_t)gm * N + gn] = acc[mi][ni][r];
      }
    }
  }
}

// The same partial products with ONE 16-row K-tile in LDS (17 KB instead of 68 KB) and at most 128 registers: four workgroups
// per CU instead of two (the same step the NT kernel took: a workgroup's barriers, prologue and register -> HBM epilogue are
// covered by three neighbours instead of one).  f32 mode, aligned whole-float4 operands only; same summation order.
template <int BKT>
__global__ __launch_bounds__(kBlk, 4) void gemm_tn_occ_kernel(const float* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ B, int64_t ldb,
                                                              float* __restrict__ P, int M, int N, int64_t K, int nbm,
                                                              int nbn, int64_t kchunk) {
  constexpr int NLD = BKT * 32 / kBlk;    // float4 per thread per operand tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [BKT][LDM]
  float* Bs = smem + BKT * LDM;           // [BKT][LDM]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles = nbm * nbn;
  // consecutive workgroup ids go round the 8 XCDs: keep ALL output tiles of a split (they read the same operand rows) on one XCD,
  // so that its L2 fetches those rows once — XCD x takes the splits x, x + 8, ...
  int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const int splits = gridDim.x / tiles;
  if (splits % kNumXCD == 0) {
    const int x = blockIdx.x % kNumXCD, j = blockIdx.x / kNumXCD;
    split = x + kNumXCD * (j / tiles);
    tile = j % tiles;
  }
  const int tile_m = tile / nbn, tile_n = tile % nbn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int64_t kbeg = (int64_t)split * kchunk;
  int64_t kend = kbeg + kchunk;
  if (kend > K) kend = K;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[NLD], rb[NLD];
  const bool whole_mn = m0 + BM <= M && n0 + BN <= N;
  auto gload = [&](int64_t k0) {
    const bool whole = whole_mn && k0 + BKT <= kend;
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int idx = tid + kBlk * r;
      const int krow = idx >> 5, c4 = idx & 31;
      const int64_t k = k0 + krow;
      const int cm = m0 + c4 * 4, cn = n0 + c4 * 4;
      if (whole) {
        ra[r] = *reinterpret_cast<const float4*>(A + k * lda + cm);
        rb[r] = *reinterpret_cast<const float4*>(B + k * ldb + cn);
      } else {
        const bool kin = k < kend, ain = kin && cm < M, bin = kin && cn < N;
        const int64_t kc = kin ? k : kend - 1;
        // widths that are no multiple of 4 (the skip layer's 473 columns inside a 512-wide buffer): the launcher has checked that
        // the row strides cover the rounded-up widths, so the last float4 of a row reads up to 3 elements of padding — they only
        // reach output rows / columns >= M / N, which are never stored
        float4 a = *reinterpret_cast<const float4*>(A + kc * lda + (cm < M ? cm : (M - 1) & ~3));
        float4 b = *reinterpret_cast<const float4*>(B + kc * ldb + (cn < N ? cn : (N - 1) & ~3));
        ra[r] = make_float4(ain ? a.x : 0.f, ain ? a.y : 0.f, ain ? a.z : 0.f, ain ? a.w : 0.f);
        rb[r] = make_float4(bin ? b.x : 0.f, bin ? b.y : 0.f, bin ? b.z : 0.f, bin ? b.w : 0.f);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int idx = tid + kBlk * r;
      const int krow = idx >> 5, c4 = idx & 31;
      *reinterpret_cast<float4*>(As + krow * LDM + c4 * 4) = ra[r];
      *reinterpret_cast<float4*>(Bs + krow * LDM + c4 * 4) = rb[r];
    }
  };

  const int nk = (int)((kend - kbeg + BKT - 1) / BKT);
  if (nk > 0) {
    gload(kbeg);
    lstore();
  }
  __syncthreads();
  const int acol = wm * 64 + (lane & 31), bcol = wn * 64 + (lane & 31), kh = lane >> 5;
  const float* as = As + kh * LDM + acol;
  const float* bs = Bs + kh * LDM + bcol;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kbeg + (int64_t)(kt + 1) * BKT);
#pragma unroll
    for (int k2 = 0; k2 < BKT / 2; ++k2) {
      const float a0 = as[k2 * 2 * LDM], a1 = as[k2 * 2 * LDM + 32];
      const float b0 = bs[k2 * 2 * LDM], b1 = bs[k2 * 2 * LDM + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) lstore();
    __syncthreads();
  }

  float* Ps = P + (int64_t)split * M * N;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int gn = n0 + wn * 64 + ni * 32 + (lane & 31);
    if (gn >= N) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm < M) Ps[(int64_t)gm * N + gn] = acc[mi][ni][r];
      }
    }
  }
}

// C[m][n] = sum_s P[s][m][n]   (fixed order -> deterministic)
__global__ __launch_bounds__(kBlk) void splitk_reduce_kernel(const float* __restrict__ P, float* __restrict__ C,
                                                             int64_t ldc, int M, int N, int splits) {
  const int64_t total = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlk) {
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += P[(int64_t)sp * total + i];
    C[(i / N) * ldc + (i % N)] = s;
  }
}

// The same sums in the same order, four columns per lane and eight partial tiles requested before the first is added: the
// scalar loop above asks for one 4-byte value per split and lane at a time (64 dependent round trips for 64 splits).
__global__ __launch_bounds__(kBlk) void splitk_reduce4_kernel(const float4* __restrict__ P, float* __restrict__ C,
                                                              int64_t ldc, int M, int N, int splits) {
  const int64_t total4 = (int64_t)M * N / 4;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < total4; i += (int64_t)gridDim.x * kBlk) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = 0;
    for (; sp + 8 <= splits; sp += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = P[(int64_t)(sp + u) * total4 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s.x += v[u].x;
        s.y += v[u].y;
        s.z += v[u].z;
        s.w += v[u].w;
      }
    }
    for (; sp < splits; ++sp) {
      const float4 v = P[(int64_t)sp * total4 + i];
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
    const int64_t e = i * 4;
    *reinterpret_cast<float4*>(C + (e / N) * ldc + (e % N)) = s;
  }
}

// ------------------------------------------------------------------------------------------ posenc
struct PeWeights {
  float w[32];  // by-value kernel argument: no H2D copy, graph-capturable
};
__global__ __launch_bounds__(kBlk) void posenc_kernel(const float* __restrict__ x, int64_t ldx,
                                                      float* __restrict__ out, int64_t ldo, int64_t ldo_fill,
                                                      int64_t P, int L, PeWeights w, float out_scale) {
  const int nf = 1 + 2 * L;  // identity + (sin,cos) per band
  const int64_t total = P * nf;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlk) {
    const int64_t p = e / nf;
    const int f = (int)(e % nf);
    const float x0 = x[p * ldx], x1 = x[p * ldx + 1], x2 = x[p * ldx + 2];
    float* o = out + p * ldo + 3 * f;
    if (f == 0) {
      o[0] = x0 * out_scale;
      o[1] = x1 * out_scale;
      o[2] = x2 * out_scale;
      for (int64_t c = 3 * nf; c < ldo_fill; ++c) out[p * ldo + c] = 0.f;
    } else {
      const int band = (f - 1) >> 1;
      const float freq = (float)(1 << band);  // 2**linspace(0, L-1, L): exact powers of two
      const float wt = w.w[f - 1];
      float v0, v1, v2;
      if ((f - 1) & 1) {
        v0 = cosf(x0 * freq); v1 = cosf(x1 * freq); v2 = cosf(x2 * freq);
      } else {
        v0 = sinf(x0 * freq); v1 = sinf(x1 * freq); v2 = sinf(x2 * freq);
      }
      o[0] = wt * v0 * out_scale;
      o[1] = wt * v1 * out_scale;
      o[2] = wt * v2 * out_scale;
    }
  }
}

// ---- optional per-launch HIP-event timing of the MFMA kernels (bench.py's roofline object) ----------------------
// Events are recorded on the stream the kernel is launched on, immediately before and after the launch.
struct LaunchRec {
  hipEvent_t a, b;
  int variant;
  double flops, bytes;
};
struct Profiler {
  bool on = false;
  double min_flops = 0.0;          // launches below this are counted but not bracketed by events
  double untimed[14][2] = {};      // [variant][launches, flops]
  std::vector<LaunchRec> recs;
  std::vector<hipEvent_t> pool;
  double large[14][3] = {};        // of the last recmv_profile_end: launches / seconds / FLOP of the bracketed launches of >= 4 GFLOP
  double timed_bytes[14] = {};     // of the last recmv_profile_end: algorithmic bytes (4 (MK + NK + MN)) of the bracketed launches, per variant
  double busy_union_s = 0.0, busy_span_s = 0.0;   // of the last recmv_profile_end: union of the bracketed intervals, first start -> last end
  std::mutex mu;        // autograd's backward thread launches too
  hipEvent_t get() {
    std::lock_guard<std::mutex> lk(mu);
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
};
Profiler g_prof;

struct ScopedLaunchTimer {
  LaunchRec r;
  hipStream_t s;
  bool active;
  ScopedLaunchTimer(int variant, double M, double N, double K, hipStream_t stream) : s(stream), active(g_prof.on) {
    if (!active) return;
    const double flops = 2.0 * M * N * K;
    r.bytes = 4.0 * (M * K + N * K + M * N);
    if (flops < g_prof.min_flops) {
      std::lock_guard<std::mutex> lk(g_prof.mu);
      g_prof.untimed[variant][0] += 1.0;
      g_prof.untimed[variant][1] += flops;
      active = false;
      return;
    }
    r.variant = variant;
    r.flops = flops;
    r.a = g_prof.get();
    r.b = g_prof.get();
    if (!r.a || !r.b) {
      active = false;
      return;
    }
    (void)hipEventRecord(r.a, s);
  }
  ~ScopedLaunchTimer() {
    if (!active) return;
    (void)hipEventRecord(r.b, s);
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.recs.push_back(r);
  }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int kNtLds = (2 * BM * LDK + 2 * BN * LDK) * 4;   // 73728 B (T=2); half of it for T=1
constexpr int kTnLds = (4 * BK * LDM) * 4;                  // 67584 B

bool tn_occ() {     // RECMV_GEMM_OCC=0: the two-workgroups-per-CU kernels (A/B)
  static const bool v = [] { const char* e = getenv("RECMV_GEMM_OCC"); return !(e && e[0] == '0'); }();
  return v;
}

int tn_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ceil_div(M, BM) * ceil_div(N, BN);
  int64_t want = ceil_div((int64_t)kNumCU * 4, tiles);        // ~4 workgroups per CU overall
  const int64_t maxs = ceil_div(K, (int64_t)BK * 4);          // at least 4 K-tiles per split
  if (want > maxs) want = maxs;
  if (want > 128) want = 128;
  if (want < 1) want = 1;
  return (int)want;
}

}  // namespace
}  // namespace recmv

using namespace recmv;

int g_b3_families = 7;   // mode 1 only: bit 0 the 128 x 128 NT kernels, bit 1 the 64 x 64 / 64 x 32 NT kernels, bit 2 the TN (dW) kernel
int g_gemm_mode = 0;     // 0: f32 MFMA (exact f32 products), 1: bf16x6 (3-way bf16 split, six bf16 MFMA products)

template <int T, bool FAST, bool AMUL, bool BF3>
static int launch_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C,
                     int64_t ldc, int64_t M, int64_t N, int64_t K, int act, float act_param, float out_scale,
                     bool a_vec, bool b_vec, bool c_vec, const AMul& am, hipStream_t stream) {
  constexpr int lds = kNtLds / (T == 2 ? 1 : 2);
  static bool attr_set = false;
  if (!attr_set) {
    RECMV_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_kernel<T, FAST, AMUL, BF3>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int nbm = (int)ceil_div(M, 64 * T), nbn = (int)ceil_div(N, 64 * T);
  ScopedLaunchTimer timer((T - 1) + 2 * (FAST ? 1 : 0) + 4 * (AMUL ? 1 : 0), (double)M, (double)N, (double)K, stream);
  hipLaunchKernelGGL((gemm_nt_kernel<T, FAST, AMUL, BF3>), dim3((unsigned)((int64_t)nbm * nbn)), dim3(kBlk), lds, stream, A,
                     lda, B, ldb, bias, C, ldc, (int)M, (int)N, (int)K, act, act_param, out_scale, nbm, nbn, a_vec,
                     b_vec, c_vec, am);
  return check_launch("gemm_nt");
}

template <bool AMUL, int BKT, bool SINGLE, int MI, int NI>
static int launch_nt_occ(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C,
                         int64_t ldc, int64_t M, int64_t N, int64_t K, int act, float act_param, float out_scale,
                         bool c_vec, const AMul& am, hipStream_t stream) {
  constexpr int lds_ops = (SINGLE ? 1 : 2) * 64 * (MI + NI) * (BKT + 4) * 4, lds_c = 32 * MI * (64 * NI + 4) * 4;
  constexpr int lds = lds_ops > lds_c ? lds_ops : lds_c;
  const int nbm = (int)ceil_div(M, 64 * MI), nbn = (int)ceil_div(N, 64 * NI);
  ScopedLaunchTimer timer(AMUL ? 11 : (MI == 2 ? 9 : 10), (double)M, (double)N, (double)K, stream);
  hipLaunchKernelGGL((gemm_nt_occ_kernel<AMUL, BKT, SINGLE, MI, NI>), dim3((unsigned)((int64_t)nbm * nbn)), dim3(kBlk), lds,
                     stream, A, lda, B, ldb, bias, C, ldc, (int)M, (int)N, (int)K, act, act_param, out_scale, nbm, nbn, c_vec,
                     am);
  return check_launch("gemm_nt(occ)");
}

// Weight matrices whose bf16 planes exist (recmv_b3_split), by the address the products get them under.
struct B3Entry {
  const char* planes;
  int64_t N, K, ldb, Kp;
};
static std::mutex g_b3_mu;
static std::unordered_map<const float*, B3Entry> g_b3;
static bool b3_lookup(const float* B, int64_t N, int64_t K, int64_t ldb, B3Entry* out) {
  std::lock_guard<std::mutex> lock(g_b3_mu);
  auto it = g_b3.find(B);
  if (it == g_b3.end() || it->second.K != K || it->second.ldb != ldb || it->second.N < N) return false;
  *out = it->second;
  return true;
}

template <int T, bool AMUL, bool PRE>
static int launch_nt_b3_as(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc,
                           int64_t M, int64_t N, int64_t K, int act, float act_param, float out_scale, bool c_vec, const AMul& am,
                           const B3Pre& pre, hipStream_t stream) {
  constexpr int lds_ops = 6 * 64 * T * 64, lds_c = 64 * T * (64 * T + 4) * 4;
  constexpr int lds = lds_ops > lds_c ? lds_ops : lds_c;
  static bool attr_set = false;
  if (!attr_set) {
    RECMV_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_b3_kernel<T, AMUL, PRE>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int nbm = (int)ceil_div(M, 64 * T), nbn = (int)ceil_div(N, 64 * T);
  ScopedLaunchTimer timer((T - 1) + 2 + 4 * (AMUL ? 1 : 0), (double)M, (double)N, (double)K, stream);
  hipLaunchKernelGGL((gemm_nt_b3_kernel<T, AMUL, PRE>), dim3((unsigned)((int64_t)nbm * nbn)), dim3(kBlk), lds, stream, A, lda,
                     B, ldb, bias, C, ldc, (int)M, (int)N, (int)K, act, act_param, out_scale, nbm, nbn, c_vec, am, pre);
  return check_launch("gemm_nt(b3)");
}

template <int T, bool AMUL>
static int launch_nt_b3(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C,
                        int64_t ldc, int64_t M, int64_t N, int64_t K, int act, float act_param, float out_scale,
                        bool c_vec, const AMul& am, hipStream_t stream) {
  B3Entry e, e2;
  if (b3_lookup(B, N, K, ldb, &e) && (!am.B2 || (b3_lookup(am.B2, N, K, ldb, &e2) && e2.Kp == e.Kp))) {
    B3Pre pre = {e.planes, am.B2 ? e2.planes : nullptr, e.N * e.Kp * 2, am.B2 ? e2.N * e2.Kp * 2 : 0, (int)e.Kp};
    return launch_nt_b3_as<T, AMUL, true>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param, out_scale, c_vec, am, pre, stream);
  }
  B3Pre none = {nullptr, nullptr, 0, 0, 0};
  return launch_nt_b3_as<T, AMUL, false>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param, out_scale, c_vec, am, none, stream);
}

constexpr int kNarrowLds = 2 * (64 + 32) * LDK * 4;   // 27648 B

template <bool FAST, bool AMUL, bool BF3>
static int launch_nt_narrow(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C,
                            int64_t ldc, int64_t M, int64_t N, int64_t K, int act, float act_param, float out_scale,
                            bool a_vec, bool b_vec, bool c_vec, const AMul& am, hipStream_t stream) {
  const int nbm = (int)ceil_div(M, 64), nbn = (int)ceil_div(N, 32);
  ScopedLaunchTimer timer(FAST && !AMUL ? 12 : 13, (double)M, (double)N, (double)K, stream);
  hipLaunchKernelGGL((gemm_nt_narrow_kernel<FAST, AMUL, BF3>), dim3((unsigned)((int64_t)nbm * nbn)), dim3(kBlk),
                     kNarrowLds, stream, A, lda, B, ldb, bias, C, ldc, (int)M, (int)N, (int)K, act, act_param, out_scale,
                     nbm, nbn, a_vec, b_vec, c_vec, am);
  return check_launch("gemm_nt(narrow)");
}

template <bool AMUL>
static int dispatch_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, float* C,
                       int64_t ldc, int64_t M, int64_t N, int64_t K, int act, float act_param, float out_scale,
                       const AMul& am, hipStream_t s) {
  const bool a_vec = aligned16(A) && lda % 4 == 0 && (!AMUL || (aligned16(am.Y) && am.ldy % 4 == 0));
  const bool b_vec = aligned16(B) && ldb % 4 == 0 && (!am.B2 || aligned16(am.B2));
  const bool c_vec = aligned16(C) && ldc % 4 == 0;
  const bool fast = a_vec && b_vec && K % 4 == 0 && K > 0;
  // tile choice: 128x128 tiles unless they would leave the 256 CUs under-filled (< 2 workgroups per CU)
  const int64_t big_blocks = ceil_div(M, BM) * ceil_div(N, BN);
  // (which kernel families take the bf16x6 path in mode 1: recmv_set_b3_families, all of them by default)
  const bool bf3_big = g_gemm_mode == 1 && (g_b3_families & 1), bf3_mid = g_gemm_mode == 1 && (g_b3_families & 2);
#define RECMV_NT(TT, FF)                                                                                          \
  ((TT == 2 ? bf3_big : bf3_mid) ? launch_nt<TT, FF, AMUL, true>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param, out_scale, \
                                                    a_vec, b_vec, c_vec, am, s)                                    \
                    : launch_nt<TT, FF, AMUL, false>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param,        \
                                                     out_scale, a_vec, b_vec, c_vec, am, s))
  if (big_blocks >= 2 * kNumCU) {
    if (bf3_big && fast) {
      return launch_nt_b3<2, AMUL>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param, out_scale, c_vec, am, s);
    }
    // f32 mode, aligned operands: the high-occupancy kernels (gemm_nt_occ_kernel) — 128x128 tiles at four workgroups per CU
    // for the largest launches, 64x128 tiles at five per CU below ~3600 large tiles (finer tail, measured crossover between
    // 90 k and 150 k rows at N = 512: profiles/r03_gemm_occupancy_variants.txt).  RECMV_GEMM_OCC=0 keeps the two-per-CU
    // kernel for the A/B.
    static const bool occ = [] { const char* e = getenv("RECMV_GEMM_OCC"); return !(e && e[0] == '0'); }();
    if (fast && !bf3_big && occ) {
      if (big_blocks >= 3600)
        return launch_nt_occ<AMUL, 16, true, 2, 2>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param, out_scale, c_vec, am, s);
      return launch_nt_occ<AMUL, 16, true, 1, 2>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param, out_scale, c_vec, am, s);
    }
    return fast ? RECMV_NT(2, true) : RECMV_NT(2, false);
  }
  // 64x64 tiles unless they would give the CUs fewer than ~2.5 workgroups each: then 64x32 tiles (twice as many)
  const int64_t mid_blocks = ceil_div(M, 64) * ceil_div(N, 64);
  if (mid_blocks < (5 * kNumCU) / 2) {
#define RECMV_NTN(FF)                                                                                                  \
  (bf3_mid ? launch_nt_narrow<FF, AMUL, true>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param, out_scale, \
                                                       a_vec, b_vec, c_vec, am, s)                                     \
                    : launch_nt_narrow<FF, AMUL, false>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param,          \
                                                        out_scale, a_vec, b_vec, c_vec, am, s))
    return fast ? RECMV_NTN(true) : RECMV_NTN(false);
#undef RECMV_NTN
  }
  return fast ? RECMV_NT(1, true) : RECMV_NT(1, false);
#undef RECMV_NT
}

// Matrix mode of recmv_gemm_nt: 0 = f32-input MFMA (default; bit-for-bit an f32 fma chain), 1 = "bf16x6" (each f32
// operand split into three bf16 pieces in registers, six bf16 MFMA products per tile step, f32 accumulation: f32-level
// accuracy at up to 2.7x the f32 matrix rate).  Returns the previous mode.
extern "C" int recmv_set_gemm_mode(int mode) {
  const int prev = g_gemm_mode;
  if (mode == 0 || mode == 1) g_gemm_mode = mode;
  return prev;
}

extern "C" int recmv_get_gemm_mode(void) { return g_gemm_mode; }

// Mode 1 only — which kernel families compute in bf16x6 (the others stay on the exact-f32 kernels): bit 0 = 128 x 128 NT tiles, bit 1 =
// 64 x 64 and 64 x 32 NT tiles, bit 2 = TN (dW) tiles.  7 (default) = all.  A bisect / A-B switch (tools/erratum/loop_repro_inproc.py); returns
// the previous mask.
extern "C" int recmv_set_b3_families(int mask) {
  const int prev = g_b3_families;
  if (mask >= 0 && mask <= 7) g_b3_families = mask;
  return prev;
}

extern "C" int64_t recmv_b3_planes_bytes(int64_t N, int64_t K) {
  if (N <= 0 || K <= 0) return 0;
  return 3 * N * (ceil_div(K, (int64_t)BK) * BK) * 2;
}

extern "C" int recmv_b3_split(const float* B, int64_t ldb, int64_t N, int64_t K, void* planes, int64_t planes_bytes, void* stream) {
  RECMV_REQUIRE(B && planes, "b3_split: NULL pointer");
  RECMV_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldb >= K && ldb % 4 == 0 && aligned16(B) && aligned16(planes),
                "b3_split: needs K %% 8 == 0 and 16-byte aligned rows (N=%lld K=%lld ldb=%lld)", (long long)N, (long long)K,
                (long long)ldb);
  RECMV_REQUIRE(K < (1ll << 30) && N < (1ll << 31), "b3_split: size overflow");
  const int64_t Kp = ceil_div(K, (int64_t)BK) * BK;
  RECMV_REQUIRE(planes_bytes >= 3 * N * Kp * 2, "b3_split: planes buffer %lld < %lld bytes", (long long)planes_bytes,
                (long long)(3 * N * Kp * 2));
  hipLaunchKernelGGL(b3_split_kernel, dim3(stream_grid(N * (Kp / 8), kBlk)), dim3(kBlk), 0, (hipStream_t)stream, B, ldb, N, (int)K,
                     (int)Kp, (char*)planes);
  {
    const int rc = check_launch("b3_split");
    if (rc) return rc;
  }
  std::lock_guard<std::mutex> lock(g_b3_mu);
  g_b3[B] = B3Entry{(const char*)planes, N, K, ldb, Kp};
  return RECMV_OK;
}

extern "C" int recmv_b3_forget(const float* B) {
  std::lock_guard<std::mutex> lock(g_b3_mu);
  g_b3.erase(B);
  return RECMV_OK;
}

extern "C" int recmv_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                             float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                             float act_param, float out_scale, void* stream) {
  RECMV_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_nt: negative size");
  if (M == 0 || N == 0) return RECMV_OK;
  RECMV_REQUIRE(A && B && C, "gemm_nt: NULL pointer");
  RECMV_REQUIRE(lda >= K && ldb >= K && ldc >= N, "gemm_nt: leading dimension too small");
  RECMV_REQUIRE(M < (1ll << 31) - BM && N < (1ll << 31) - BN && K < (1ll << 31) - BK, "gemm_nt: size overflow");
  RECMV_REQUIRE(act >= RECMV_ACT_NONE && act <= RECMV_ACT_TANH, "gemm_nt: unknown activation %d", act);
  AMul am = {nullptr, 0, RECMV_ACT_NONE, 0.f, 1.f, 1.f};
  return dispatch_nt<false>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param, out_scale, am, (hipStream_t)stream);
}

// C[M,N] = (G (.) act'(y_scale * Y) * g_scale) . B^T : the activation-gradient step fused into the product.
// G's row stride ldg may be 0 (one cotangent row for every point).
extern "C" int recmv_gemm_nt_actgrad(const float* G, int64_t ldg, const float* Y, int64_t ldy, const float* B,
                                     int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                                     float act_param, float y_scale, float g_scale, void* stream) {
  RECMV_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_nt_actgrad: negative size");
  if (M == 0 || N == 0) return RECMV_OK;
  RECMV_REQUIRE(G && Y && B && C, "gemm_nt_actgrad: NULL pointer");
  RECMV_REQUIRE((ldg >= K || ldg == 0) && ldy >= K && ldb >= K && ldc >= N,
                "gemm_nt_actgrad: leading dimension too small");
  RECMV_REQUIRE(M < (1ll << 31) - BM && N < (1ll << 31) - BN && K < (1ll << 31) - BK, "gemm_nt_actgrad: size overflow");
  RECMV_REQUIRE(act >= RECMV_ACT_NONE && act <= RECMV_ACT_TANH, "gemm_nt_actgrad: unknown activation %d", act);
  AMul am = {Y, ldy, act, act_param, y_scale, g_scale};
  return dispatch_nt<true>(G, ldg, B, ldb, nullptr, C, ldc, M, N, K, RECMV_ACT_NONE, 0.f, 1.f, am, (hipStream_t)stream);
}

// C[M,N] = (A . B^T) (.) act'(y_scale * Y) * scale : the activation-gradient step of the NEXT backward layer fused into the
// epilogue of the product that creates its cotangent (Y [M,N], row stride ldy).
extern "C" int recmv_gemm_nt_mulgrad(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                     int64_t M, int64_t N, int64_t K, const float* Y, int64_t ldy, int act,
                                     float act_param, float y_scale, float scale, void* stream) {
  RECMV_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_nt_mulgrad: negative size");
  if (M == 0 || N == 0) return RECMV_OK;
  RECMV_REQUIRE(A && B && C && Y, "gemm_nt_mulgrad: NULL pointer");
  RECMV_REQUIRE(lda >= K && ldb >= K && ldc >= N && ldy >= N, "gemm_nt_mulgrad: leading dimension too small");
  RECMV_REQUIRE(M < (1ll << 31) - BM && N < (1ll << 31) - BN && K < (1ll << 31) - BK, "gemm_nt_mulgrad: size overflow");
  RECMV_REQUIRE(act >= RECMV_ACT_NONE && act <= RECMV_ACT_TANH, "gemm_nt_mulgrad: unknown activation %d", act);
  AMul em = {Y, ldy, act, act_param, y_scale, scale};
  return dispatch_nt<false>(A, lda, B, ldb, nullptr, C, ldc, M, N, K, RECMV_ACT_NONE, 0.f, 1.f, em, (hipStream_t)stream);
}

// The same products over a row block that holds the rows of TWO nets of one shape: rows [0, split_row) use B / bias, rows
// [split_row, M) use B2 / bias2 (split_row a multiple of 128, the largest tile height; B2 == NULL: the plain product).
extern "C" int recmv_gemm_nt_seg(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                                 const float* B2, const float* bias2, int64_t split_row, float* C, int64_t ldc, int64_t M,
                                 int64_t N, int64_t K, int act, float act_param, float out_scale, void* stream) {
  RECMV_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_nt_seg: negative size");
  if (M == 0 || N == 0) return RECMV_OK;
  RECMV_REQUIRE(A && B && C, "gemm_nt_seg: NULL pointer");
  RECMV_REQUIRE(lda >= K && ldb >= K && ldc >= N, "gemm_nt_seg: leading dimension too small");
  RECMV_REQUIRE(M < (1ll << 31) - BM && N < (1ll << 31) - BN && K < (1ll << 31) - BK, "gemm_nt_seg: size overflow");
  RECMV_REQUIRE(act >= RECMV_ACT_NONE && act <= RECMV_ACT_TANH, "gemm_nt_seg: unknown activation %d", act);
  RECMV_REQUIRE(!B2 || (split_row >= 0 && split_row % BM == 0), "gemm_nt_seg: split_row must be a multiple of %d", BM);
  RECMV_REQUIRE(!B2 || !bias == !bias2, "gemm_nt_seg: both nets with or both without a bias");
  AMul am = {nullptr, 0, RECMV_ACT_NONE, 0.f, 1.f, 1.f, B2, bias2, (int)(split_row < M ? split_row : M)};
  return dispatch_nt<false>(A, lda, B, ldb, bias, C, ldc, M, N, K, act, act_param, out_scale, am, (hipStream_t)stream);
}

extern "C" int recmv_gemm_nt_mulgrad_seg(const float* A, int64_t lda, const float* B, const float* B2, int64_t split_row,
                                         int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const float* Y,
                                         int64_t ldy, int act, float act_param, float y_scale, float scale, void* stream) {
  RECMV_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_nt_mulgrad_seg: negative size");
  if (M == 0 || N == 0) return RECMV_OK;
  RECMV_REQUIRE(A && B && C && Y, "gemm_nt_mulgrad_seg: NULL pointer");
  RECMV_REQUIRE(lda >= K && ldb >= K && ldc >= N && ldy >= N, "gemm_nt_mulgrad_seg: leading dimension too small");
  RECMV_REQUIRE(M < (1ll << 31) - BM && N < (1ll << 31) - BN && K < (1ll << 31) - BK, "gemm_nt_mulgrad_seg: size overflow");
  RECMV_REQUIRE(act >= RECMV_ACT_NONE && act <= RECMV_ACT_TANH, "gemm_nt_mulgrad_seg: unknown activation %d", act);
  RECMV_REQUIRE(!B2 || (split_row >= 0 && split_row % BM == 0), "gemm_nt_mulgrad_seg: split_row must be a multiple of %d", BM);
  AMul em = {Y, ldy, act, act_param, y_scale, scale, B2, nullptr, (int)(split_row < M ? split_row : M)};
  return dispatch_nt<false>(A, lda, B, ldb, nullptr, C, ldc, M, N, K, RECMV_ACT_NONE, 0.f, 1.f, em, (hipStream_t)stream);
}

extern "C" int64_t recmv_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int64_t)tn_splits(M, N, K) * M * N * 4;
}

extern "C" int recmv_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                             int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes,
                             void* stream) {
  RECMV_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_tn: negative size");
  if (M == 0 || N == 0) return RECMV_OK;
  RECMV_REQUIRE(A && B && C, "gemm_tn: NULL pointer");
  RECMV_REQUIRE(lda >= M && ldb >= N && ldc >= N, "gemm_tn: leading dimension too small");
  RECMV_REQUIRE(M < (1 << 20) && N < (1 << 20), "gemm_tn: output too large");
  hipStream_t s = (hipStream_t)stream;
  if (K == 0) {
    for (int64_t m = 0; m < M; ++m) RECMV_HIP_TRY(hipMemsetAsync(C + m * ldc, 0, N * 4, s));
    return RECMV_OK;
  }
  const int splits = tn_splits(M, N, K);
  const int64_t need = (int64_t)splits * M * N * 4;
  if (!workspace || workspace_bytes < need) {
    set_error("gemm_tn: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    return RECMV_ERR_WORKSPACE;
  }
  static bool attr_set = false;
  if (!attr_set) {
    RECMV_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kTnLds));
    RECMV_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kTnLds));
    attr_set = true;
  }
  const int nbm = (int)ceil_div(M, BM), nbn = (int)ceil_div(N, BN);
  int64_t kchunk = ceil_div(ceil_div(K, splits), BK) * BK;
  const bool a_vec = aligned16(A) && lda % 4 == 0, b_vec = aligned16(B) && ldb % 4 == 0;
  int rc;
  {                          // the events bracket the product kernel alone (slot 8 = one kernel symbol); its reduction pass follows
  ScopedLaunchTimer timer(8, (double)M, (double)N, (double)K, s);
  const bool bf3_tn = g_gemm_mode == 1 && (g_b3_families & 4);
  if (!bf3_tn && tn_occ() && a_vec && b_vec && lda >= ((M + 3) & ~3ll) && ldb >= ((N + 3) & ~3ll) && M >= 4 && N >= 4) {
    kchunk = ceil_div(ceil_div(K, splits), 16) * 16;
    hipLaunchKernelGGL(gemm_tn_occ_kernel<16>, dim3((unsigned)(nbm * nbn * splits)), dim3(kBlk), 2 * 16 * LDM * 4, s, A, lda,
                       B, ldb, (float*)workspace, (int)M, (int)N, K, nbm, nbn, kchunk);
  } else if (bf3_tn)
    hipLaunchKernelGGL(gemm_tn_kernel<true>, dim3((unsigned)(nbm * nbn * splits)), dim3(kBlk), kTnLds, s, A, lda, B,
                       ldb, (float*)workspace, (int)M, (int)N, K, nbm, nbn, kchunk, a_vec, b_vec);
  else
    hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3((unsigned)(nbm * nbn * splits)), dim3(kBlk), kTnLds, s, A, lda, B,
                       ldb, (float*)workspace, (int)M, (int)N, K, nbm, nbn, kchunk, a_vec, b_vec);
  rc = check_launch("gemm_tn");
  }
  if (rc) return rc;
  if (N % 4 == 0 && ldc % 4 == 0 && aligned16(C) && aligned16(workspace))
    hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(stream_grid(M * N / 4, kBlk)), dim3(kBlk), 0, s,
                       (const float4*)workspace, C, ldc, (int)M, (int)N, splits);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(stream_grid(M * N, kBlk)), dim3(kBlk), 0, s,
                       (const float*)workspace, C, ldc, (int)M, (int)N, splits);
  return check_launch("gemm_tn/reduce");
}

extern "C" int recmv_posenc_forward(const float* x, int64_t ldx, float* out, int64_t ldo, int64_t ldo_fill,
                                    int64_t P, int L, const float* weights_host, float out_scale,
                                    void* stream) {
  RECMV_REQUIRE(P >= 0 && L >= 0 && L <= 16, "posenc: bad size (P=%lld, L=%d)", (long long)P, L);
  if (P == 0) return RECMV_OK;
  RECMV_REQUIRE(x && out, "posenc: NULL pointer");
  RECMV_REQUIRE(ldx >= 3 && ldo >= 3 + 6 * L && ldo_fill <= ldo, "posenc: leading dimension too small");
  hipStream_t s = (hipStream_t)stream;
  // the 2L annealing weights are python floats in the reference (utils/utils.py:40-46); ship them by value
  PeWeights hw;
  for (int i = 0; i < 32; ++i) hw.w[i] = (weights_host && i < 2 * L) ? weights_host[i] : 1.f;
  hipLaunchKernelGGL(posenc_kernel, dim3(stream_grid(P * (1 + 2 * L), kBlk)), dim3(kBlk), 0, s, x, ldx, out, ldo,
                     ldo_fill, P, L, hw, out_scale);
  return check_launch("posenc");
}

// Per-launch HIP-event timing of the MFMA kernels.  recmv_profile_begin() starts recording (events on the launch
// stream around every gemm_nt / gemm_tn launch); recmv_profile_end() waits for the recorded events and returns, per
// kernel variant v (0..7: gemm_nt_kernel<T, FAST, AMUL> with v = (T-1) + 2*FAST + 4*AMUL; 8: gemm_tn_occ_kernel / gemm_tn_kernel
// (the product alone, its split-K reduction pass is not bracketed); 9 / 10: gemm_nt_occ_kernel<false, ...> with 128x128 / 64x128 tiles, 11: gemm_nt_occ_kernel<true, ...>; 12 / 13: gemm_nt_narrow_kernel<true, false, .> / its other instantiations), out[5v] = timed launches, out[5v+1] = their summed duration in seconds, out[5v+2] = their
// summed algorithmic FLOP (2 M N K), out[5v+3] / out[5v+4] = launches / FLOP of the launches below `min_flops`, which
// are only counted (bracketing tens of thousands of ~20 us launches with events would perturb the run being timed).
extern "C" int recmv_profile_begin(double min_flops) {
  g_prof.recs.clear();
  for (auto& u : g_prof.untimed) u[0] = u[1] = 0.0;
  g_prof.min_flops = min_flops;
  g_prof.on = true;
  return RECMV_OK;
}

// Algorithmic bytes (4 (M K + N K + M N): both operands read once, the result written once) of the launches the last
// recmv_profile_end bracketed, per variant with the same slot folding as its `out`.  (ABI v7)
extern "C" int recmv_profile_bytes(double* out, int n_variants) {
  RECMV_REQUIRE(out && n_variants >= 9, "profile_bytes: need room for 9 variants");
  for (int i = 0; i < n_variants; ++i) out[i] = 0.0;
  auto slot = [&](int v) { return v < n_variants ? v : (v == 11 ? 7 : (v == 12 ? 2 : (v == 13 ? 6 : 3))); };
  for (int v = 0; v < 14; ++v) out[slot(v)] += g_prof.timed_bytes[v];
  return RECMV_OK;
}

// The launches of >= 4 GFLOP among those the last recmv_profile_end bracketed (a pass that brackets EVERY launch still yields the
// large products' own rate): out[3v] = launches, out[3v+1] = seconds, out[3v+2] = FLOP, same slots as recmv_profile_end.  (ABI v7)
extern "C" int recmv_profile_large(double* out, int n_variants) {
  RECMV_REQUIRE(out && n_variants >= 9, "profile_large: need room for 9 variants");
  for (int i = 0; i < 3 * n_variants; ++i) out[i] = 0.0;
  auto slot = [&](int v) { return v < n_variants ? v : (v == 11 ? 7 : (v == 12 ? 2 : (v == 13 ? 6 : 3))); };
  for (int v = 0; v < 14; ++v)
    for (int q = 0; q < 3; ++q) out[3 * slot(v) + q] += g_prof.large[v][q];
  return RECMV_OK;
}

extern "C" int recmv_profile_busy(double* out2) {
  RECMV_REQUIRE(out2, "profile_busy: NULL");
  out2[0] = g_prof.busy_union_s;
  out2[1] = g_prof.busy_span_s;
  return RECMV_OK;
}

extern "C" int recmv_profile_end(double* out, int n_variants) {
  g_prof.on = false;
  RECMV_REQUIRE(out && n_variants >= 9, "profile_end: need room for 9 variants");
  for (int i = 0; i < 5 * n_variants; ++i) out[i] = 0.0;
  // slots 9..11 (the high-occupancy NT kernels) fold into the slots of the kernels they replace for a caller with 9 slots
  auto slot = [&](int v) { return v < n_variants ? v : (v == 11 ? 7 : (v == 12 ? 2 : (v == 13 ? 6 : 3))); };
  for (int v = 0; v < 14; ++v) {
    out[5 * slot(v) + 3] += g_prof.untimed[v][0];
    out[5 * slot(v) + 4] += g_prof.untimed[v][1];
  }
  // every bracket also as an interval on one time axis (the first recorded event is the origin; events of different streams of a
  // device share a clock): the union of the intervals is the time in which at least one bracketed MFMA kernel was running
  std::vector<std::pair<double, double>> iv;
  iv.reserve(g_prof.recs.size());
  hipEvent_t origin = g_prof.recs.empty() ? nullptr : g_prof.recs.front().a;
  for (auto& b : g_prof.timed_bytes) b = 0.0;
  for (auto& l : g_prof.large) l[0] = l[1] = l[2] = 0.0;
  for (auto& r : g_prof.recs) {
    g_prof.timed_bytes[r.variant] += r.bytes;
    RECMV_HIP_TRY(hipEventSynchronize(r.b));
    float ms = 0.f;
    RECMV_HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
    if (r.flops >= 4.0e9) {
      g_prof.large[r.variant][0] += 1.0;
      g_prof.large[r.variant][1] += (double)ms * 1e-3;
      g_prof.large[r.variant][2] += r.flops;
    }
    out[5 * slot(r.variant) + 0] += 1.0;
    out[5 * slot(r.variant) + 1] += (double)ms * 1e-3;
    out[5 * slot(r.variant) + 2] += r.flops;
    float t0 = 0.f;
    if (r.a != origin && hipEventElapsedTime(&t0, origin, r.a) != hipSuccess) t0 = -1.f;
    if (t0 >= 0.f) iv.emplace_back((double)t0 * 1e-3, (double)(t0 + ms) * 1e-3);
  }
  for (auto& r : g_prof.recs) {
    g_prof.pool.push_back(r.a);
    g_prof.pool.push_back(r.b);
  }
  g_prof.recs.clear();
  g_prof.busy_union_s = g_prof.busy_span_s = 0.0;
  if (!iv.empty()) {
    std::sort(iv.begin(), iv.end());
    double lo = iv[0].first, hi = iv[0].second, first = iv[0].first, last = iv[0].second;
    for (size_t i = 1; i < iv.size(); ++i) {
      if (iv[i].first > hi) {
        g_prof.busy_union_s += hi - lo;
        lo = iv[i].first;
        hi = iv[i].second;
      } else if (iv[i].second > hi) {
        hi = iv[i].second;
      }
      if (iv[i].second > last) last = iv[i].second;
    }
    g_prof.busy_union_s += hi - lo;
    g_prof.busy_span_s = last - first;
  }
  return RECMV_OK;
}
