// Why does a per-thread, atomics-free kernel (csrc/def_regu.hip) give different bits for the same input when the bf16x6 mode's NT
// product kernels run beside it on other streams?  (tools/def_regu_stress.py: a few launches in a hundred, always lanes 48..63 of a
// wave, errors up to the size of the values.)  Variants of the kernel, each launched `iters` times on the same input while two other
// streams run recmv_gemm_nt (bf16x6, 64 x 64 tiles); a launch is "bad" when its output differs from the first launch's.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_disturb_repro tools/valu_disturb_repro.hip -Iinclude -Lrec-mv_amd/lib -lrecmv_hip -Wl,-rpath,$PWD/rec-mv_amd/lib
//   tools/bin/valu_disturb_repro [iters=300]
//
//   V0  the product's kernel: LDS staging of the 3x3 matrices + libm sqrtf / logf / division
//   V1  the same arithmetic, global loads and stores only (no LDS)
//   V2  LDS staging, raw hardware transcendentals (v_rcp / v_sqrt / v_log) instead of libm's refined sequences
//   V3  V2 with `s_nop 7` tied to every transcendental's result
//   V4  no transcendental at all: a dependent chain of FMAs per thread, LDS staging
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "recmv_hip.h"

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e__ = (x);                                                                  \
    if (e__ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)

constexpr int kBlk = 256;

// FMA-only stand-ins (V6): no v_rcp / v_sqrt / v_rsq / v_log anywhere — integer-seeded Newton iterations and an atanh series
__device__ __forceinline__ float n_rcp(float x) {
  float r = __uint_as_float(0x7ef311c7u - __float_as_uint(fabsf(x)));
#pragma unroll
  for (int i = 0; i < 4; ++i) r = r * (2.f - fabsf(x) * r);
  return __uint_as_float(__float_as_uint(r) | (__float_as_uint(x) & 0x80000000u));
}
__device__ __forceinline__ float n_rsqrt(float x) {
  float r = __uint_as_float(0x5f3759dfu - (__float_as_uint(x) >> 1));
#pragma unroll
  for (int i = 0; i < 4; ++i) r = r * (1.5f - 0.5f * x * r * r);
  return r;
}
__device__ __forceinline__ float n_log(float x) {       // x > 0, normal
  const unsigned u = __float_as_uint(x);
  const float e = (float)((int)(u >> 23) - 127);
  const float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);      // [1, 2)
  const float t = (m - 1.f) * n_rcp(m + 1.f), t2 = t * t;
  const float p = t * (2.f + t2 * (0.66666667f + t2 * (0.4f + t2 * (0.28571429f + t2 * 0.22222222f))));
  return e * 0.6931471805599453f + p;
}

template <int V>
__device__ __forceinline__ float t_rcp(float x) {
  if (V == 6) return n_rcp(x);
  if (V == 2 || V == 5) return __builtin_amdgcn_rcpf(x);
  float r = __builtin_amdgcn_rcpf(x);
  asm volatile("s_nop 7" : "+v"(r));
  return r;
}
template <int V>
__device__ __forceinline__ float t_sqrt(float x) {
  if (V == 6) return x * n_rsqrt(x);
  if (V == 2 || V == 5) return __builtin_amdgcn_sqrtf(x);
  float r = __builtin_amdgcn_sqrtf(x);
  asm volatile("s_nop 7" : "+v"(r));
  return r;
}
template <int V>
__device__ __forceinline__ float t_log(float x) {
  if (V == 6) return n_log(x);
  if (V == 2 || V == 5) return __builtin_amdgcn_logf(x) * 0.6931471805599453f;
  float r = __builtin_amdgcn_logf(x);
  asm volatile("s_nop 7" : "+v"(r));
  return r * 0.6931471805599453f;
}

template <int V, int p, int q>
__device__ __forceinline__ void rotate(float (&b)[3][3], float (&v)[3][3]) {
  const float alpha = b[0][p] * b[0][p] + b[1][p] * b[1][p] + b[2][p] * b[2][p];
  const float beta = b[0][q] * b[0][q] + b[1][q] * b[1][q] + b[2][q] * b[2][q];
  const float gamma = b[0][p] * b[0][q] + b[1][p] * b[1][q] + b[2][p] * b[2][q];
  if (V != 5) {
    if (fabsf(gamma) < 1e-37f) return;
  }
  float zeta, t, c;
  if (V <= 1) {
    zeta = (beta - alpha) / (2.f * gamma);
    t = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(zeta * zeta + 1.f));
    c = 1.f / sqrtf(t * t + 1.f);
  } else {
    zeta = (beta - alpha) * t_rcp<V>(V == 5 ? 2.f * gamma + copysignf(1e-30f, gamma) : 2.f * gamma);
    t = (zeta >= 0.f ? 1.f : -1.f) * t_rcp<V>(fabsf(zeta) + t_sqrt<V>(zeta * zeta + 1.f));
    c = t_rcp<V>(t_sqrt<V>(t * t + 1.f));
  }
  const float s = t * c;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float bp = b[k][p], bq = b[k][q];
    b[k][p] = c * bp - s * bq;
    b[k][q] = s * bp + c * bq;
    const float vp = v[k][p], vq = v[k][q];
    v[k][p] = c * vp - s * vq;
    v[k][q] = s * vp + c * vq;
  }
}

template <int V>
__global__ __launch_bounds__(kBlk) void regu_variant(const float* __restrict__ J, long P, float inv_c2, float* __restrict__ y,
                                                     float* __restrict__ gJ) {
  __shared__ float tile[kBlk * 9];
  const long base = (long)blockIdx.x * kBlk;
  const int n = (int)((P - base) < kBlk ? (P - base) : kBlk);
  const int t = threadIdx.x;
  if (V != 1) {
    for (int e = threadIdx.x; e < n * 9; e += kBlk) tile[e] = J[base * 9 + e];
    __syncthreads();
  }
  float g[9];
  if (t < n) {
    float b[3][3], v[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        b[i][j] = V != 1 ? tile[t * 9 + 3 * i + j] : J[(base + t) * 9 + 3 * i + j];
        v[i][j] = i == j ? 1.f : 0.f;
      }
    if (V == 4) {
      // no transcendental: 60 dependent FMAs per entry
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float z = b[i][j];
#pragma unroll
          for (int r = 0; r < 60; ++r) z = fmaf(z, 0.99f, b[(i + r) % 3][(j + r / 3) % 3] * 0.01f);
          g[3 * i + j] = z;
          acc += z;
        }
      y[base + t] = acc;
    } else {
#pragma unroll
      for (int sweep = 0; sweep < 5; ++sweep) {
        rotate<V, 0, 1>(b, v);
        rotate<V, 0, 2>(b, v);
        rotate<V, 1, 2>(b, v);
      }
      float x = 0.f, w[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float lam = b[0][i] * b[0][i] + b[1][i] * b[1][i] + b[2][i] * b[2][i];
        const bool live = lam > 1e-20f;
        const float lc = live ? lam : 1e-20f;
        const float s = 0.5f * (V <= 1 ? logf(lc) : t_log<V>(lc));
        x += s * s;
        w[i] = live ? (V <= 1 ? s / lc : s * t_rcp<V>(lc)) : 0.f;
      }
      const float u = x * inv_c2;
      const float d = V <= 1 ? 1.f / (u + 4.f) : t_rcp<V>(u + 4.f);
      y[base + t] = 2.f * u * d;
      const float k2 = 2.f * 8.f * inv_c2 * d * d;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g[3 * i + j] = k2 * (b[i][0] * w[0] * v[j][0] + b[i][1] * w[1] * v[j][1] + b[i][2] * w[2] * v[j][2]);
    }
  }
  if (V == 1) {
    if (t < n)
#pragma unroll
      for (int e = 0; e < 9; ++e) gJ[(base + t) * 9 + e] = g[e];
    return;
  }
  __syncthreads();
  if (t < n) {
#pragma unroll
    for (int e = 0; e < 9; ++e) tile[t * 9 + e] = g[e];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n * 9; e += kBlk) gJ[base * 9 + e] = tile[e];
}


// ---- aggressors: one instruction family each, on two side streams -------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// A0: v_cvt_pk_bf16_f32 only     A1: bf16 MFMA only     A2: both interleaved (the split-in-loop kernels)     A3: f32 MFMA only
// A4: transcendentals (v_exp_f32) only     A5: software f32 -> bf16 rounding (integer ops) + bf16 MFMA
template <int A>
__global__ __launch_bounds__(256) void aggressor(float* __restrict__ out, int rounds) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float x0 = 1.0f + 1e-3f * (tid & 1023), x1 = 0.5f + 1e-3f * (tid & 511);
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  unsigned h = 0x3f803f80u;
  for (int i = 0; i < rounds; ++i) {
    if (A == 0 || A == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x2_t v = {x0, x1};
        h ^= __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        x0 = x0 * 1.0001f + 0.25f;
        x1 = x1 * 0.9999f + 0.125f;
      }
    }
    if (A == 5) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        unsigned a = __float_as_uint(x0), b = __float_as_uint(x1);
        a += 0x7fffu + ((a >> 16) & 1u);
        b += 0x7fffu + ((b >> 16) & 1u);
        h ^= (a >> 16) | (b & 0xffff0000u);
        x0 = x0 * 1.0001f + 0.25f;
        x1 = x1 * 0.9999f + 0.125f;
      }
    }
    if (A == 1 || A == 2 || A == 5) {
      const bf16x8_t f = __builtin_bit_cast(bf16x8_t, (u32x4_t){h, h ^ 0x00010001u, h, h ^ 0x00020002u});
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, f, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, f, acc, 0, 0, 0);
    }
    if (A == 3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, x0, acc, 0, 0, 0);
      x0 = x0 * 1.0001f + 0.25f;
    }
    if (A == 4) {
#pragma unroll
      for (int u = 0; u < 8; ++u) x0 = __builtin_amdgcn_exp2f(x0 * 0.001f) + x1;
    }
  }
  float sum = x0 + x1 + __uint_as_float(h & 0x3fffffffu);
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += acc[r];
  out[tid] = sum;
}
// A6..A11: the same with LDS traffic, as the product kernels have it.  W = bytes per ds_read (16: ds_read_b128 as in the NT kernels, 4:
// ds_read_b32 as in the TN kernel); CVT: split the values read into bf16 with v_cvt_pk_bf16_f32; MF: 0 none, 1 bf16 MFMA, 2 f32 MFMA;
// BAR: a workgroup barrier + LDS rewrite per round (the K-tile loop's shape)
template <int W, bool CVT, int MF, bool BAR>
__global__ __launch_bounds__(256) void aggressor_lds(float* __restrict__ out, int rounds) {
  __shared__ __attribute__((aligned(16))) float sm[8192];          // 32 KB
  const int tid = threadIdx.x;
  for (int e = tid; e < 8192; e += 256) sm[e] = 1.0f + 1e-4f * e;
  __syncthreads();
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float keep = 0.f;
  unsigned h = 0x3f803f80u;
  for (int i = 0; i < rounds; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float4 a, b;
      const int o = ((tid * 8 + u * 2048 + i * 64) & 8191) & ~7;
      if (W == 16) {
        a = *reinterpret_cast<const float4*>(sm + o);
        b = *reinterpret_cast<const float4*>(sm + o + 4);
      } else {
        a = make_float4(sm[o], sm[(o + 36) & 8191], sm[(o + 72) & 8191], sm[(o + 108) & 8191]);
        b = make_float4(sm[(o + 144) & 8191], sm[(o + 180) & 8191], sm[(o + 216) & 8191], sm[(o + 252) & 8191]);
      }
      u32x4_t f;
      if (CVT) {
        f[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a.x, a.y}, bf16x2_t));
        f[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a.z, a.w}, bf16x2_t));
        f[2] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){b.x, b.y}, bf16x2_t));
        f[3] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){b.z, b.w}, bf16x2_t));
      } else {
        f[0] = __float_as_uint(a.x) ^ h;
        f[1] = __float_as_uint(a.z);
        f[2] = __float_as_uint(b.x);
        f[3] = __float_as_uint(b.z);
        keep += a.y + a.w + b.y + b.w;
      }
      if (MF == 1) {
        const bf16x8_t fr = __builtin_bit_cast(bf16x8_t, f);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr, fr, acc, 0, 0, 0);
      } else if (MF == 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        h ^= f[1];
      } else {
        h ^= f[0] ^ f[1] ^ f[2] ^ f[3];
      }
    }
    if (BAR) {
      __syncthreads();
      sm[(tid * 4 + i) & 8191] = keep + 1.0f;
      __syncthreads();
    }
  }
  float sum = keep + __uint_as_float(h & 0x3fffffffu);
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += acc[r];
  out[blockIdx.x * 256 + tid] = sum;
}
// A14..: the inner loop of gemm_nt_kernel<1, FAST, false, BF3> as it is (LDS tile of f32 rows, two ds_read_b128 per operand row and
// 16-column step, the three-way bf16 split of csrc/gemm_f32.hip split8, six dependent bf16 MFMAs per 32 x 32 tile), without the
// global traffic.  SPLIT: 0 = v_cvt_pk_bf16_f32 (the product's), 1 = integer round-to-nearest-even;  MF: 0 = pieces only, no MFMA.
struct Pcs {
  bf16x8_t h, m, l;
};
template <int SPLIT>
__device__ __forceinline__ void split2_(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  auto cvt = [](float a, float b) -> unsigned {
    if (SPLIT == 0) return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua += 0x7fffu + ((ua >> 16) & 1u);
    ub += 0x7fffu + ((ub >> 16) & 1u);
    return (ua >> 16) | (ub & 0xffff0000u);
  };
  h = cvt(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt(r0, r1);
  const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt(q0, q1);
}
template <int SPLIT>
__device__ __forceinline__ Pcs split8_(float4 a, float4 b) {
  unsigned h[4], m[4], l[4];
  split2_<SPLIT>(a.x, a.y, h[0], m[0], l[0]);
  split2_<SPLIT>(a.z, a.w, h[1], m[1], l[1]);
  split2_<SPLIT>(b.x, b.y, h[2], m[2], l[2]);
  split2_<SPLIT>(b.z, b.w, h[3], m[3], l[3]);
  Pcs p;
  p.h = __builtin_bit_cast(bf16x8_t, (u32x4_t){h[0], h[1], h[2], h[3]});
  p.m = __builtin_bit_cast(bf16x8_t, (u32x4_t){m[0], m[1], m[2], m[3]});
  p.l = __builtin_bit_cast(bf16x8_t, (u32x4_t){l[0], l[1], l[2], l[3]});
  return p;
}
template <int SPLIT, int MF>
__global__ __launch_bounds__(256) void aggressor_loop(float* __restrict__ out, int rounds) {
  constexpr int LDK = 36;
  __shared__ __attribute__((aligned(16))) float As[2 * 64 * LDK], Bs[2 * 64 * LDK];       // 36 KB, as the kernel's
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 2 * 64 * LDK; e += 256) {
    As[e] = 1.0f + 1e-4f * e;
    Bs[e] = 0.5f - 1e-4f * e;
  }
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31), khalf = (lane >> 5) * 4;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  unsigned sink = 0;
  for (int kt = 0; kt < rounds; ++kt) {
    const int buf = kt & 1;
    const float* as = As + (buf * 64 + arow) * LDK + 2 * khalf;
    const float* bs = Bs + (buf * 64 + brow) * LDK + 2 * khalf;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const Pcs pa = split8_<SPLIT>(*reinterpret_cast<const float4*>(as + ks * 16), *reinterpret_cast<const float4*>(as + ks * 16 + 4));
      const Pcs pb = split8_<SPLIT>(*reinterpret_cast<const float4*>(bs + ks * 16), *reinterpret_cast<const float4*>(bs + ks * 16 + 4));
      if (MF) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.h, pb.l, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.l, pb.h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.m, pb.m, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.h, pb.m, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.m, pb.h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.h, pb.h, acc, 0, 0, 0);
      } else {
        const u32x4_t a = __builtin_bit_cast(u32x4_t, pa.l), b = __builtin_bit_cast(u32x4_t, pb.l);
        sink ^= a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3];
      }
    }
    __syncthreads();
    As[(tid * 4 + kt) % (2 * 64 * LDK)] += 1e-6f;
    __syncthreads();
  }
  float sum = __uint_as_float(sink & 0x3fffffffu);
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += acc[r];
  out[blockIdx.x * 256 + tid] = sum;
}
// A18 / A19: the K loop above FED FROM GLOBAL MEMORY like the product kernel's (every round: two float4 per thread from a large buffer
// into registers while the previous tile is multiplied, then into LDS) — bf16 MFMA (A18) or f32 MFMA (A19)
template <int MF>
__global__ __launch_bounds__(256) void aggressor_gl(float* __restrict__ out, int rounds, const float* __restrict__ src, long src_floats) {
  constexpr int LDK = 36;
  __shared__ __attribute__((aligned(16))) float As[2 * 64 * LDK], Bs[2 * 64 * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31), khalf = (lane >> 5) * 4;
  const int row = tid >> 3, c4 = tid & 7;            // staging map of gemm_nt_kernel<1, ...>: 64 rows x 8 float4, two per thread
  long base = ((long)blockIdx.x * 64 * 4096) % (src_floats - 64 * 4096 - 8192);
  const float* pa = src + base + (long)row * 4096 + c4 * 4;
  const float* pb = src + ((base + 32 * 4096) % (src_floats - 64 * 4096 - 8192)) + (long)row * 4096 + c4 * 4;
  float4 ra[2], rb[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      ra[r] = *reinterpret_cast<const float4*>(pa + (long)r * 32 * 4096 + k0);
      rb[r] = *reinterpret_cast<const float4*>(pb + (long)r * 32 * 4096 + k0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      *reinterpret_cast<float4*>(As + (buf * 64 + row + 32 * r) * LDK + c4 * 4) = ra[r];
      *reinterpret_cast<float4*>(Bs + (buf * 64 + row + 32 * r) * LDK + c4 * 4) = rb[r];
    }
  };
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < rounds; ++kt) {
    const int buf = kt & 1;
    gload(((kt + 1) * 32) & 4095);
    const float* as = As + (buf * 64 + arow) * LDK + 2 * khalf;
    const float* bs = Bs + (buf * 64 + brow) * LDK + 2 * khalf;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (MF == 1) {
        const Pcs pa_ = split8_<0>(*reinterpret_cast<const float4*>(as + ks * 16), *reinterpret_cast<const float4*>(as + ks * 16 + 4));
        const Pcs pb_ = split8_<0>(*reinterpret_cast<const float4*>(bs + ks * 16), *reinterpret_cast<const float4*>(bs + ks * 16 + 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa_.h, pb_.l, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa_.l, pb_.h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa_.m, pb_.m, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa_.h, pb_.m, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa_.m, pb_.h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa_.h, pb_.h, acc, 0, 0, 0);
      } else {
        const float4 a0 = *reinterpret_cast<const float4*>(as + ks * 16), b0 = *reinterpret_cast<const float4*>(bs + ks * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
      }
    }
    lstore(buf ^ 1);
    __syncthreads();
  }
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += acc[r];
  out[blockIdx.x * 256 + tid] = sum;
}
typedef void (*Agg)(float*, int);


// ---- bisecting the victim: which part of the regulariser's code is it?
//   B0  the 15 rotations (raw transcendentals, selects, early-out) applied to b and v; outputs = sums of b and v (no log tail)
//   B1  15 rotations with CONSTANT c, s: only the column updates (the packed f32 multiplies / FMAs on 18 live registers)
//   B2  the 15 (c, s) computations (dot products, v_rcp, v_sqrt, compare + select) WITHOUT applying them; outputs their sum
//   B3  B1 with the updates written so that the compiler cannot pair them (scalar v_mul / v_fma only: a volatile-asm barrier per value)
template <int B>
__global__ __launch_bounds__(kBlk) void bisect_variant(const float* __restrict__ J, long P, float inv_c2, float* __restrict__ y,
                                                       float* __restrict__ gJ) {
  const long i0 = (long)blockIdx.x * kBlk + threadIdx.x;
  if (i0 >= P) return;
  float b[3][3], v[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      b[i][j] = J[i0 * 9 + 3 * i + j];
      v[i][j] = i == j ? 1.f : 0.f;
    }
  float acc = 0.f;
#pragma unroll
  for (int sweep = 0; sweep < 5; ++sweep) {
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      float c = 0.8f, s_ = 0.6f;
      if (B == 0 || B == 2) {
        const float alpha = b[0][p] * b[0][p] + b[1][p] * b[1][p] + b[2][p] * b[2][p];
        const float beta = b[0][q] * b[0][q] + b[1][q] * b[1][q] + b[2][q] * b[2][q];
        const float gamma = b[0][p] * b[0][q] + b[1][p] * b[1][q] + b[2][p] * b[2][q];
        const bool skip = fabsf(gamma) < 1e-37f;
        const float zeta = (beta - alpha) * __builtin_amdgcn_rcpf(2.f * gamma);
        const float t = (zeta >= 0.f ? 1.f : -1.f) * __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(zeta * zeta + 1.f));
        c = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(t * t + 1.f));
        s_ = t * c;
        if (skip) {
          c = 1.f;
          s_ = 0.f;
        }
        acc += c + s_;
      }
      if (B != 2) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float bp = b[k][p], bq = b[k][q], vp = v[k][p], vq = v[k][q];
          if (B == 3) {
            asm volatile("" : "+v"(bp));
            asm volatile("" : "+v"(bq));
            asm volatile("" : "+v"(vp));
            asm volatile("" : "+v"(vq));
          }
          float n0 = c * bp - s_ * bq, n1 = s_ * bp + c * bq, n2 = c * vp - s_ * vq, n3 = s_ * vp + c * vq;
          if (B == 3) {
            asm volatile("" : "+v"(n0));
            asm volatile("" : "+v"(n1));
            asm volatile("" : "+v"(n2));
            asm volatile("" : "+v"(n3));
          }
          b[k][p] = n0;
          b[k][q] = n1;
          v[k][p] = n2;
          v[k][q] = n3;
        }
      } else {
        b[0][p] += 1e-3f * c;       // keep the inputs of the next (c, s) moving
        b[1][q] -= 1e-3f * s_;
      }
    }
  }
  float sb = acc;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      sb += b[i][j];
      gJ[i0 * 9 + 3 * i + j] = b[i][j] + 2.f * v[i][j];
    }
  y[i0] = sb * inv_c2;
}

typedef void (*Kern)(const float*, long, float, float*, float*);

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 300;
  const long P = 30714;
  std::vector<float> hJ(P * 9);
  unsigned s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) * (1.f / 16777216.f)) - 0.5f;
  };
  for (long i = 0; i < P; ++i)
    for (int e = 0; e < 9; ++e) hJ[i * 9 + e] = (e % 4 == 0 ? 1.f : 0.f) + 0.06f * rnd();
  float *J, *y, *g, *y0, *g0;
  CK(hipMalloc(&J, P * 36));
  CK(hipMalloc(&y, P * 4));
  CK(hipMalloc(&g, P * 36));
  CK(hipMalloc(&y0, P * 4));
  CK(hipMalloc(&g0, P * 36));
  CK(hipMemcpy(J, hJ.data(), P * 36, hipMemcpyHostToDevice));
  // the disturbing work: bf16x6 products with 64 x 64 tiles (12 000 x 512 x 512) on two side streams
  const long M = 12000, N = 512, K = 512;
  float *A, *B, *C[2];
  CK(hipMalloc(&A, M * K * 4));
  CK(hipMalloc(&B, N * K * 4));
  std::vector<float> hA(M * K), hB(N * K);
  for (auto& v : hA) v = rnd();
  for (auto& v : hB) v = rnd() * 0.05f;
  CK(hipMemcpy(A, hA.data(), M * K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hB.data(), N * K * 4, hipMemcpyHostToDevice));
  hipStream_t main_s, side[2];
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    CK(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
    CK(hipMalloc(&C[i], M * N * 4));
  }
  Kern kerns[7] = {regu_variant<0>, regu_variant<1>, regu_variant<2>, regu_variant<3>, regu_variant<4>, regu_variant<5>, regu_variant<6>};
  const char* names[7] = {"V0 LDS + libm (the product's kernel)", "V1 no LDS, libm", "V2 LDS, raw v_rcp/v_sqrt/v_log",
                          "V3 LDS, raw transcendentals + s_nop 7 behind each", "V4 LDS, FMA chain, no transcendental",
                          "V5 = V2 without the divergent early-out", "V6 = V2's control flow, FMA-only rcp / sqrt / log"};
  std::vector<float> hy(P), hy0(P), hg(P * 9), hg0(P * 9);
  for (int mode = 1; mode >= 0; --mode) {
    recmv_set_gemm_mode(mode);
    for (int busy = 0; busy < 2; ++busy)
      for (int v = 0; v < 7; ++v) {
        hipLaunchKernelGGL(kerns[v], dim3((unsigned)((P + kBlk - 1) / kBlk)), dim3(kBlk), 0, main_s, J, P, 1111.f, y0, g0);
        CK(hipStreamSynchronize(main_s));
        CK(hipMemcpy(hy0.data(), y0, P * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hg0.data(), g0, P * 36, hipMemcpyDeviceToHost));
        int bad = 0;
        long first_lane = -1, nbad_elems = 0;
        for (int it = 0; it < iters; ++it) {
          if (busy)
            for (int i = 0; i < 2; ++i)
              for (int r = 0; r < 2; ++r)
                if (recmv_gemm_nt(A, K, B, K, nullptr, C[i], N, M, N, K, RECMV_ACT_RELU, 0.f, 1.f, side[i]) != 0) {
                  fprintf(stderr, "gemm_nt: %s\n", recmv_last_error());
                  return 2;
                }
          hipLaunchKernelGGL(kerns[v], dim3((unsigned)((P + kBlk - 1) / kBlk)), dim3(kBlk), 0, main_s, J, P, 1111.f, y, g);
          CK(hipMemcpyAsync(hy.data(), y, P * 4, hipMemcpyDeviceToHost, main_s));
          CK(hipMemcpyAsync(hg.data(), g, P * 36, hipMemcpyDeviceToHost, main_s));
          CK(hipStreamSynchronize(main_s));
          bool diff = false;
          for (long i = 0; i < P; ++i) {
            bool d = memcmp(&hy[i], &hy0[i], 4) != 0 || memcmp(&hg[i * 9], &hg0[i * 9], 36) != 0;
            if (d) {
              diff = true;
              ++nbad_elems;
              if (first_lane < 0) first_lane = i % 64;
            }
          }
          bad += diff;
        }
        CK(hipDeviceSynchronize());
        printf("%-7s products %-5s  %-52s %3d of %d launches differ from the first (%ld matrices in all; lane of the first: %ld)\n",
               mode ? "bf16x6" : "f32", busy ? "busy" : "idle", names[v], bad, iters, nbad_elems, first_lane);
        fflush(stdout);
      }
  }
  if (argc > 2) {                   // (variants only) + the bisect kernels beside the bf16x6 products
    recmv_set_gemm_mode(1);
    Kern bk[4] = {bisect_variant<0>, bisect_variant<1>, bisect_variant<2>, bisect_variant<3>};
    const char* bn[4] = {"B0 15 rotations (c, s from rcp / sqrt / select) applied, no log tail", "B1 15 rotations with constant c, s (column updates only)",
                         "B2 15 (c, s) computations, not applied", "B3 = B1 with unpaired scalar multiplies / FMAs"};
    for (int v = 0; v < 4; ++v) {
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(bk[v], dim3((unsigned)((P + kBlk - 1) / kBlk)), dim3(kBlk), 0, main_s, J, P, 1111.f, y0, g0);
      CK(hipStreamSynchronize(main_s));
      CK(hipMemcpy(hy0.data(), y0, P * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hg0.data(), g0, P * 36, hipMemcpyDeviceToHost));
      int bad = 0;
      long nbad = 0, first_lane = -1;
      for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < 2; ++i)
          for (int r = 0; r < 2; ++r) recmv_gemm_nt(A, K, B, K, nullptr, C[i], N, M, N, K, RECMV_ACT_RELU, 0.f, 1.f, side[i]);
        hipLaunchKernelGGL(bk[v], dim3((unsigned)((P + kBlk - 1) / kBlk)), dim3(kBlk), 0, main_s, J, P, 1111.f, y, g);
        CK(hipMemcpyAsync(hy.data(), y, P * 4, hipMemcpyDeviceToHost, main_s));
        CK(hipMemcpyAsync(hg.data(), g, P * 36, hipMemcpyDeviceToHost, main_s));
        CK(hipStreamSynchronize(main_s));
        bool diff = false;
        for (long i = 0; i < P; ++i)
          if (memcmp(&hy[i], &hy0[i], 4) != 0 || memcmp(&hg[i * 9], &hg0[i * 9], 36) != 0) {
            diff = true;
            ++nbad;
            if (first_lane < 0) first_lane = i % 64;
          }
        bad += diff;
      }
      printf("bisect  %-76s %3d of %d launches differ (%ld matrices; lane of the first: %ld)\n", bn[v], bad, iters, nbad, first_lane);
      fflush(stdout);
    }
    // ---- a library-free aggressor?  The K loop fed from global memory, against the packed-f32 victim (B1) and the regulariser (V0)
    {
      float* aout;
      CK(hipMalloc(&aout, 2048 * 256 * 4));
      const long src_floats = (long)M * K;        // the product's own A operand as the source buffer
      for (int mf = 1; mf <= 2; ++mf)
        for (int vi = 0; vi < 2; ++vi) {
          Kern k = vi == 0 ? bk[1] : kerns[0];
          CK(hipDeviceSynchronize());
          hipLaunchKernelGGL(k, dim3((unsigned)((P + kBlk - 1) / kBlk)), dim3(kBlk), 0, main_s, J, P, 1111.f, y0, g0);
          CK(hipStreamSynchronize(main_s));
          CK(hipMemcpy(hy0.data(), y0, P * 4, hipMemcpyDeviceToHost));
          CK(hipMemcpy(hg0.data(), g0, P * 36, hipMemcpyDeviceToHost));
          int bad = 0;
          for (int it = 0; it < iters; ++it) {
            for (int i = 0; i < 2; ++i) {
              if (mf == 1) hipLaunchKernelGGL(aggressor_gl<1>, dim3(1024), dim3(256), 0, side[i], aout + i * 1024 * 256, 400, A, src_floats);
              else hipLaunchKernelGGL(aggressor_gl<2>, dim3(1024), dim3(256), 0, side[i], aout + i * 1024 * 256, 400, A, src_floats);
            }
            hipLaunchKernelGGL(k, dim3((unsigned)((P + kBlk - 1) / kBlk)), dim3(kBlk), 0, main_s, J, P, 1111.f, y, g);
            CK(hipMemcpyAsync(hy.data(), y, P * 4, hipMemcpyDeviceToHost, main_s));
            CK(hipMemcpyAsync(hg.data(), g, P * 36, hipMemcpyDeviceToHost, main_s));
            CK(hipStreamSynchronize(main_s));
            bad += memcmp(hy.data(), hy0.data(), P * 4) != 0 || memcmp(hg.data(), hg0.data(), P * 36) != 0;
          }
          printf("library-free aggressor: K loop fed from global memory, %-9s MFMA, victim %-44s %3d of %d launches differ\n",
                 mf == 1 ? "bf16" : "f32", vi == 0 ? bn[1] + 0 : names[0], bad, iters);
          fflush(stdout);
        }
    }
    return 0;
  }
  // ---- the same instruction stream on different DATA: does what the product kernel multiplies matter?
  {
    recmv_set_gemm_mode(1);
    const char* dnames[4] = {"random operands (as above)", "all-zero operands", "all-ones operands", "A random, B zero"};
    std::vector<float> z(M * K, 0.f), o(M * K, 1.f);
    for (int d = 0; d < 4; ++d) {
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(A, d == 0 || d == 3 ? hA.data() : (d == 1 ? z.data() : o.data()), M * K * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(B, d == 0 ? hB.data() : (d == 2 ? o.data() : z.data()), N * K * 4, hipMemcpyHostToDevice));
      const int v = 2;
      hipLaunchKernelGGL(kerns[v], dim3((unsigned)((P + kBlk - 1) / kBlk)), dim3(kBlk), 0, main_s, J, P, 1111.f, y0, g0);
      CK(hipStreamSynchronize(main_s));
      CK(hipMemcpy(hy0.data(), y0, P * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hg0.data(), g0, P * 36, hipMemcpyDeviceToHost));
      int bad = 0;
      for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < 2; ++i)
          for (int r = 0; r < 2; ++r) recmv_gemm_nt(A, K, B, K, nullptr, C[i], N, M, N, K, RECMV_ACT_RELU, 0.f, 1.f, side[i]);
        hipLaunchKernelGGL(kerns[v], dim3((unsigned)((P + kBlk - 1) / kBlk)), dim3(kBlk), 0, main_s, J, P, 1111.f, y, g);
        CK(hipMemcpyAsync(hy.data(), y, P * 4, hipMemcpyDeviceToHost, main_s));
        CK(hipMemcpyAsync(hg.data(), g, P * 36, hipMemcpyDeviceToHost, main_s));
        CK(hipStreamSynchronize(main_s));
        bad += memcmp(hy.data(), hy0.data(), P * 4) != 0 || memcmp(hg.data(), hg0.data(), P * 36) != 0;
      }
      printf("victim V2 beside the bf16x6 64 x 64 product kernel on %-28s %3d of %d launches differ\n", dnames[d], bad, iters);
      fflush(stdout);
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(A, hA.data(), M * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), N * K * 4, hipMemcpyHostToDevice));
  }
  // ---- which instruction family of the side streams' work disturbs the victim?
  {
    constexpr int NA = 18;
    Agg aggs[NA] = {aggressor<0>, aggressor<1>, aggressor<2>, aggressor<3>, aggressor<4>, aggressor<5>,
                    aggressor_lds<16, false, 0, false>, aggressor_lds<16, false, 1, false>, aggressor_lds<16, true, 1, false>,
                    aggressor_lds<16, true, 1, true>, aggressor_lds<4, true, 1, true>, aggressor_lds<16, false, 2, true>,
                    aggressor_lds<16, true, 0, true>, aggressor_lds<16, false, 1, true>,
                    aggressor_loop<0, 1>, aggressor_loop<1, 1>, aggressor_loop<0, 0>, aggressor_loop<1, 0>};
    const char* anames[NA] = {"A0 v_cvt_pk_bf16_f32 only", "A1 bf16 MFMA only", "A2 v_cvt_pk_bf16_f32 + bf16 MFMA", "A3 f32 MFMA only",
                              "A4 v_exp_f32 only", "A5 integer f32->bf16 rounding + bf16 MFMA", "A6 ds_read_b128 only",
                              "A7 ds_read_b128 + bf16 MFMA", "A8 ds_read_b128 + cvt_pk + bf16 MFMA", "A9 A8 + barrier + LDS store per round",
                              "A10 ds_read_b32 + cvt_pk + bf16 MFMA + barrier", "A11 ds_read_b128 + f32 MFMA + barrier",
                              "A12 ds_read_b128 + cvt_pk + barrier, no MFMA", "A13 ds_read_b128 + bf16 MFMA + barrier, no cvt",
                              "A14 the product kernel's K loop (cvt_pk split)", "A15 the K loop, integer-rounded split",
                              "A16 the K loop's split (cvt_pk), no MFMA", "A17 integer-rounded split, no MFMA"};
    float* aout;
    CK(hipMalloc(&aout, 2048 * 256 * 4));
    const int victims[1] = {2};
    for (int vi = 0; vi < 1; ++vi)
      for (int a = 6; a < NA; ++a) {
        const int v = victims[vi];
        hipLaunchKernelGGL(kerns[v], dim3((unsigned)((P + kBlk - 1) / kBlk)), dim3(kBlk), 0, main_s, J, P, 1111.f, y0, g0);
        CK(hipStreamSynchronize(main_s));
        CK(hipMemcpy(hy0.data(), y0, P * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hg0.data(), g0, P * 36, hipMemcpyDeviceToHost));
        int bad = 0;
        long nbad = 0;
        for (int it = 0; it < iters; ++it) {
          // 32 KB static (LDS variants) or 36 KB dynamic LDS per workgroup: at most four aggressor workgroups (four waves per SIMD) on a CU,
          // so the victim's waves run BESIDE them, as they do beside the 64 x 64 product kernel (36 KB)
          for (int i = 0; i < 2; ++i)
            hipLaunchKernelGGL(aggs[a], dim3(1024), dim3(256), a < 6 ? 36864 : (a < 14 ? 4096 : 0), side[i], aout + i * 1024 * 256, 400);
          hipLaunchKernelGGL(kerns[v], dim3((unsigned)((P + kBlk - 1) / kBlk)), dim3(kBlk), 0, main_s, J, P, 1111.f, y, g);
          CK(hipMemcpyAsync(hy.data(), y, P * 4, hipMemcpyDeviceToHost, main_s));
          CK(hipMemcpyAsync(hg.data(), g, P * 36, hipMemcpyDeviceToHost, main_s));
          CK(hipStreamSynchronize(main_s));
          bool diff = false;
          for (long i = 0; i < P; ++i)
            if (memcmp(&hy[i], &hy0[i], 4) != 0 || memcmp(&hg[i * 9], &hg0[i * 9], 36) != 0) {
              diff = true;
              ++nbad;
            }
          bad += diff;
        }
        CK(hipDeviceSynchronize());
        printf("victim %-32.32s beside %-44s %3d of %d launches differ (%ld matrices)\n", names[v], anames[a], bad, iters, nbad);
        fflush(stdout);
      }
  }
  return 0;
}
