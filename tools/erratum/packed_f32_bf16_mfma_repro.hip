// Stand-alone reproducer (no library, one file): on gfx950 / ROCm 7.2, a kernel made of PACKED-F32 VALU instructions
// (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) returns wrong values in lanes 48..63 of some waves when it runs beside a kernel whose K
// loop feeds v_mfma_f32_32x32x16_bf16 from global memory through LDS (the shape of a bf16 GEMM).  The same victim with its FMAs and adds
// kept scalar (the compiler still pairs the multiplies: 178 v_pk_mul_f32, so the multiply is not the instruction that fails) never
// fails; the same aggressor with v_mfma_f32_32x32x2_f32 never disturbs.  MI355X, 200 launches: 168 bad, every wrong thread in lanes
// 48..63 (profiles/r05_race_13_standalone_repro.txt).
//
//   hipcc --offload-arch=gfx950 -O3 -o packed_f32_bf16_mfma_repro tools/packed_f32_bf16_mfma_repro.hip && ./packed_f32_bf16_mfma_repro [launches=200]
//
// Found while chasing a run-to-run divergence of rec-mv_amd's optional bf16x6 matrix mode (tools/valu_disturb_repro.hip has
// the bisect that led here).  Victim: per thread, fifteen plane rotations of the columns of two 3x3 matrices with constant (c, s) —
// nothing but multiplies and FMAs on eighteen live registers; `PAIRED` lets the compiler form v_pk_* (check the ISA: --save-temps),
// the other instantiation puts a register barrier on every value so that the same arithmetic stays scalar.  A launch is BAD when
// its output differs from the first launch's (same input, no atomics, no data-dependent control flow).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e__ = (x);                                                                  \
    if (e__ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ victim
template <bool PAIRED>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ J, long P, float* __restrict__ out) {
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
  if (i0 >= P) return;
  float b[3][3], v[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      b[i][j] = J[i0 * 9 + 3 * i + j];
      v[i][j] = i == j ? 1.f : 0.f;
    }
  const float c = 0.8f, s = 0.6f;
#pragma unroll
  for (int sweep = 0; sweep < 5; ++sweep)
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float bp = b[k][p], bq = b[k][q], vp = v[k][p], vq = v[k][q];
        if (!PAIRED) {
          asm volatile("" : "+v"(bp));
          asm volatile("" : "+v"(bq));
          asm volatile("" : "+v"(vp));
          asm volatile("" : "+v"(vq));
        }
        float n0 = c * bp - s * bq, n1 = s * bp + c * bq, n2 = c * vp - s * vq, n3 = s * vp + c * vq;
        if (!PAIRED) {
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
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) out[i0 * 9 + 3 * i + j] = b[i][j] + 2.f * v[i][j];
}

// ------------------------------------------------------------------------------------------------ aggressor
// A 64 x 64 GEMM tile's K loop: two float4 per thread and operand from global memory into registers while the previous K-tile is
// multiplied, then into LDS; every lane converts 8 consecutive k of its operand rows to bf16 (three-way split, as a bf16x3 GEMM
// does) and issues v_mfma_f32_32x32x16_bf16 — or, BF16 = false, v_mfma_f32_32x32x2_f32 on the f32 values.
struct Pcs {
  bf16x8_t h, m, l;
};
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  auto cvt = [](float a, float b) -> unsigned { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t)); };
  h = cvt(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt(r0, r1);
  l = cvt(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ Pcs split8(float4 a, float4 b) {
  unsigned h[4], m[4], l[4];
  split2(a.x, a.y, h[0], m[0], l[0]);
  split2(a.z, a.w, h[1], m[1], l[1]);
  split2(b.x, b.y, h[2], m[2], l[2]);
  split2(b.z, b.w, h[3], m[3], l[3]);
  Pcs p;
  p.h = __builtin_bit_cast(bf16x8_t, (u32x4_t){h[0], h[1], h[2], h[3]});
  p.m = __builtin_bit_cast(bf16x8_t, (u32x4_t){m[0], m[1], m[2], m[3]});
  p.l = __builtin_bit_cast(bf16x8_t, (u32x4_t){l[0], l[1], l[2], l[3]});
  return p;
}

template <bool BF16>
__global__ __launch_bounds__(256) void aggressor(float* __restrict__ out, int rounds, const float* __restrict__ src, long src_floats) {
  constexpr int LDK = 36;
  __shared__ __attribute__((aligned(16))) float As[2 * 64 * LDK], Bs[2 * 64 * LDK];       // 36 KB: four workgroups per CU at most
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31), khalf = (lane >> 5) * 4;
  const int row = tid >> 3, c4 = tid & 7;
  const long span = src_floats - 64 * 4096 - 8192;
  const long base = ((long)blockIdx.x * 64 * 4096) % span;
  const float* pa = src + base + (long)row * 4096 + c4 * 4;
  const float* pb = src + ((base + 32 * 4096) % span) + (long)row * 4096 + c4 * 4;
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
      if (BF16) {
        const Pcs a = split8(*reinterpret_cast<const float4*>(as + ks * 16), *reinterpret_cast<const float4*>(as + ks * 16 + 4));
        const Pcs b = split8(*reinterpret_cast<const float4*>(bs + ks * 16), *reinterpret_cast<const float4*>(bs + ks * 16 + 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
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

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200;
  const long P = 30714;
  std::vector<float> hJ(P * 9);
  unsigned s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) * (1.f / 16777216.f)) - 0.5f;
  };
  for (long i = 0; i < P; ++i)
    for (int e = 0; e < 9; ++e) hJ[i * 9 + e] = (e % 4 == 0 ? 1.f : 0.f) + 0.06f * rnd();
  float *J, *o, *o0, *src, *aout;
  const long src_floats = 12000l * 4096;
  CK(hipMalloc(&J, P * 36));
  CK(hipMalloc(&o, P * 36));
  CK(hipMalloc(&o0, P * 36));
  CK(hipMalloc(&src, src_floats * 4));
  CK(hipMalloc(&aout, 2 * 1024 * 256 * 4));
  CK(hipMemcpy(J, hJ.data(), P * 36, hipMemcpyHostToDevice));
  std::vector<float> hs(src_floats);
  for (auto& v : hs) v = rnd();
  CK(hipMemcpy(src, hs.data(), src_floats * 4, hipMemcpyHostToDevice));
  hipStream_t main_s, side[2];
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  for (auto& st : side) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  std::vector<float> h(P * 9), h0(P * 9);
  const unsigned grid = (unsigned)((P + 255) / 256);
  for (int bf16 = 1; bf16 >= 0; --bf16)
    for (int paired = 1; paired >= 0; --paired) {
      CK(hipDeviceSynchronize());
      if (paired) hipLaunchKernelGGL(victim<true>, dim3(grid), dim3(256), 0, main_s, J, P, o0);
      else hipLaunchKernelGGL(victim<false>, dim3(grid), dim3(256), 0, main_s, J, P, o0);
      CK(hipStreamSynchronize(main_s));
      CK(hipMemcpy(h0.data(), o0, P * 36, hipMemcpyDeviceToHost));
      int bad = 0;
      long elems = 0, quarter[4] = {0, 0, 0, 0};
      for (int it = 0; it < launches; ++it) {
        for (int i = 0; i < 2; ++i) {
          if (bf16) hipLaunchKernelGGL(aggressor<true>, dim3(1024), dim3(256), 0, side[i], aout + i * 1024 * 256, 400, src, src_floats);
          else hipLaunchKernelGGL(aggressor<false>, dim3(1024), dim3(256), 0, side[i], aout + i * 1024 * 256, 400, src, src_floats);
        }
        if (paired) hipLaunchKernelGGL(victim<true>, dim3(grid), dim3(256), 0, main_s, J, P, o);
        else hipLaunchKernelGGL(victim<false>, dim3(grid), dim3(256), 0, main_s, J, P, o);
        CK(hipMemcpyAsync(h.data(), o, P * 36, hipMemcpyDeviceToHost, main_s));
        CK(hipStreamSynchronize(main_s));
        bool diff = false;
        for (long i = 0; i < P; ++i)
          if (memcmp(&h[i * 9], &h0[i * 9], 36) != 0) {
            diff = true;
            ++elems;
            ++quarter[(i & 63) >> 4];
          }
        bad += diff;
      }
      printf("aggressor %-33s victim %-34s %3d of %d launches differ from the first; threads by quarter of the wave "
             "[0-15 | 16-31 | 32-47 | 48-63]: %ld %ld %ld %ld\n",
             bf16 ? "v_mfma_f32_32x32x16_bf16 K loop," : "v_mfma_f32_32x32x2_f32 K loop,", paired ? "v_pk_fma/add/mul_f32," : "scalar FMAs/adds (+ v_pk_mul_f32),",
             bad, launches, quarter[0], quarter[1], quarter[2], quarter[3]);
      fflush(stdout);
    }
  return 0;
}
