// Follow-up of tools/valu_disturb_repro.hip: WHAT goes wrong in lanes 48..63 of a wave that runs beside the bf16x6 mode's NT product
// kernel — the transcendental unit's results, or the lane masks that compares produce (VCC / SGPR pairs, EXEC)?
// Self-checking victims: every lane compares what the suspect instruction gave with the same quantity formed by integer
// arithmetic on the operand's bits (no compare, no select, no branch in the check itself) and ORs / adds the mismatches.
//
//   W1  v_cmp + v_cndmask   sel = x < 0.5f ? 1 : 0            against the sign bit of (x - 0.5f)
//   W2  v_cmp + branch      if (x < 0.5f) ++count             against the sum of the sign bits
//   W3  v_rcp_f32 twice on the same operand                   bitwise equal?
//   W4  v_cmp -> SGPR pair, read back with ballot             against the sign bit of (x - 0.5f)
//   W5  v_sqrt_f32 / v_log_f32 twice                          bitwise equal?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mask_disturb_repro tools/mask_disturb_repro.hip -Iinclude -Lrec-mv_amd/lib -lrecmv_hip -Wl,-rpath,'$ORIGIN/../../rec-mv_amd/lib'
//   tools/bin/mask_disturb_repro [iters=200]
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

__host__ __device__ inline float next_x(unsigned& s) {       // in [0, 1), never exactly 0.5
  s = s * 1664525u + 1013904223u;
  const unsigned u = 0x3f800000u | (s >> 9) | 1u;
  float f;
  memcpy(&f, &u, 4);
  return f - 1.0f;
}

template <int W>
__global__ __launch_bounds__(256) void victim(unsigned* __restrict__ out, int rounds) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  unsigned s = 0x9e3779b9u * (unsigned)(tid + 1), bad = 0, count = 0, expect = 0;
  for (int i = 0; i < rounds; ++i) {
    float x = next_x(s);
    asm volatile("" : "+v"(x));
    const unsigned sign = __float_as_uint(x - 0.5f) >> 31;       // 1 when x < 0.5
    if (W == 1) {
      unsigned sel = x < 0.5f ? 1u : 0u;
      asm volatile("" : "+v"(sel));
      bad += sel ^ sign;
    } else if (W == 2) {
      if (x < 0.5f) {
        asm volatile("v_add_u32 %0, 1, %0" : "+v"(count));
      }
      expect += sign;
    } else if (W == 3) {
      float a = x + 1.0f, b = a;
      asm volatile("" : "+v"(a));
      asm volatile("" : "+v"(b));
      const float r1 = __builtin_amdgcn_rcpf(a), r2 = __builtin_amdgcn_rcpf(b);
      bad |= __float_as_uint(r1) ^ __float_as_uint(r2);
    } else if (W == 4) {
      const unsigned long long m = __ballot(x < 0.5f);
      bad += ((unsigned)(m >> (threadIdx.x & 63)) & 1u) ^ sign;
    } else if (W == 5) {
      float a = x + 1.0f, b = a;
      asm volatile("" : "+v"(a));
      asm volatile("" : "+v"(b));
      const float r1 = __builtin_amdgcn_sqrtf(a) + __builtin_amdgcn_logf(a), r2 = __builtin_amdgcn_sqrtf(b) + __builtin_amdgcn_logf(b);
      bad |= __float_as_uint(r1) ^ __float_as_uint(r2);
    } else if (W == 6 || W == 7) {
      // the form the compiler emits in the regulariser: a VOP3 compare into an SGPR PAIR (not VCC) and a VOP3 select reading that pair,
      // W6 back to back, W7 with `s_nop 4` between them
      unsigned long long m;
      unsigned sel;
      const float half = 0.5f;
      if (W == 6)
        asm volatile("v_cmp_lt_f32_e64 %0, %2, %3\n\tv_cndmask_b32_e64 %1, 0, 1, %0" : "=&s"(m), "=v"(sel) : "v"(x), "v"(half));
      else
        asm volatile("v_cmp_lt_f32_e64 %0, %2, %3\n\ts_nop 4\n\tv_cndmask_b32_e64 %1, 0, 1, %0" : "=&s"(m), "=v"(sel) : "v"(x), "v"(half));
      bad += sel ^ sign;
    } else if (W == 8) {
      // packed f32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32), each checked against the scalar instruction's bits
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 a = {x, x + 0.25f}, b = {x * 0.5f + 0.1f, 1.5f - x}, c = {0.75f, x};
      asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
      f2 pm, pf, pa;
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pm) : "v"(a), "v"(b));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pf) : "v"(a), "v"(b), "v"(c));
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pa) : "v"(a), "v"(b));
      float sm0, sm1, sf0, sf1, sa0, sa1;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(sm0) : "v"(a.x), "v"(b.x));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(sm1) : "v"(a.y), "v"(b.y));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(sf0) : "v"(a.x), "v"(b.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(sf1) : "v"(a.y), "v"(b.y), "v"(c.y));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(sa0) : "v"(a.x), "v"(b.x));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(sa1) : "v"(a.y), "v"(b.y));
      bad |= (__float_as_uint(pm.x) ^ __float_as_uint(sm0)) | (__float_as_uint(pm.y) ^ __float_as_uint(sm1)) |
             (__float_as_uint(pf.x) ^ __float_as_uint(sf0)) | (__float_as_uint(pf.y) ^ __float_as_uint(sf1)) |
             (__float_as_uint(pa.x) ^ __float_as_uint(sa0)) | (__float_as_uint(pa.y) ^ __float_as_uint(sa1));
    }
  }
  out[tid] = W == 2 ? (count ^ expect) : bad;
}
typedef void (*Vic)(unsigned*, int);

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int blocks = 120, threads = blocks * 256;
  unsigned* out;
  CK(hipMalloc(&out, threads * 4));
  std::vector<unsigned> h(threads);
  const long M = 12000, N = 512, K = 4096;
  float *A, *B, *C[2];
  CK(hipMalloc(&A, M * K * 4));
  CK(hipMalloc(&B, N * K * 4));
  std::vector<float> hA(M * K), hB(N * K);
  unsigned s = 1u;
  for (auto& v : hA) v = next_x(s) - 0.5f;
  for (auto& v : hB) v = (next_x(s) - 0.5f) * 0.05f;
  CK(hipMemcpy(A, hA.data(), M * K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hB.data(), N * K * 4, hipMemcpyHostToDevice));
  hipStream_t main_s, side[2];
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    CK(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
    CK(hipMalloc(&C[i], M * N * 4));
  }
  constexpr int NW = 8;
  Vic vics[NW] = {victim<1>, victim<2>, victim<3>, victim<4>, victim<5>, victim<6>, victim<7>, victim<8>};
  const char* names[NW] = {"W1 v_cmp + v_cndmask vs sign bit", "W2 v_cmp + divergent branch vs sign bits", "W3 v_rcp_f32 twice, same operand",
                           "W4 v_cmp -> SGPR pair (ballot) vs sign bit", "W5 v_sqrt_f32 + v_log_f32 twice",
                           "W6 v_cmp_e64 -> SGPR pair -> v_cndmask_e64", "W7 = W6 with s_nop 4 between", "W8 v_pk_{mul,fma,add}_f32 vs scalar ops"};
  for (int mode = 1; mode >= 0; --mode) {
    recmv_set_gemm_mode(mode);
    for (int busy = 1; busy >= 1; --busy)
      for (int w = 0; w < NW; ++w) {
        CK(hipDeviceSynchronize());
        int bad_launches = 0;
        long bad_lanes = 0, hist[4] = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
          if (busy)
            for (int i = 0; i < 2; ++i)
              if (recmv_gemm_nt(A, K, B, K, nullptr, C[i], N, M, N, K, RECMV_ACT_RELU, 0.f, 1.f, side[i]) != 0) {
                fprintf(stderr, "gemm_nt: %s\n", recmv_last_error());
                return 2;
              }
          hipLaunchKernelGGL(vics[w], dim3(blocks), dim3(256), 0, main_s, out, 2000);
          CK(hipMemcpyAsync(h.data(), out, threads * 4, hipMemcpyDeviceToHost, main_s));
          CK(hipStreamSynchronize(main_s));
          bool any = false;
          for (int t = 0; t < threads; ++t)
            if (h[t]) {
              any = true;
              ++bad_lanes;
              ++hist[(t & 63) >> 4];
            }
          bad_launches += any;
        }
        printf("%-6s products %-4s  %-44s %3d of %d launches with a mismatch; lanes by quarter of the wave [0-15 | 16-31 | 32-47 | 48-63]: "
               "%ld %ld %ld %ld\n", mode ? "bf16x6" : "f32", busy ? "busy" : "idle", names[w], bad_launches, iters, hist[0], hist[1], hist[2],
               hist[3]);
        fflush(stdout);
      }
  }
  return 0;
}
