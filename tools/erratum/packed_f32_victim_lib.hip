// The packed-f32 victim of tools/packed_f32_bf16_mfma_repro.hip as a tiny shared library, so that a Python script can run it beside
// OTHER aggressors (torch's own bf16 matmul: tools/packed_f32_vs_torch_bf16.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/bin/libpacked_victim.so tools/packed_f32_victim_lib.hip
#include <hip/hip_runtime.h>

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

extern "C" int launch_victim(const float* J, long P, float* out, int paired, void* stream) {
  const unsigned grid = (unsigned)((P + 255) / 256);
  if (paired) hipLaunchKernelGGL(victim<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, J, P, out);
  else hipLaunchKernelGGL(victim<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, J, P, out);
  return (int)hipGetLastError();
}
