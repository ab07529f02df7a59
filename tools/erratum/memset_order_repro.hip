// Is a hipMemsetAsync between a stream's kernels ordered with them while OTHER streams have fills and kernels in flight?
//
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/memset_order_repro tools/memset_order_repro.hip && gpurun_out/memset_order_repro [iters]
//
// The jet pass of csrc/mlp_jet.hip prepared its stacked input like this in rounds 1-4 (stream s):
//     posenc_forward   value rows   [0, P)   x [0, d_pe)
//     gather_rows      value rows   [0, P)   x [d_pe, ld)
//     hipMemsetAsync   tangent rows [P, 4P)  x [0, ld)      = 0
//     posenc_jvp x 3   tangent rows [P, 4P)  x [0, d_pe)
//     product kernels read all 4P rows
// and in the bf16x6 matrix mode, with the second garment's chain on a side stream, about one run of the loop in four parted in the
// last bits in exactly those rows.  This program replays the sequence on stream A, on a buffer dirtied by the previous
// pass, with three other streams issuing fills and small kernels, and COUNTS — per pass — tangent elements that are not what a
// stream-ordered execution leaves: `late_fill` = a column of [0, d_pe) that holds 0 instead of the jvp value (the fill ran after the
// jvp), `early_read` = a column of [d_pe, ld) that still holds the previous pass's value (the reader ran before the fill).
// Zero in both columns for every pass = the runtime's fill is ordered here and the loop's divergence has another cause.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e__ = (x);                                                              \
    if (e__ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

__global__ void write_cols(float* p, long ld, long rows, int c0, int width, float v) {
  const long total = rows * width;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / width;
    p[r * ld + c0 + (int)(e - r * width)] = v;
  }
}

__global__ void check_rows(const float* p, long ld, long rows, int d_pe, float v, unsigned long long* err) {
  const long total = rows * ld;
  unsigned long long late = 0, early = 0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / ld;
    const int c = (int)(e - r * ld);
    const float x = p[e];
    if (c < d_pe) late += (x != v);
    else early += (x != 0.f);
  }
  if (late) atomicAdd(&err[0], late);
  if (early) atomicAdd(&err[1], early);
}

__global__ void busy(float* p, long n, int rounds) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    float a = p[e];
    for (int i = 0; i < rounds; ++i) a = a * 1.0001f + 0.5f;
    p[e] = a;
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 1500;
  const long P = 30720, ld = 168;
  const int d_pe = 39;
  hipStream_t sA, sB[3];
  CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
  for (auto& s : sB) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // one large block carved like a caching allocator's: the fill covers an INTERIOR range of it
  char* block;
  const size_t bytes = (size_t)4 * P * ld * 4;
  CK(hipMalloc(&block, 3 * bytes));
  float* in = (float*)(block + bytes);
  unsigned long long* err;
  CK(hipMalloc(&err, 16));
  float* other[3];
  const long on = 8l << 20;
  for (auto& o : other) CK(hipMalloc(&o, on * 4));
  std::vector<unsigned long long> per_pass(2 * iters, 0);
  unsigned long long* host;
  CK(hipHostMalloc(&host, 16 * (size_t)iters));
  unsigned long long bad_passes = 0, tot[2] = {0, 0};
  for (int mode = 0; mode < 2; ++mode) {      // 0: the other streams idle, 1: fills + kernels in flight on three other streams
    bad_passes = tot[0] = tot[1] = 0;
    for (int i = 0; i < iters; ++i) {
      const float v = 0.1f * (float)(i % 97 + 1);
      if (mode == 1)
        for (int k = 0; k < 3; ++k) {
          CK(hipMemsetAsync(other[k], 0, (size_t)(on / (k + 1)) * 4, sB[k]));
          hipLaunchKernelGGL(busy, dim3(64 * (k + 1)), dim3(256), 0, sB[k], other[k], on / 8, 40 * (k + 1));
          CK(hipMemsetAsync(other[k] + on / 2, 0, 4096 * (k + 1), sB[k]));
          hipLaunchKernelGGL(busy, dim3(16), dim3(256), 0, sB[k], other[k], on / 64, 10);
        }
      CK(hipMemsetAsync(err, 0, 16, sA));
      hipLaunchKernelGGL(write_cols, dim3(2048), dim3(256), 0, sA, in, ld, 4 * P, 0, (int)ld, v);            // dirty: the previous pass
      hipLaunchKernelGGL(write_cols, dim3(1024), dim3(256), 0, sA, in, ld, P, 0, (int)ld, v + 1.f);           // value rows
      CK(hipMemsetAsync(in + P * ld, 0, (size_t)3 * P * ld * 4, sA));                                        // the fill under test
      for (int k = 0; k < 3; ++k)
        hipLaunchKernelGGL(write_cols, dim3(512), dim3(256), 0, sA, in + (long)(k + 1) * P * ld, ld, P, 0, d_pe, v + 2.f);   // jvp
      hipLaunchKernelGGL(check_rows, dim3(2048), dim3(256), 0, sA, in + P * ld, ld, 3 * P, d_pe, v + 2.f, err);
      CK(hipMemcpyAsync(host + 2 * i, err, 16, hipMemcpyDeviceToHost, sA));
    }
    CK(hipDeviceSynchronize());
    for (int i = 0; i < iters; ++i) {
      tot[0] += host[2 * i];
      tot[1] += host[2 * i + 1];
      bad_passes += (host[2 * i] | host[2 * i + 1]) != 0;
    }
    printf("memset_order_repro: %s: %d passes of [dirty, value rows, hipMemsetAsync %.1f MB, 3 x jvp, check]: %llu passes with a "
           "misordered element (late_fill %llu, early_read %llu elements)\n",
           mode ? "three other streams busy with fills + kernels" : "other streams idle", iters, 3.0 * P * ld * 4 / 1e6, bad_passes,
           tot[0], tot[1]);
  }
  return 0;
}
