// Backward of one fused layer y = act(x W^T + b) in ONE call — gfx950.
//
// The reference leaves this to autograd of nn.Linear + activation (cuBLAS gemms + elementwise + a column reduction
// for the bias, 5 launches driven from Python per layer).  recmv_linear_backward enqueues, on the caller's stream:
//   gz = gy (.) act'(z)                    act_grad_2d kernel (through y = act(z))
//   gb = column sums of gz                 deterministic two-stage reduction (no float atomics)
//   gx = gz W      = gz (W^T)^T            MFMA gemm_nt against the cached W^T
//   gW = gz^T x                            MFMA gemm_tn (split-K over the points, fixed reduction order)
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;
constexpr int kColTile = 64;       // columns per block (one wave-width: coalesced 256-byte rows)
constexpr int kMaxChunks = 96;      // partial rows per column; 8 column tiles x 96 chunks = 768 workgroups at N=512

// partial[chunk][c] = sum over the chunk's rows of g[r*ld + c]
__global__ __launch_bounds__(kBlk) void colsum_partial_kernel(const float* __restrict__ g, int64_t ld, int64_t rows,
                                                              int cols, int64_t rows_per_chunk,
                                                              float* __restrict__ partial) {
  __shared__ float sh[kBlk / kColTile][kColTile];
  const int ctile = blockIdx.x, chunk = blockIdx.y;
  const int lane = threadIdx.x % kColTile, rgrp = threadIdx.x / kColTile;
  const int c = ctile * kColTile + lane;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  float s = 0.f;
  if (c < cols)
    for (int64_t r = r0 + rgrp; r < r1; r += kBlk / kColTile) s += g[r * ld + c];
  sh[rgrp][lane] = s;
  __syncthreads();
  if (rgrp == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < kBlk / kColTile; ++q) t += sh[q][lane];
    partial[(int64_t)chunk * cols + c] = t;
  }
}

__device__ __forceinline__ float dact_y(float y, int act, float p) {
  switch (act) {
    case RECMV_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case RECMV_ACT_SOFTPLUS: return -expm1f(-p * y);
    case RECMV_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// gz[r,c] = gy[r,c] * act'(z) (through y = act(z)), written out, AND partial[chunk][c] = column sums of gz over the
// chunk's rows: the bias gradient comes out of the same pass over dY that produces dZ.
// VEC = 4: 16-byte accesses (16 threads span the 64-column tile, 16 rows in flight per block pass); VEC = 1: scalar.
template <int VEC>
__global__ __launch_bounds__(kBlk) void act_grad_colsum_kernel(const float* __restrict__ gy, int64_t ldg,
                                                               const float* __restrict__ y, int64_t ldy,
                                                               float* __restrict__ gz, int64_t ldz, int64_t rows,
                                                               int cols, int64_t rows_per_chunk, int act, float p,
                                                               float* __restrict__ partial) {
  constexpr int CT = kColTile / VEC;          // threads across the column tile
  constexpr int RG = kBlk / CT;               // rows in flight
  __shared__ float sh[RG][kColTile];
  const int ctile = blockIdx.x, chunk = blockIdx.y;
  const int cl = threadIdx.x % CT, rgrp = threadIdx.x / CT;
  const int c = ctile * kColTile + cl * VEC;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  float s[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s[v] = 0.f;
  if (c < cols) {
    for (int64_t r = r0 + rgrp; r < r1; r += RG) {
      if (VEC == 4) {
        const float4 g4 = *reinterpret_cast<const float4*>(gy + r * ldg + c);
        const float4 y4 = *reinterpret_cast<const float4*>(y + r * ldy + c);
        float4 o;
        o.x = g4.x * dact_y(y4.x, act, p);
        o.y = g4.y * dact_y(y4.y, act, p);
        o.z = g4.z * dact_y(y4.z, act, p);
        o.w = g4.w * dact_y(y4.w, act, p);
        *reinterpret_cast<float4*>(gz + r * ldz + c) = o;
        s[0] += o.x;
        s[VEC > 1 ? 1 : 0] += o.y;
        s[VEC > 2 ? 2 : 0] += o.z;
        s[VEC > 3 ? 3 : 0] += o.w;
      } else {
        const float v = gy[r * ldg + c] * dact_y(y[r * ldy + c], act, p);
        gz[r * ldz + c] = v;
        s[0] += v;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) sh[rgrp][cl * VEC + v] = s[v];
  __syncthreads();
  if (threadIdx.x < kColTile) {
    const int cc = ctile * kColTile + threadIdx.x;
    if (cc < cols) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < RG; ++q) t += sh[q][threadIdx.x];
      partial[(int64_t)chunk * cols + cc] = t;
    }
  }
}

__global__ __launch_bounds__(kBlk) void colsum_final_kernel(const float* __restrict__ partial, int chunks, int cols,
                                                            float* __restrict__ out) {
  const int c = blockIdx.x * kBlk + threadIdx.x;
  if (c >= cols) return;
  // fixed summation order; 8 independent loads in flight per step instead of one dependent L2 round trip each
  float s = 0.f;
  int q = 0;
  for (; q + 8 <= chunks; q += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(q + u) * cols + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; q < chunks; ++q) s += partial[(int64_t)q * cols + c];
  out[c] = s;
}

inline int colsum_chunks(int64_t rows) {
  int64_t ch = ceil_div(rows, 256);
  if (ch > kMaxChunks) ch = kMaxChunks;
  if (ch < 1) ch = 1;
  return (int)ch;
}

inline int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int64_t recmv_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return (int64_t)colsum_chunks(rows) * cols * 4;
}

extern "C" int recmv_colsum(const float* g, int64_t ld, int64_t rows, int64_t cols, float* out, void* workspace,
                            int64_t workspace_bytes, void* stream) {
  RECMV_REQUIRE(rows >= 0 && cols >= 0 && cols < (1 << 24), "colsum: bad size");
  if (cols == 0) return RECMV_OK;
  RECMV_REQUIRE(out, "colsum: NULL output");
  hipStream_t s = (hipStream_t)stream;
  if (rows == 0) {
    RECMV_HIP_TRY(hipMemsetAsync(out, 0, cols * 4, s));
    return RECMV_OK;
  }
  RECMV_REQUIRE(g && ld >= cols, "colsum: bad input");
  const int chunks = colsum_chunks(rows);
  if (!workspace || workspace_bytes < (int64_t)chunks * cols * 4) {
    set_error("colsum: workspace too small");
    return RECMV_ERR_WORKSPACE;
  }
  const int64_t rpc = ceil_div(rows, chunks);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)ceil_div(cols, kColTile), (unsigned)chunks), dim3(kBlk), 0, s,
                     g, ld, rows, (int)cols, rpc, (float*)workspace);
  int rc = check_launch("colsum/partial");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)ceil_div(cols, kBlk)), dim3(kBlk), 0, s,
                     (const float*)workspace, chunks, (int)cols, out);
  return check_launch("colsum/final");
}

extern "C" int64_t recmv_linear_backward_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0) return 256;
  return align256(M * ((N + 3) & ~3ll) * 4) + align256(recmv_gemm_tn_workspace_bytes(N, K, M)) +
         align256(recmv_colsum_workspace_bytes(M, N)) + 256;
}

// y, gy: [M,N] (outputs of the layer and their cotangent); x: [M,K]; Wt: [K,N] = W^T.
// gx [M,K], gW [N,K], gb [N]: any may be NULL.
extern "C" int recmv_linear_backward(const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x,
                                     int64_t ldx, const float* Wt, int64_t ldwt, int64_t M, int64_t N, int64_t K,
                                     int act, float act_param, float* gx, int64_t ldgx, float* gW, float* gb,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  RECMV_REQUIRE(M >= 0 && N >= 0 && K >= 0, "linear_backward: negative size");
  RECMV_REQUIRE(gy && (act == RECMV_ACT_NONE || y), "linear_backward: NULL pointer");
  const int64_t need = recmv_linear_backward_workspace_bytes(M, N, K);
  if (!workspace || workspace_bytes < need) {
    set_error("linear_backward: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    return RECMV_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  float* gzbuf = (float*)ws;
  const int64_t ldz = (N + 3) & ~3ll;     // dZ rows padded to whole float4s: the dW product then takes its aligned path for N = 473
  char* tn_ws = ws + align256(M * ldz * 4);
  const int64_t tn_bytes = align256(recmv_gemm_tn_workspace_bytes(N, K, M));
  char* cs_ws = tn_ws + tn_bytes;
  const int64_t cs_bytes = align256(recmv_colsum_workspace_bytes(M, N));
  const float* gz = gy;
  int64_t ldgz = ldgy;
  int rc;
  bool gb_done = false;
  if (act != RECMV_ACT_NONE && M > 0 && N > 0) {
    if (gb) {
      // dZ and its column sums (the bias gradient) in one pass over dY
      const int chunks = colsum_chunks(M);
      const int64_t rpc = ceil_div(M, chunks);
      const bool vec = N % 4 == 0 && ldgy % 4 == 0 && ldy % 4 == 0 &&
                       ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
      if (vec)
        hipLaunchKernelGGL(act_grad_colsum_kernel<4>, dim3((unsigned)ceil_div(N, kColTile), (unsigned)chunks), dim3(kBlk),
                           0, (hipStream_t)stream, gy, ldgy, y, ldy, gzbuf, ldz, M, (int)N, rpc, act, act_param,
                           (float*)cs_ws);
      else
        hipLaunchKernelGGL(act_grad_colsum_kernel<1>, dim3((unsigned)ceil_div(N, kColTile), (unsigned)chunks), dim3(kBlk),
                           0, (hipStream_t)stream, gy, ldgy, y, ldy, gzbuf, ldz, M, (int)N, rpc, act, act_param,
                           (float*)cs_ws);
      rc = check_launch("linear_backward/act_grad_colsum");
      if (rc) return rc;
      hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)ceil_div(N, kBlk)), dim3(kBlk), 0, (hipStream_t)stream,
                         (const float*)cs_ws, chunks, (int)N, gb);
      rc = check_launch("linear_backward/colsum_final");
      if (rc) return rc;
      gb_done = true;
    } else {
      rc = recmv_act_grad_2d(gy, ldgy, y, ldy, gzbuf, ldz, M, N, act, act_param, 1.f, 1.f, stream);
      if (rc) return rc;
    }
    gz = gzbuf;
    ldgz = ldz;
  }
  if (gb && !gb_done) {
    rc = recmv_colsum(gz, ldgz, M, N, gb, cs_ws, cs_bytes, stream);
    if (rc) return rc;
  }
  if (gx && K > 0) {
    RECMV_REQUIRE(Wt, "linear_backward: gx needs W^T");
    rc = recmv_gemm_nt(gz, ldgz, Wt, ldwt, nullptr, gx, ldgx, M, K, N, RECMV_ACT_NONE, 0.f, 1.f, stream);
    if (rc) return rc;
  }
  if (gW && K > 0) {
    RECMV_REQUIRE(x, "linear_backward: gW needs x");
    rc = recmv_gemm_tn(gz, ldgz, x, ldx, gW, K, N, K, M, tn_ws, tn_bytes, stream);
    if (rc) return rc;
  }
  return RECMV_OK;
}
