// Batched 3x3 inverse, forward and backward — gfx950.
//
// Semantics follow FastMinv/Matrix3x3InvKernels.cu:22-104 of the reference (cofactor expansion,
// absolute-determinant singularity test, division by det, backward = -(A^-T G A^-T) from the saved
// inverse).  The design does not: the reference reads/writes 36-byte AoS records straight from each
// thread (uncoalesced, 1024-thread blocks on the default stream); here a 256-thread workgroup moves
// its 256 matrices (9216 B) through LDS with 16-byte coalesced global accesses, the stride-9 LDS reads
// are bank-conflict free (gcd(9,32)=1), and the launch goes on the caller's stream.
//
// Algorithmic traffic: 73 B / matrix forward (36 in + 36 out + 1), 108 B / matrix backward.
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;

// Arithmetic is kept un-contracted so the f32/f64 results are bit-identical to the CPU oracle
// (oracle/recmv_oracle.c is compiled with -ffp-contract=off).  The kernel is memory bound.
#pragma clang fp contract(off)

template <typename T>
__device__ __forceinline__ bool inv_one(const T* m, T* inv) {
  T cof00 = m[4] * m[8] - m[5] * m[7];
  T cof01 = -m[3] * m[8] + m[5] * m[6];
  T cof02 = m[3] * m[7] - m[4] * m[6];
  T cof10 = -m[1] * m[8] + m[2] * m[7];
  T cof11 = m[0] * m[8] - m[2] * m[6];
  T cof12 = -m[0] * m[7] + m[1] * m[6];
  T cof20 = m[1] * m[5] - m[2] * m[4];
  T cof21 = -m[0] * m[5] + m[2] * m[3];
  T cof22 = m[0] * m[4] - m[1] * m[3];
  T det = m[0] * cof00 + m[1] * cof01 + m[2] * cof02;
  // reference: fabs(det) < 0.0001 with a double literal -> the comparison is done in double
  if (fabs((double)det) < 0.0001) {
#pragma unroll
    for (int i = 0; i < 9; ++i) inv[i] = (T)0;
    return false;
  }
  inv[0] = cof00 / det;
  inv[1] = cof10 / det;
  inv[2] = cof20 / det;
  inv[3] = cof01 / det;
  inv[4] = cof11 / det;
  inv[5] = cof21 / det;
  inv[6] = cof02 / det;
  inv[7] = cof12 / det;
  inv[8] = cof22 / det;
  return true;
}

template <typename T>
__device__ __forceinline__ void inv_bwd_one(const T* g, const T* c, T* out) {
  // out[r][s] = -sum_{i,j} g[i][j] * c[i][r] * c[s][j]     (c = saved inverse)
  // summation order identical to the reference's unrolled expression (i outer, j inner).
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      T acc = (T)0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          T term = g[3 * i + j] * c[3 * i + r] * c[3 * s + j];
          acc = (i == 0 && j == 0) ? term : acc + term;
        }
      }
      out[3 * r + s] = -acc;
    }
  }
}

// Cooperative, coalesced copy of `count` scalars global<->LDS for one workgroup.
template <typename T>
__device__ __forceinline__ void load_tile(const T* __restrict__ g, T* lds, int count, bool vec_ok) {
  if (vec_ok) {
    constexpr int V = 16 / sizeof(T);
    const int nvec = count / V;
    using vec_t = typename std::conditional<sizeof(T) == 4, float4, double2>::type;
    const vec_t* gv = reinterpret_cast<const vec_t*>(g);
    vec_t* lv = reinterpret_cast<vec_t*>(lds);
    for (int i = threadIdx.x; i < nvec; i += kBlk) lv[i] = gv[i];
    for (int i = nvec * V + threadIdx.x; i < count; i += kBlk) lds[i] = g[i];
  } else {
    for (int i = threadIdx.x; i < count; i += kBlk) lds[i] = g[i];
  }
}
template <typename T>
__device__ __forceinline__ void store_tile(T* __restrict__ g, const T* lds, int count, bool vec_ok) {
  if (vec_ok) {
    constexpr int V = 16 / sizeof(T);
    const int nvec = count / V;
    using vec_t = typename std::conditional<sizeof(T) == 4, float4, double2>::type;
    vec_t* gv = reinterpret_cast<vec_t*>(g);
    const vec_t* lv = reinterpret_cast<const vec_t*>(lds);
    for (int i = threadIdx.x; i < nvec; i += kBlk) gv[i] = lv[i];
    for (int i = nvec * V + threadIdx.x; i < count; i += kBlk) g[i] = lds[i];
  } else {
    for (int i = threadIdx.x; i < count; i += kBlk) g[i] = lds[i];
  }
}

template <typename T>
__global__ __launch_bounds__(kBlk) void inv3x3_fwd_kernel(const T* __restrict__ ms, T* __restrict__ invs,
                                                          uint8_t* __restrict__ checks, int64_t n,
                                                          bool vec_ok) {
  __shared__ __attribute__((aligned(16))) T tile[kBlk * 9];
  for (int64_t base = (int64_t)blockIdx.x * kBlk; base < n; base += (int64_t)gridDim.x * kBlk) {
    const int cnt = (int)((n - base < kBlk) ? (n - base) : kBlk);
    load_tile(ms + base * 9, tile, cnt * 9, vec_ok);
    __syncthreads();
    T m[9], inv[9];
    bool ok = false;
    if ((int)threadIdx.x < cnt) {
#pragma unroll
      for (int i = 0; i < 9; ++i) m[i] = tile[threadIdx.x * 9 + i];
      ok = inv_one(m, inv);
    }
    __syncthreads();
    if ((int)threadIdx.x < cnt) {
#pragma unroll
      for (int i = 0; i < 9; ++i) tile[threadIdx.x * 9 + i] = inv[i];
      checks[base + threadIdx.x] = ok ? 1 : 0;
    }
    __syncthreads();
    store_tile(invs + base * 9, tile, cnt * 9, vec_ok);
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(kBlk) void inv3x3_bwd_kernel(const T* __restrict__ grads,
                                                          const T* __restrict__ invs, T* __restrict__ outs,
                                                          int64_t n, bool vec_ok) {
  __shared__ __attribute__((aligned(16))) T tg[kBlk * 9];
  __shared__ __attribute__((aligned(16))) T tc[kBlk * 9];
  for (int64_t base = (int64_t)blockIdx.x * kBlk; base < n; base += (int64_t)gridDim.x * kBlk) {
    const int cnt = (int)((n - base < kBlk) ? (n - base) : kBlk);
    load_tile(grads + base * 9, tg, cnt * 9, vec_ok);
    load_tile(invs + base * 9, tc, cnt * 9, vec_ok);
    __syncthreads();
    T g[9], c[9], o[9];
    if ((int)threadIdx.x < cnt) {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        g[i] = tg[threadIdx.x * 9 + i];
        c[i] = tc[threadIdx.x * 9 + i];
      }
      inv_bwd_one(g, c, o);
    }
    __syncthreads();
    if ((int)threadIdx.x < cnt) {
#pragma unroll
      for (int i = 0; i < 9; ++i) tg[threadIdx.x * 9 + i] = o[i];
    }
    __syncthreads();
    store_tile(outs + base * 9, tg, cnt * 9, vec_ok);
    __syncthreads();
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int recmv_inv3x3_forward(const void* ms, void* invs, uint8_t* checks, int64_t n, int dtype,
                                    void* stream) {
  RECMV_REQUIRE(n >= 0, "inv3x3_forward: n=%lld < 0", (long long)n);
  if (n == 0) return RECMV_OK;
  RECMV_REQUIRE(ms && invs && checks, "inv3x3_forward: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  const int grid = stream_grid(n, kBlk);
  // every workgroup chunk starts at base*9 elements: 256*9*sizeof(T) is a multiple of 16
  const bool vec_ok = aligned16(ms) && aligned16(invs);
  if (dtype == RECMV_F32) {
    hipLaunchKernelGGL(inv3x3_fwd_kernel<float>, dim3(grid), dim3(kBlk), 0, s, (const float*)ms,
                       (float*)invs, checks, n, vec_ok);
  } else if (dtype == RECMV_F64) {
    hipLaunchKernelGGL(inv3x3_fwd_kernel<double>, dim3(grid), dim3(kBlk), 0, s, (const double*)ms,
                       (double*)invs, checks, n, vec_ok);
  } else {
    set_error("inv3x3_forward: dtype %d unsupported (f32|f64 only, as the reference)", dtype);
    return RECMV_ERR_UNSUPPORTED;
  }
  return check_launch("inv3x3_forward");
}

extern "C" int recmv_inv3x3_backward(const void* grads, const void* invs, void* outs, int64_t n,
                                     int dtype, void* stream) {
  RECMV_REQUIRE(n >= 0, "inv3x3_backward: n=%lld < 0", (long long)n);
  if (n == 0) return RECMV_OK;
  RECMV_REQUIRE(grads && invs && outs, "inv3x3_backward: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  const int grid = stream_grid(n, kBlk);
  const bool vec_ok = aligned16(grads) && aligned16(invs) && aligned16(outs);
  if (dtype == RECMV_F32) {
    hipLaunchKernelGGL(inv3x3_bwd_kernel<float>, dim3(grid), dim3(kBlk), 0, s, (const float*)grads,
                       (const float*)invs, (float*)outs, n, vec_ok);
  } else if (dtype == RECMV_F64) {
    hipLaunchKernelGGL(inv3x3_bwd_kernel<double>, dim3(grid), dim3(kBlk), 0, s, (const double*)grads,
                       (const double*)invs, (double*)outs, n, vec_ok);
  } else {
    set_error("inv3x3_backward: dtype %d unsupported", dtype);
    return RECMV_ERR_UNSUPPORTED;
  }
  return check_launch("inv3x3_backward");
}
