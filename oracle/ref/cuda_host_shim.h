/*
 * cuda_host_shim.h — TEST INFRASTRUCTURE (oracle/): the few CUDA spellings the reference's kernels use, given plain
 * host meanings so that the kernel bodies of /root/reference/MCGpu/CudaKernels.cu and
 * /root/reference/FastMinv/Matrix3x3InvKernels.cu compile with g++ and run serially on the host.  Nothing under
 * rec-mv_amd/ includes this.
 *
 *   __global__ / __device__ / __host__   -> nothing (ordinary functions)
 *   threadIdx / blockIdx / blockDim       -> plain structs the driver sets before every call (one "thread" at a time)
 *   CUDA_KERNEL_LOOP(i, n)                -> a serial loop over 0..n-1 in the order t -> (t * ref_loop_stride) mod n.
 *                                            stride 1 is index order; any stride coprime with n visits every index
 *                                            once in a scrambled order — used to show that the canonicalised result
 *                                            does not depend on the order in which the reference's atomics fire.
 *   atomicAdd / atomicExch (int)          -> the read-modify-write they are, without the atomicity nobody needs here.
 */
#ifndef RECMV_ORACLE_CUDA_HOST_SHIM_H
#define RECMV_ORACLE_CUDA_HOST_SHIM_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __inline__ inline

struct ref_dim3 { int x, y, z; };
static ref_dim3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0}, blockDim = {1, 1, 1}, gridDim = {1, 1, 1};

static long ref_loop_stride = 1;
#define CUDA_KERNEL_LOOP(i, n)                                                               \
  for (long ref_t_ = 0, ref_n_ = (n), i = 0;                                                 \
       ref_t_ < ref_n_ && ((i = (ref_t_ * ref_loop_stride) % ref_n_), true); ++ref_t_)

static inline int atomicAdd(int* p, int v) { int old = *p; *p = old + v; return old; }
static inline int atomicExch(int* p, int v) { int old = *p; *p = v; return old; }
static inline int atomicMax(int* p, int v) { int old = *p; if (v > old) *p = v; return old; }
#endif
