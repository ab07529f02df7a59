/*
 * ref_minv.cpp — TEST INFRASTRUCTURE (oracle/_ref): runs the REFERENCE's own 3x3 inverse kernels on the host.
 *
 * The kernel text (`index`, cu3x3MInv<scalar_t>, cu3x3MInv_backward<scalar_t> =
 * /root/reference/FastMinv/Matrix3x3InvKernels.cu:18-104) is cut out of the reference tree at build time into
 * oracle/_ref/minv_kernels_extract.inc (git-ignored build output, oracle/Makefile) and compiled through
 * cuda_host_shim.h.  Written here: only the launch wrappers M3x3Inv_float/_double/_backward_* (ibid. :106-145),
 * one "thread" per matrix instead of `<<<blocks,1024>>>`.
 * Built with -ffp-contract=off: the cofactor expressions `a*b-c*d` have no unique contraction (nvcc picks one product
 * to fuse, compilers differ), so the uncontracted IEEE sequence is the one the oracle and the HIP kernel follow.
 */
#include "cuda_host_shim.h"

#include "minv_kernels_extract.inc"

template <typename T>
static void run_fwd(const T* ms, T* invs, bool* checks, int N) {
  blockDim.x = 1; threadIdx.x = 0;
  for (int m = 0; m < N; ++m) { blockIdx.x = m; cu3x3MInv<T>(ms, invs, checks, N); }
}
template <typename T>
static void run_bwd(const T* grads, const T* invs, T* outs, int N) {
  blockDim.x = 1; threadIdx.x = 0;
  for (int m = 0; m < N; ++m) { blockIdx.x = m; cu3x3MInv_backward<T>(grads, invs, outs, N); }
}

extern "C" {
void ref_M3x3Inv_float(const float* ms, float* invs, bool* checks, int N) { run_fwd<float>(ms, invs, checks, N); }
void ref_M3x3Inv_double(const double* ms, double* invs, bool* checks, int N) { run_fwd<double>(ms, invs, checks, N); }
void ref_M3x3Inv_backward_float(const float* g, const float* invs, float* outs, int N) { run_bwd<float>(g, invs, outs, N); }
void ref_M3x3Inv_backward_double(const double* g, const double* invs, double* outs, int N) { run_bwd<double>(g, invs, outs, N); }
}
