/*
 * ref_gs3d.cpp — TEST INFRASTRUCTURE (oracle/_ref): runs the REFERENCE's own 3-D grid sampler kernels on the host.
 *
 * The kernel text — the anonymous namespace of /root/reference/MCAcc/cuda/GridSamplerMineKernel.cu:29-914: the coordinate
 * helpers, grid_sampler_3d_kernel, grid_sampler_3d_backward_kernel, grid_sampler_3d_backward_backward_kernel — is cut out of
 * the reference tree at build time into oracle/_ref/gs3d_kernels_extract.inc (git-ignored build output, oracle/Makefile) and
 * compiled through torch_host_shim.h.  Written here: only the launch wrappers (ibid. :917-1022: output allocation is the
 * caller's, `count = N*D*H*W`, one call per kernel — CUDA_KERNEL_LOOP runs every index serially, in index order or in a
 * scrambled order, which only matters for the order of the atomicAdd into grad_input).
 * Two builds: with and without fma contraction (nvcc contracts `a*b+c` by default; which product of a longer expression it
 * fuses is its own choice, so the tests say which outputs are bit-equal under which build and bound the rest in ulps).
 */
#include "torch_host_shim.h"

#include "gs3d_kernels_extract.inc"

template <typename T>
static TensorInfo<T, int> info(const T* data, const int* sizes, const int* strides) {
  TensorInfo<T, int> t;
  t.data = const_cast<T*>(data);
  for (int i = 0; i < 5; ++i) { t.sizes[i] = sizes[i]; t.strides[i] = strides[i]; }
  return t;
}

template <typename T>
static void fwd(const T* in, const int* is, const int* ist, const T* gr, const int* gs, const int* gst, T* out, const int* os,
                const int* ost, int interp, int pad) {
  const int count = gs[0] * gs[1] * gs[2] * gs[3];
  if (count > 0)
    grid_sampler_3d_kernel<T>(count, info(in, is, ist), info(gr, gs, gst), info(out, os, ost),
                              static_cast<GridSamplerInterpolation>(interp), static_cast<GridSamplerPadding>(pad));
}

template <typename T>
static void bwd(const T* go, const int* gos, const int* gost, const T* in, const int* is, const int* ist, const T* gr,
                const int* gs, const int* gst, T* gi, const int* gis, const int* gist, T* gg, const int* ggs, const int* ggst,
                int interp, int pad) {
  const int count = gs[0] * gs[1] * gs[2] * gs[3];
  if (count > 0)
    grid_sampler_3d_backward_kernel<T>(count, info(go, gos, gost), info(in, is, ist), info(gr, gs, gst), info(gi, gis, gist),
                                       info(gg, ggs, ggst), static_cast<GridSamplerInterpolation>(interp),
                                       static_cast<GridSamplerPadding>(pad));
}

template <typename T>
static void dbwd(const T* goi, const int* gois, const int* goist, const T* gog, const int* gogs, const int* gogst, const T* go,
                 const int* gos, const int* gost, const T* in, const int* is, const int* ist, const T* gr, const int* gs,
                 const int* gst, T* gi, const int* gis, const int* gist, T* gg, const int* ggs, const int* ggst, T* ggo,
                 const int* ggos, const int* ggost, int interp, int pad) {
  const int count = gs[0] * gs[1] * gs[2] * gs[3];
  if (count > 0)
    grid_sampler_3d_backward_backward_kernel<T>(count, info(goi, gois, goist), info(gog, gogs, gogst), info(go, gos, gost),
                                                info(in, is, ist), info(gr, gs, gst), info(gi, gis, gist), info(gg, ggs, ggst),
                                                info(ggo, ggos, ggost), static_cast<GridSamplerInterpolation>(interp),
                                                static_cast<GridSamplerPadding>(pad));
}

#define REF_GS3D(SUFFIX, T)                                                                                                  \
  void ref_gs3d_forward_##SUFFIX(const T* in, const int* is, const int* ist, const T* gr, const int* gs, const int* gst,    \
                                 T* out, const int* os, const int* ost, int interp, int pad) {                               \
    fwd<T>(in, is, ist, gr, gs, gst, out, os, ost, interp, pad);                                                            \
  }                                                                                                                          \
  void ref_gs3d_backward_##SUFFIX(const T* go, const int* gos, const int* gost, const T* in, const int* is, const int* ist, \
                                  const T* gr, const int* gs, const int* gst, T* gi, const int* gis, const int* gist,       \
                                  T* gg, const int* ggs, const int* ggst, int interp, int pad) {                             \
    bwd<T>(go, gos, gost, in, is, ist, gr, gs, gst, gi, gis, gist, gg, ggs, ggst, interp, pad);                             \
  }                                                                                                                          \
  void ref_gs3d_dbackward_##SUFFIX(const T* goi, const int* gois, const int* goist, const T* gog, const int* gogs,          \
                                   const int* gogst, const T* go, const int* gos, const int* gost, const T* in,             \
                                   const int* is, const int* ist, const T* gr, const int* gs, const int* gst, T* gi,        \
                                   const int* gis, const int* gist, T* gg, const int* ggs, const int* ggst, T* ggo,         \
                                   const int* ggos, const int* ggost, int interp, int pad) {                                 \
    dbwd<T>(goi, gois, goist, gog, gogs, gogst, go, gos, gost, in, is, ist, gr, gs, gst, gi, gis, gist, gg, ggs, ggst, ggo,  \
            ggos, ggost, interp, pad);                                                                                       \
  }

extern "C" {
REF_GS3D(float, float)
REF_GS3D(double, double)
void ref_gs3d_set_loop_stride(long s) { ref_loop_stride = s; }
}
