/*
 * ref_mc.cpp — TEST INFRASTRUCTURE (oracle/_ref): runs the REFERENCE's own marching-cubes kernels on the host.
 *
 * The kernel text (tables + d_fGetOffset + d_mc_get_mesh_on_gpu + d_conver_ijkd_to_pindex + d_set_int +
 * d_scale_vertices = /root/reference/MCGpu/CudaKernels.cu:4-521) is NOT in this repository: oracle/Makefile cuts it
 * out of the reference tree at build time into oracle/_ref/mc_kernels_extract.inc (a git-ignored build output) and
 * this file compiles it through cuda_host_shim.h.  What is written here is only the host-side sequence of
 * MCGpu::init / MC / scaleVertices (CudaKernels.cu:572-639) and mc_gpu (MCGpu.cpp:20-56) with malloc in place of
 * cudaMalloc — the part of the reference that cannot be compiled without nvcc (`<<<...>>>`).
 *
 * Differences from a run of the reference on an NVIDIA GPU, all irrelevant after canonicalisation:
 *   - the kernel loop is serial (order = index order, or scrambled: ref_mc_set_loop_stride), so "atomic" ids are
 *     handed out in loop order instead of in hardware scheduling order;
 *   - the output buffers are sized for the worst case (3 vertices and 5 triangles per voxel); the reference sizes
 *     them for 5 % of that (CudaKernels.cu:590-592) and silently overruns on denser volumes — `*exceeds_ref_capacity`
 *     reports when the reference itself would have.
 * Build flags matter for the last bit of the scaled vertices: nvcc's default -fmad=true contracts
 * `v*step+min` (CudaKernels.cu:517-519) into one fma; libref_mc_fma.so is built with -ffp-contract=fast -mfma (same
 * contraction), libref_mc_nofma.so with -ffp-contract=off.
 */
#include "cuda_host_shim.h"

typedef long int ref_long;   /* `long int` of CudaKernels.cu: 64-bit on this ABI, the int64 faces of MCGpu.cpp:33-37 */

#include "mc_kernels_extract.inc"

extern "C" {

void ref_mc_set_loop_stride(long stride) { ref_loop_stride = stride > 0 ? stride : 1; }

/* Phase 1 (init + MC): returns counts; keeps the state in a heap context. */
struct ref_mc_ctx {
  int NX, NY, NZ;
  int number_record[2];
  int* edge_point_state;   /* [NX*NY*NZ*3], -1 = no vertex on that lattice edge (CudaKernels.cu:589) */
  float* points_coor;      /* lattice-space vertex positions until ref_mc_scale */
  ref_long* faces_index;
  int* faces_ijkd;
};

ref_mc_ctx* ref_mc_run(const float* sdf, int nx, int ny, int nz, float target, int* n_verts, int* n_faces,
                       int* exceeds_ref_capacity) {
  if (nx <= 0 || ny <= 0 || nz <= 0) return NULL;                                   /* MCGpu::init :574-575 */
  ref_mc_ctx* c = (ref_mc_ctx*)calloc(1, sizeof(ref_mc_ctx));
  const size_t n = (size_t)nx * ny * nz;
  c->NX = nx; c->NY = ny; c->NZ = nz;
  c->edge_point_state = (int*)malloc(sizeof(int) * n * 3);
  c->points_coor = (float*)malloc(sizeof(float) * 3 * n * 3);
  c->faces_index = (ref_long*)malloc(sizeof(ref_long) * 3 * n * 5);
  c->faces_ijkd = (int*)malloc(sizeof(int) * 3 * 4 * n * 5);
  const long keep = ref_loop_stride;
  ref_loop_stride = 1;
  d_set_int((int)(n * 3), -1, c->edge_point_state);                                  /* :589 */
  ref_loop_stride = keep;
  /* MCGpu::MC :620-634 — the constant tables are passed as the flat device copies the constructor makes (:535-552) */
  d_mc_get_mesh_on_gpu((int)n, const_cast<float*>(sdf), c->edge_point_state, nx, ny, nz, target, c->number_record,
                       const_cast<float*>(&a2fVertexOffset[0][0]), const_cast<int*>(&a2iEdgeConnection[0][0]),
                       const_cast<float*>(&a2fEdgeDirection[0][0]), const_cast<int*>(&aiCubeEdgeFlags[0]),
                       const_cast<int*>(&a2iTriangleConnectionTable[0][0]), c->points_coor, c->faces_ijkd,
                       c->faces_index);
  ref_loop_stride = 1;
  d_conver_ijkd_to_pindex(c->number_record[1], nx, ny, nz, c->edge_point_state, c->faces_ijkd, c->faces_index,
                          c->number_record);
  ref_loop_stride = keep;
  *n_verts = c->number_record[0];
  *n_faces = c->number_record[1];
  if (exceeds_ref_capacity)
    *exceeds_ref_capacity = (c->number_record[0] > (int)(n * 12 * 0.05)) || (c->number_record[1] > (int)(n * 5 * 0.05));
  return c;
}

/* scaleVertices :635-639 + the copies of MCGpu.cpp:51-54.  `edge_state` (optional) receives the lattice-edge ->
 * vertex-id table, from which the canonical vertex order (ascending edge key) is read off exactly. */
void ref_mc_fetch(ref_mc_ctx* c, float xstep, float ystep, float zstep, float xmin, float ymin, float zmin,
                  float* vertices, int64_t* faces, int32_t* edge_state) {
  const long keep = ref_loop_stride;
  ref_loop_stride = 1;
  d_scale_vertices(c->number_record[0], xstep, ystep, zstep, xmin, ymin, zmin, c->points_coor);
  ref_loop_stride = keep;
  memcpy(vertices, c->points_coor, sizeof(float) * 3 * (size_t)c->number_record[0]);
  for (size_t i = 0; i < (size_t)3 * c->number_record[1]; ++i) faces[i] = (int64_t)c->faces_index[i];
  if (edge_state) memcpy(edge_state, c->edge_point_state, sizeof(int) * (size_t)c->NX * c->NY * c->NZ * 3);
}

void ref_mc_free(ref_mc_ctx* c) {
  if (!c) return;
  free(c->edge_point_state); free(c->points_coor); free(c->faces_index); free(c->faces_ijkd); free(c);
}

}  /* extern "C" */
