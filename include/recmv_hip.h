/*
 * recmv_hip.h — C ABI of librecmv_hip.so, the MI355X (gfx950) implementation of the
 * REC-MV per-frame implicit-surface optimisation hot path.
 *
 * This is the drop-in boundary: every entry point replaces one function of the reference's
 * pybind11/CUDA extensions (FastMinv, MCGpu, GridSamplerMine, interp2x_boundary3d) or one dense
 * contraction that the reference leaves to torch/cuBLAS inside its nn.Modules.  Plain pointers and
 * sizes only — no torch types.  All pointers are DEVICE pointers unless the name says `host`.
 * `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  Every function
 * returns RECMV_OK (0) or a negative RECMV_ERR_* code; recmv_last_error() gives a message for the
 * calling thread.  The library never allocates caller-visible memory; marching cubes is two-phase
 * (count -> caller allocates -> emit) and keeps a caller-provided workspace.
 *
 * Reference citations are relative to the REC-MV tree.
 */
#ifndef RECMV_HIP_H_
#define RECMV_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RECMV_OK               0
#define RECMV_ERR_ARG         -1   /* bad argument (NULL pointer, negative size, unsupported mode) */
#define RECMV_ERR_HIP         -2   /* a HIP runtime call or kernel launch failed                  */
#define RECMV_ERR_UNSUPPORTED -3   /* combination not implemented (e.g. dtype)                    */
#define RECMV_ERR_WORKSPACE   -4   /* caller workspace too small                                  */

#define RECMV_F32 0
#define RECMV_F64 1

/* ABI version, bumped on any signature change. */
int recmv_abi_version(void);
/* 1 when every kernel of the library was built without packed-f32 VALU instructions (RECMV_NO_PACKED_F32=1 at build time): the build
 * the optional bf16x6 matrix mode needs — beside its NT product kernels, waves executing v_pk_*_f32 were caught computing wrong values in
 * lanes 48-63.  (ABI v7) */
int recmv_no_packed_f32(void);
/* Message of the last error on this thread ("" if none). */
const char* recmv_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * A. FastMinv — batched 3x3 inverse.
 *   replaces Fast3x3Minv            (FastMinv/M3x3Inv.cpp:12-36, kernel Matrix3x3InvKernels.cu:22-61)
 *            Fast3x3Minv_backward   (FastMinv/M3x3Inv.cpp:38-59, kernel Matrix3x3InvKernels.cu:64-104)
 * ms/invs/grads/outs: [n,3,3] contiguous, dtype f32|f64.  checks: [n] bytes (0/1) == torch.bool.
 * Singular rule: fabs(det) < 1e-4 (absolute) -> inverse = 0, check = 0.
 * ---------------------------------------------------------------------------------------------- */
int recmv_inv3x3_forward(const void* ms, void* invs, uint8_t* checks, int64_t n, int dtype, void* stream);
int recmv_inv3x3_backward(const void* grads, const void* invs, void* outs, int64_t n, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A2. Deformation regulariser of the render loss: value and gradient of GM(sum_i log^2 sigma_i(J)) per matrix.
 *   replaces the host SVD + its autograd of OptimGarmentNetwork.py:1143-1155
 *            (Jacobs.cpu() -> torch.svd -> log -> utils.GMRobustError(., def_regu.c, True), utils/utils.py:87-91).
 * J: [P,3,3] contiguous f32 (Jacobian of the offset MLP at the sampled points).  y: [P] = 2 x / c^2 / (x / c^2 + 4) with
 * x = sum_i log^2 sigma_i;  gJ: [P,3,3] = dy/dJ (singular values below 1e-10 are clamped there: value from the bound, no gradient).
 * The caller takes the mean and scales gJ in its backward pass.
 * ---------------------------------------------------------------------------------------------- */
int recmv_def_regu(const float* J, int64_t P, float c, float* y, float* gJ, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A3. Camera of the loop (ABI v8): world points -> image coordinates, pixels -> world rays, with their backward passes.
 *   replaces the element-wise torch chains of RectifiedPerspectiveCameras, model/CameraMine.py:62-88 + 281-300 (the calibration
 *            matrix behind transform_points: fx' = fx / (W/2), px' = 1 - 1/W - px / (W/2)), :104-142 (transform_points_screen),
 *            :146-169 (view_rays) and `project`.
 * cam16: 16 device floats = R [3][3] row-major (world -> view: v = p R + T), T [3], focal length (fx, fy), principal point (px, py).
 * W, H: image size in pixels.  mode: 0 -> out [P,3] = (x_ndc, y_ndc, z_view); 1 -> out [P,3] = (screen_x, screen_y, 1 / z_view),
 *   screen = (S - 1) / 2 - S * ndc / 2; 2 -> out [P,2] = pixel (px - x fx / z, py - y fy / z).
 * Backward: g_pts [P,3] (may be NULL) and g_cam7 = d/d(T[3], focal[2], principal point[2]) summed over the points in a FIXED order
 *   (per-workgroup partial sums into `partial`, recmv_cam_partial_floats(P) floats owned by the caller, then one pass over them):
 *   bit-reproducible, no float atomics.  R gets no gradient (the loop's camera rotation is not optimised).
 * Rays: pixel (x, y, w) as [P,3] floats in `pix`, or — pix NULL — integer (col, row) with w = 1 -> unit world ray
 *   normalize(-x / fx + w px / fx, -y / fy + w py / fy, w) R^T; backward: g_cam4 = d/d(focal[2], principal point[2]).
 * ---------------------------------------------------------------------------------------------- */
int64_t recmv_cam_partial_floats(int64_t P);
int recmv_cam_project(const float* pts, int64_t P, const float* cam16, float W, float H, int mode, float* out, void* stream);
int recmv_cam_project_backward(const float* pts, const float* g_out, int64_t P, const float* cam16, float W, float H, int mode,
                               float* g_pts, float* g_cam7, float* partial, int64_t partial_floats, void* stream);
int recmv_cam_rays(const float* pix, const int64_t* col, const int64_t* row, int64_t P, const float* cam16, float* out, void* stream);
int recmv_cam_rays_backward(const float* pix, const int64_t* col, const int64_t* row, const float* g_out, int64_t P,
                            const float* cam16, float* g_cam4, float* partial, int64_t partial_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * C. GridSamplerMine — 3-D trilinear sampler, padding=border, align_corners=False, with first and
 * second derivative.
 *   replaces GridSamplerMine.forward / backward / dbackward
 *            (MCAcc/cuda/GridSamplerMine.cpp:73-96; kernels GridSamplerMineKernel.cu:162-328,
 *             333-570, 575-914).
 * Tensors are described by sizes/strides in ELEMENTS (like the reference's TensorInfo), so any
 * strided layout is accepted.  input: [N,C,D,H,W]; grid: [N,Do,Ho,Wo,3]; output/grad_output:
 * [N,C,Do,Ho,Wo].  A channels-last input (stride[1]==1) with contiguous grid takes the vectorised
 * fast path.  interp must be 0 (bilinear) and pad 1 (border), as in the reference's check()
 * (GridSamplerMine.cpp:58-63) — anything else returns RECMV_ERR_UNSUPPORTED.
 * ---------------------------------------------------------------------------------------------- */
typedef struct recmv_tensor5 {
  int64_t size[5];
  int64_t stride[5];
} recmv_tensor5;

int recmv_grid_sample3d_forward(const void* input, const recmv_tensor5* input_desc,
                                const void* grid, const recmv_tensor5* grid_desc,
                                void* output, const recmv_tensor5* output_desc,
                                int interp, int pad, int dtype, void* stream);

/* grad_input may be NULL: the scatter into the (frozen) volume is skipped — the hot path's volume
 * is a registered buffer, never a parameter (model/Deformer.py:240-243).  When non-NULL it must be
 * zero-filled by the caller (the reference zero-fills it itself: GridSamplerMineKernel.cu:955).
 * grad_grid: [N,Do,Ho,Wo,3] contiguous (the reference assumes this too: ...Kernel.cu:538-545). */
/* Summation order of backward / double backward when only grad_grid (and grad_grad_output) are asked for on a channels-last f32
 * volume: 0 (default) = record-coalesced lanes — G lanes per point, per-point sums finished by a lane butterfly: the reference's
 * terms in a different order, within a few ulp of sum |terms| (north_star: gradients within an f32 tolerance); 1 = one lane per
 * point with the channel loop in the reference's order (GridSamplerMineKernel.cu:333-914), bit-equal to the oracle.  Requests
 * with grad_input or ggI, f64, or other layouts always take the exact kernels.  Returns the previous mode. */
int recmv_set_sampler_mode(int mode);
/* The mode in force, read without changing it (ABI v7; GridSamplerMine.current_mode of the Python side reads it at forward time). */
int recmv_get_sampler_mode(void);

int recmv_grid_sample3d_backward(const void* input, const recmv_tensor5* input_desc,
                                 const void* grid, const recmv_tensor5* grid_desc,
                                 const void* grad_output, const recmv_tensor5* grad_output_desc,
                                 void* grad_input, const recmv_tensor5* grad_input_desc,
                                 void* grad_grid,
                                 int interp, int pad, int dtype, void* stream);

/* Double backward.  ggI = grad wrt backward's grad_input output (layout like input; may be NULL ->
 * treated as zeros), ggG = grad wrt backward's grad_grid output ([N,Do,Ho,Wo,3], strided).
 * Outputs: grad_input (like input; NULL to skip; caller zero-fills), grad_grid (contiguous),
 * grad_grad_output (like grad_output; written, not accumulated). */
int recmv_grid_sample3d_dbackward(const void* ggI, const recmv_tensor5* ggI_desc,
                                  const void* ggG, const recmv_tensor5* ggG_desc,
                                  const void* input, const recmv_tensor5* input_desc,
                                  const void* grid, const recmv_tensor5* grid_desc,
                                  const void* grad_output, const recmv_tensor5* grad_output_desc,
                                  void* grad_input, const recmv_tensor5* grad_input_desc,
                                  void* grad_grid,
                                  void* grad_grad_output, const recmv_tensor5* grad_grad_output_desc,
                                  int interp, int pad, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * C2. interp2x_boundary3d — (2n-1) trilinear upsample + boundary mask.
 *   replaces interp2x_boundary3d.forward / backward
 *            (MCAcc/cuda/interp2x_boundary3d.cpp:17-37; kernels interp2x_boundary3d_kernel.cu:11-239)
 * input: [B,C,d,h,w] contiguous -> output [B,C,2d-1,2h-1,2w-1], is_boundary same shape (bytes).
 * ---------------------------------------------------------------------------------------------- */
int recmv_interp2x_boundary3d_forward(const void* input, void* output, uint8_t* is_boundary,
                                      int64_t bc, int64_t d, int64_t h, int64_t w,
                                      float balance_value, int dtype, void* stream);
/* grad_output: [B,C,D,H,W] contiguous (D,H,W odd) -> grad_input [B,C,(D+1)/2,(H+1)/2,(W+1)/2]. */
int recmv_interp2x_boundary3d_backward(const void* grad_output, void* grad_input,
                                       int64_t bc, int64_t D, int64_t H, int64_t W,
                                       int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * E. MCGpu — marching cubes on an x-major f32 volume sdf[NX][NY][NZ] (index i*NY*NZ+j*NZ+k).
 *   replaces MCGpu.mc_gpu (MCGpu/MCGpu.cpp:20-56; MCGpu::init/MC/scaleVertices
 *            MCGpu/CudaKernels.cu:572-639; kernels :316-521).
 * Deterministic: vertices ordered by edge key ((x*NY+y)*NZ+z)*3+dir, faces by (voxel, triangle#),
 * corner order reversed exactly as the reference (CudaKernels.cu:502).  An edge whose owner voxel is
 * outside the grid has no vertex and is referenced as -1, as in the reference.
 *
 * recmv_mc_run is the whole extraction on the stream with NO host round trip (the reference reads its counters
 * back between its two kernels, CudaKernels.cu:628): the caller passes output buffers with capacities, the
 * kernels never write past them, and `counts_device` (3 x int32 {n_vertices, n_faces, n_active_voxels}, DEVICE
 * pointer) receives the true sizes — if one exceeds its capacity the caller re-allocates and calls
 * recmv_mc_emit with the same workspace (the classification is still in it).
 * Two-phase form: recmv_mc_count classifies + scans and returns the sizes through `counts_host` (HOST pointer,
 * written after an internal stream sync — the same D2H round trip as the reference's); the caller allocates
 * vertices [V,3] f32 and faces [F,3] i64 and calls recmv_mc_emit with the same workspace and n_active_voxels
 * (the number of voxels that own a vertex or a triangle: recmv_mc_emit runs one lane per such voxel from the
 * list the scan left in the workspace).
 * recmv_mc_workspace_bytes gives the workspace size for a volume (0 if the volume is not supported).
 * Limits: at most 2^28 lattice points, every dimension < 32768.
 * ---------------------------------------------------------------------------------------------- */
int64_t recmv_mc_workspace_bytes(int64_t nx, int64_t ny, int64_t nz);
int recmv_mc_count(const float* sdf, int64_t nx, int64_t ny, int64_t nz, float iso,
                   void* workspace, int64_t workspace_bytes, int32_t* counts_host, void* stream);
int recmv_mc_emit(const float* sdf, int64_t nx, int64_t ny, int64_t nz, float iso,
                  float xstep, float ystep, float zstep, float xmin, float ymin, float zmin,
                  const void* workspace, int64_t workspace_bytes, int64_t n_active_voxels,
                  float* vertices, int64_t vertex_capacity, int64_t* faces, int64_t face_capacity, void* stream);
int recmv_mc_run(const float* sdf, int64_t nx, int64_t ny, int64_t nz, float iso,
                 float xstep, float ystep, float zstep, float xmin, float ymin, float zmin,
                 void* workspace, int64_t workspace_bytes,
                 float* vertices, int64_t vertex_capacity, int64_t* faces, int64_t face_capacity,
                 int32_t* counts_device, void* stream);
/* The same for up to 4 volumes of ONE lattice size in one set of four launches (grid y = volume): the three nets of a re-mesh —
 * body + two garments, engineer/networks/OptimGarmentNetwork.py:581-618 calls MCGpu.mc_gpu once per net — share every launch; the
 * passes after the volume stream work on kilobytes and are launch-latency bound.  sdf / workspaces / vertices / faces /
 * counts_device: HOST arrays of n device pointers (every volume its own workspace of recmv_mc_workspace_bytes); capacities per
 * volume; results per volume exactly those of recmv_mc_run. */
int recmv_mc_run_batch(int n, const float* const* sdf, int64_t nx, int64_t ny, int64_t nz, float iso, float xstep, float ystep,
                       float zstep, float xmin, float ymin, float zmin, void* const* workspaces, int64_t workspace_bytes,
                       float* const* vertices, const int64_t* vertex_capacity, int64_t* const* faces,
                       const int64_t* face_capacity, int32_t* const* counts_device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * E'. Seg3dLossless bookkeeping on the device (csrc/seg3d.hip) — the per-level steps of
 *   MCAcc/seg3d_lossless.py:_forward (:233-428) between the upsampler (C above) and the MLP queries: which voxels of
 *   the level to query, their world points, the write-back with sign-conflict detection, and the growth of the
 *   conflicts' 3^3 neighbourhoods.  Volumes are [D,H,W], x fastest, linear voxel index i = (z*H + y)*W + x (int32);
 *   the set of evaluated voxels is a bit volume (uint32 words, bit i&31 of word i>>5; ceil(D*H*W/32)+1 words).
 *   Every counter is a DEVICE int32 that the call zeroes first; lists are never written past `capacity`.
 *   recmv_seg3d_select : list = voxels with dilate3x3x3(is_boundary) != 0 that are not evaluated, where "evaluated" =
 *                        bit of the parent level (sizes (D+1)/2 ...) at even coordinates; `done` receives
 *                        evaluated | listed for the whole level (replaces: box filter, coords_accum erase, transposed
 *                        nonzero, unique(dim=1) of :296-330).  List order is unspecified.
 *   recmv_seg3d_points : points[t] = ((x*sx, y*sy, z*sz) / res + (1/res)/2) * extent + bmin  (batch_eval, :99-101)
 *   recmv_seg3d_apply  : flags[t] = (occupancy[i] - balance) * (values[t] - balance) < 0, then occupancy[i] = values[t]
 *   recmv_seg3d_expand : list_out = not-yet-evaluated voxels of the 3^3 blocks (clamped to the grid) around the flagged
 *                        voxels, each exactly once; their bits are set in `done` (:348-372 without unique(dim=0))
 * ---------------------------------------------------------------------------------------------- */
int recmv_seg3d_select(const uint8_t* is_boundary, const uint32_t* done_prev, int64_t D, int64_t H, int64_t W,
                       uint32_t* done, int32_t* list, int64_t capacity, int32_t* count_device, void* stream);
int recmv_seg3d_points(const int32_t* list, int64_t n, int64_t H, int64_t W, const int32_t* stride_xyz,
                       const float* res_xyz, const float* extent_xyz, const float* bmin_xyz, float* points,
                       void* stream);
int recmv_seg3d_apply(const int32_t* list, const float* values, int64_t n, float balance, float* occupancy,
                      uint8_t* conflict_flags, int32_t* conflict_count_device, void* stream);
int recmv_seg3d_expand(const int32_t* list, const uint8_t* conflict_flags, int64_t n, int64_t D, int64_t H, int64_t W,
                       uint32_t* done, int32_t* list_out, int64_t capacity, int32_t* count_device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A/B/D. Dense f32 contractions of the three MLPs (SDF model/network.py:98-111, deformer
 * model/Deformer.py:194-199, colour model/RenderNet.py:83-94) — torch.nn.Linear/cuBLAS sgemm in the
 * reference.  f32 MFMA (v_mfma_f32_32x32x2_f32), exact-f32 accumulate.
 *
 *   recmv_gemm_nt :  C[M,N] = act( alpha * (A[M,K] . B[N,K]^T) + bias[N] ) * out_scale
 *   recmv_gemm_tn :  C[M,N] = A[K,M]^T . B[K,N]      (weight gradients: reduction over points)
 *
 * lda/ldb/ldc are row strides in elements.  act: RECMV_ACT_*.  `bias` may be NULL.
 * For softplus the reference's nn.Softplus(beta=100) semantics are used (threshold 20).
 * recmv_gemm_tn needs a workspace of recmv_gemm_tn_workspace_bytes() for its split-K partials.
 * ---------------------------------------------------------------------------------------------- */
#define RECMV_ACT_NONE     0
#define RECMV_ACT_RELU     1
#define RECMV_ACT_SOFTPLUS 2   /* softplus(beta), param = beta */
#define RECMV_ACT_TANH     3

int recmv_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                  float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                  int act, float act_param, float out_scale, void* stream);
/* C[M,N] = ( G (.) act'(z) * g_scale ) . B^T with act'(z) taken through y = act(z) = y_scale * Y[m,k]: the
 * activation-gradient step of a layer's backward fused into the A-operand staging of the dX product.
 * ldg may be 0 (one cotangent row shared by all points). */
int recmv_gemm_nt_actgrad(const float* G, int64_t ldg, const float* Y, int64_t ldy, const float* B, int64_t ldb,
                          float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, float act_param,
                          float y_scale, float g_scale, void* stream);
/* C[M,N] = (A . B^T) (.) act'(y_scale * Y) * scale with Y [M,N] (row stride ldy): the activation-gradient step of the
 * NEXT layer of a backward chain applied in the epilogue of the product that creates its cotangent — each element
 * once (recmv_gemm_nt_actgrad, the operand-side fusion, recomputes it in every column tile). */
int recmv_gemm_nt_mulgrad(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                          int64_t N, int64_t K, const float* Y, int64_t ldy, int act, float act_param, float y_scale,
                          float scale, void* stream);
/* recmv_gemm_nt / recmv_gemm_nt_mulgrad over a row block that holds the rows of TWO nets of one shape (the two garments' SDF nets
 * in the surface root finder, utils/FindSurfacePs.py:273-353: one launch per layer for both): rows [0, split_row) multiply by
 * B (+ bias), rows [split_row, M) by B2 (+ bias2).  split_row must be a multiple of 128; B2 == NULL is the plain product. */
int recmv_gemm_nt_seg(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, const float* B2,
                      const float* bias2, int64_t split_row, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                      float act_param, float out_scale, void* stream);
int recmv_gemm_nt_mulgrad_seg(const float* A, int64_t lda, const float* B, const float* B2, int64_t split_row, int64_t ldb,
                              float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const float* Y, int64_t ldy, int act,
                              float act_param, float y_scale, float scale, void* stream);
/* Matrix mode of recmv_gemm_nt (and everything built on it): 0 = f32-input MFMA, the default — bit-for-bit an f32
 * fma chain; 1 = "bf16x6": every f32 operand element is split into three bf16 pieces in registers and each tile step
 * issues the six bf16 MFMA products of weight >= 2^-18 with f32 accumulation (f32-level accuracy, up to 2.7x the f32
 * matrix rate).  Returns the previous mode. */
int recmv_set_gemm_mode(int mode);
int recmv_get_gemm_mode(void);
/* Mode 1 only: which kernel families compute in bf16x6 — bit 0 the 128 x 128 NT tiles, bit 1 the 64 x 64 / 64 x 32 NT tiles, bit 2 the
 * TN (dW) tiles; the others stay on the exact-f32 kernels.  7 (default) = all.  A bisect / A-B switch; returns the previous mask. (ABI v7) */
int recmv_set_b3_families(int mask);
/* How the jet pass (recmv_mlp_jet_forward) zeroes the code / padding columns of its tangent rows: 1 (default) = a kernel over those
 * columns only, 0 = hipMemsetAsync over the whole tangent block (rounds 1-4; the A/B of tools/loop_repro_inproc.py).  Same bits either
 * way.  Process-global; RECMV_JET_FILL_KERNEL=0 sets 0 at load.  Returns the previous value.  (ABI v7) */
int recmv_set_jet_fill(int use_kernel);
/* bf16x6 mode, weights split ONCE: recmv_b3_split writes the three bf16 planes [3][N][Kp] (Kp = K rounded up to 32, zero-padded) of
 * a weight matrix B [N][K] (row stride ldb, K % 8 == 0, 16-byte aligned rows) into `planes` (recmv_b3_planes_bytes(N, K) bytes,
 * owned by the caller) and remembers them under B's address; the large bf16x6 products (recmv_gemm_nt and everything built on it,
 * with this B, K and ldb, N rows or fewer) then read the pieces instead of splitting B's tile in every row tile — same pieces, same
 * products, same order: bit-identical results.  The caller keeps B's contents unchanged while the entry exists and calls
 * recmv_b3_forget(B) before B or `planes` is freed.  (The weights of the reference are nn.Linear parameters under weight_norm,
 * model/network.py:49-86: re-normalised once per pass, read by every product of the pass.) */
int64_t recmv_b3_planes_bytes(int64_t N, int64_t K);
int recmv_b3_split(const float* B, int64_t ldb, int64_t N, int64_t K, void* planes, int64_t planes_bytes, void* stream);
int recmv_b3_forget(const float* B);
int64_t recmv_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K);
int recmv_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb,
                  float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* Positional encoding (model/Embedder.py:4-65): out[p, 0:3] = x, then for each of L frequency bands
 * 2^i: w[2i]*sin(2^i x), w[2i+1]*cos(2^i x) (3 values each).  out row stride ldo >= 3+6L; columns
 * [3+6L, ldo_fill) are zero-filled (ldo_fill <= ldo) so padded K reads as zeros; everything is
 * multiplied by out_scale (1/sqrt(2) for the skip connection, model/network.py:105-106).
 * weights: HOST pointer to 2L floats (utils/utils.py:40-46 produces python floats), or NULL = ones. */
int recmv_posenc_forward(const float* x, int64_t ldx, float* out, int64_t ldo, int64_t ldo_fill,
                         int64_t P, int L, const float* weights_host, float out_scale, void* stream);

/* Fused element-wise steps between the MFMA layers (csrc/elementwise.hip):
 *   recmv_act_grad  : out = gy * act'(z) written through y = act(z)     (backward of a fused layer)
 *   recmv_act_grad2 : out = a * b * d(act')/dy                           (its double backward)
 *   recmv_weight_norm_forward/backward : W = g * v/||v|| per output row (nn.utils.weight_norm dim 0,
 *                     model/network.py:82-85, model/RenderNet.py:47-50); norms [rows] is saved for backward. */
int recmv_act_grad(const float* gy, const float* y, float* out, int64_t n, int act, float act_param, void* stream);
int recmv_act_grad2(const float* a, const float* b, const float* y, float* out, int64_t n, int act, float act_param,
                    void* stream);
int recmv_weight_norm_forward(const float* v, const float* g, float* W, float* norms, int64_t rows, int64_t cols,
                              void* stream);
int recmv_weight_norm_backward(const float* v, const float* g, const float* norms, const float* gW, float* gv,
                               float* gg, int64_t rows, int64_t cols, void* stream);

/* Derivatives of the positional encoding (so that autograd of any order stays one launch):
 *   recmv_posenc_vjp : t == NULL -> out[P,3] = J(x)^T g               (g: [P, 3+6L], row stride ldg)
 *                      t != NULL -> out[P,3] = t * d/dx <g, J(x) 1>   (second-derivative term, see posenc_grad.hip)
 *   recmv_posenc_jvp : out[P, 3+6L] = J(x) t                          (t: [P,3])                                   */
int recmv_posenc_vjp(const float* x, int64_t ldx, const float* g, int64_t ldg, const float* t, int64_t ldt,
                     float* out, int64_t P, int L, const float* weights_host, void* stream);
int recmv_posenc_jvp(const float* x, int64_t ldx, const float* t, int64_t ldt, float* out, int64_t ldo, int64_t P,
                     int L, const float* weights_host, void* stream);

/* ------------------------------------------------------------------------------------------------
 * B2. SMPL kinematic chain of LBSkinner (model/Deformer.py:372-405; posedSkeleton :311-334), one kernel.
 *   poses [B,24,3] axis-angle -> G [B,24,4,4] (global joint transforms, `results`) and, when init_pose
 *   [24,4,4] and A are given, A = G . init_pose.  Js_host [24,3] and parents_host [24] are HOST arrays
 *   (constants of the skinner).  Backward: (gG, gA; either may be NULL) -> gposes [B,24,3].
 * ---------------------------------------------------------------------------------------------- */
int recmv_kinematic_chain_forward(const float* poses, const float* Js_host, const int32_t* parents_host,
                                  const float* init_pose, float* G, float* A, int64_t B, void* stream);
int recmv_kinematic_chain_backward(const float* poses, const float* Js_host, const int32_t* parents_host,
                                   const float* init_pose, const float* gG, const float* gA, float* gposes,
                                   int64_t B, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-MLP launch chains (csrc/mlp_chain.hip): ONE call enqueues every kernel of a graph-free pass through
 * an MLP of the hot path — SDF net (model/network.py:98-133), offset MLP of the deformer
 * (model/Deformer.py:141-206) — on the caller's stream.  Used by the surface root finder
 * (utils/FindSurfacePs.py:273-353) and the no-grad grid queries of Seg3dLossless.
 *
 *   input   = [ gamma_L(x) | cond[cond_index[p]] ]                       (dims[0] = 3+6L+cond_dim)
 *   layer l = hidden_act(h W[l]^T + bias[l]);  W[l] is [rows[l], dims[l]] row-major
 *   layer l+1 == skip_layer: its input is [ layer l output | gamma_L(x) ] / sqrt(2)   (network.py:105-106)
 *   last layer: no activation; only its first n_out rows are evaluated; residual: out = x + out (Deformer.py:201-206)
 * Wt[l] = W[l]^T ([dims[l], rows[l]] row-major) is only read by recmv_mlp_vjp_input.
 * pe_weights: the 2L annealing weights (utils/utils.py:40-46), ones when not annealed.
 * ---------------------------------------------------------------------------------------------- */
#define RECMV_MLP_MAX_LAYERS 12
typedef struct recmv_mlp {
  int32_t n_layers, multires, cond_dim, skip_layer /* -1: none */, hidden_act, residual;
  float act_param;
  int32_t dims[RECMV_MLP_MAX_LAYERS + 1];
  int32_t rows[RECMV_MLP_MAX_LAYERS];
  const float* W[RECMV_MLP_MAX_LAYERS];
  const float* Wt[RECMV_MLP_MAX_LAYERS];
  const float* bias[RECMV_MLP_MAX_LAYERS];
  float pe_weights[32];
  /* optional second net of the same shape (ABI v5): rows [split_row, P) of a call use W2 / Wt2 / bias2 — the rows of two garments'
   * SDF nets evaluated as one block (recmv_gemm_nt_seg).  split_row = 0 or W2[0] == NULL: one net.  A multiple of 128. */
  const float* W2[RECMV_MLP_MAX_LAYERS];
  const float* Wt2[RECMV_MLP_MAX_LAYERS];
  const float* bias2[RECMV_MLP_MAX_LAYERS];
  int64_t split_row;
} recmv_mlp;

/* keep = 1 lays every layer's activation out separately (needed by recmv_mlp_vjp_input); keep = 0 ping-pongs. */
int64_t recmv_mlp_workspace_bytes(const recmv_mlp* m, int64_t P, int keep);
/* x [P,3]; cond [*, cond_dim] with row stride ld_cond, cond_index [P] (NULL: row 0 for every point); out [P,n_out]. */
int recmv_mlp_forward(const recmv_mlp* m, const float* x, const float* cond, int64_t ld_cond,
                      const int64_t* cond_index, int64_t P, int n_out, float* out, int64_t ldo,
                      void* workspace, int64_t workspace_bytes, int keep, void* stream);
/* gx [P,3] = J(x)^T g_out after recmv_mlp_forward(keep=1) on the same workspace.  g_out [P,n_out] (row stride
 * ldg), or NULL = ones with n_out == 1 (gradient of the scalar output). */
int recmv_mlp_vjp_input(const recmv_mlp* m, const float* x, int64_t P, int n_out, const float* g_out, int64_t ldg,
                        float* gx, void* workspace, int64_t workspace_bytes, void* stream);

/* Row-tile-persistent form of the two passes above for the few-thousand-row launches of the ray path (csrc/mlp_rows.hip): ONE
 * launch per pass — a workgroup keeps the activations of 16 rays in LDS across all layers (positional encoding, per-frame code
 * gather, skip concatenation, bias, activation, residual; in the reverse pass the activation gradients and the encoding's VJP),
 * the weights stream from L2 in MFMA-fragment order.  Replaces, like the chains, model/network.py:98-133 (SDF value + input
 * gradient) and model/Deformer.py:141-206 (offset MLP + VJP) as they are called by utils/FindSurfacePs.py:273-353.
 *   recmv_mlp_rows_supported : 1 when the descriptor fits (2..12 layers, widths <= 512, <= 8 encoding bands, one net).
 *   recmv_mlp_pack(_bytes)   : lays every layer's weight (and its transpose) out in fragment order, zero-padded — once per
 *                              weight version; the packed buffer is what the two passes read.
 *   recmv_mlp_rows_forward   : as recmv_mlp_forward with n_out <= 16; keep = 1 stores the hidden activations in `workspace`
 *                              (recmv_mlp_rows_workspace_bytes) for the reverse pass.
 *   recmv_mlp_rows_vjp_input : as recmv_mlp_vjp_input, after recmv_mlp_rows_forward(keep = 1) on the same workspace.
 * Same arithmetic as the per-layer kernels (exact f32 MFMA products, f32 accumulation) in another summation order: results agree
 * to rounding.  Rows are independent of the tile they sit in.
 *   recmv_set_mlp_rows_tile  : rows per workgroup of the two passes — 1: 16 rows (one round of workgroups up to 4 096 rows: lowest
 *                              latency), 2: 32 rows (every weight byte from L2 feeds twice the FLOP, half the CUs per pass),
 *                              0 (default): 16 up to 4 096 rows, 32 above.  Process-global, like recmv_set_gemm_mode. */
int recmv_mlp_rows_supported(const recmv_mlp* m);
int recmv_set_mlp_rows_tile(int row_tiles);
int64_t recmv_mlp_pack_bytes(const recmv_mlp* m);
int recmv_mlp_pack(const recmv_mlp* m, void* packed, int64_t packed_bytes, void* stream);
int64_t recmv_mlp_rows_workspace_bytes(const recmv_mlp* m, int64_t P);
int recmv_mlp_rows_forward(const recmv_mlp* m, const void* packed, const float* x, const float* cond, int64_t ld_cond,
                           const int64_t* cond_index, int64_t P, int n_out, float* out, int64_t ldo, void* workspace,
                           int64_t workspace_bytes, int keep, void* stream);
int recmv_mlp_rows_vjp_input(const recmv_mlp* m, const void* packed, const float* x, int64_t P, int n_out, const float* g_out,
                             int64_t ldg, float* gx, const void* workspace, int64_t workspace_bytes, void* stream);

/* Strided element-wise helpers of the chains:
 *   recmv_act_grad_2d   : out[r,c] = out_scale * gy[r,c] * act'(z),  y = act(z) = y_scale * ybuf[r,c]; ldg may be 0
 *   recmv_add_scaled_2d : out[r,c] = a[r,c] + s * b[r,c]                                                       */
int recmv_act_grad_2d(const float* gy, int64_t ldg, const float* y, int64_t ldy, float* out, int64_t ldo,
                      int64_t rows, int64_t cols, int act, float act_param, float y_scale, float out_scale,
                      void* stream);
int recmv_add_scaled_2d(const float* a, int64_t lda, const float* b, int64_t ldb, float s, float* out, int64_t ldo,
                        int64_t rows, int64_t cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * First-hit mesh rasteriser (csrc/rasterize_meshes.hip).
 *   replaces `self.maskRender(def_garment_mesh)` -> Fragments in find_surface_ps
 *   (engineer/networks/OptimGarmentNetwork.py:742-767; settings :2336-2347) = pytorch3d 0.4.0
 *   `rasterize_meshes(faces_per_pixel=1, perspective_correct, clip_barycentric_coords=False)`, whose outputs feed
 *   utils/FindSurfacePs.py:7-37.
 * face_verts [total_faces,3,3] f32: NDC x, y (+x left, +y up; pixel column c is centred at x = 1-(2c+1)/W, row r at
 * y = 1-(2r+1)/H — model/CameraMine.py:132-142) and view-space depth z of the three corners of every face, meshes
 * packed one after another; mesh n owns faces [mesh_first_face[n], +mesh_num_faces[n]) (device int64 [N]);
 * max_faces_per_mesh >= max(mesh_num_faces) (host value, sizes the launch).  blur_radius is pytorch3d's (squared NDC
 * distance; 0 = hard coverage).  Outputs per pixel [N,H,W]: pix_to_face (packed face index, -1 = none), zbuf,
 * bary_coords [N,H,W,3], dists (signed squared distance to the nearest edge, negative inside); -1 where empty.
 * Ties in depth go to the lowest face index; the result does not depend on scheduling.
 * ---------------------------------------------------------------------------------------------- */
int64_t recmv_rasterize_meshes_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t total_faces);
int recmv_rasterize_meshes(const float* face_verts, const int64_t* mesh_first_face, const int64_t* mesh_num_faces,
                           int64_t N, int64_t total_faces, int64_t max_faces_per_mesh, int64_t H, int64_t W,
                           float blur_radius, int perspective_correct, int cull_backfaces, int64_t* pix_to_face,
                           float* zbuf, float* bary_coords, float* dists, void* workspace, int64_t workspace_bytes,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * Point-cloud rasteriser + alpha compositor, forward and backward (csrc/rasterize_points.hip).
 *   replaces `self.pcRender(Pointclouds(...))` of the mask loss (engineer/networks/OptimGarmentNetwork.py:937;
 *   PointsRendererWithFrags(_Split) model/CameraMine.py:306-415; settings engineer/networks/OptimNetwork.py:87-100:
 *   points_per_pixel = 50, radius 0.006 / 0.00465 / 0.0041) = pytorch3d 0.4.0 `rasterize_points` +
 *   `alpha_composite` and their backward passes.
 * points [total_points,3] f32: NDC x, y (same pixel convention as recmv_rasterize_meshes) and view depth z, clouds
 * packed one after another; cloud n owns points [cloud_first_point[n], +cloud_num_points[n]) (device int64 [N]).
 * Fragments, all [N,H,W,K] with K = points_per_pixel, K fastest: idx (packed point index, int32, -1 = none), zbuf,
 * dists (squared NDC distance of the point to the pixel centre; a point is listed when it is < radius^2 and z >= 0),
 * the K nearest in depth, sorted by (depth, index), packed to the front.
 *   recmv_rasterize_points_backward : grad_points [total_points,3] = d/dpoints of (grad_dists . dists + grad_zbuf . zbuf)
 *                                     (grad_zbuf may be NULL).  One thread per point sums its pixels in row-major order:
 *                                     deterministic (upstream accumulates with float atomics).
 *   recmv_alpha_composite_forward   : images [N,C,H,W]; images[n,c,y,x] = sum_k a_k prod_{l<k}(1 - a_l) features[c,idx_k]
 *                                     with alphas [N,H,W,K] and features [C,total_points].  radius2 == 0: `alphas`
 *                                     are the opacities.  radius2 != 0: `alphas` holds the rasteriser's dists and
 *                                     a = 1 - dists / radius2 (PointsRendererWithFrags' 1 - d2/radius^2,
 *                                     model/CameraMine.py:361-362, fused).  idx lists must be packed to the front.
 *   recmv_alpha_composite_backward  : grad_alphas [N,H,W,K] = gradient w.r.t. the `alphas` INPUT (opacities or
 *                                     dists), no atomics; grad_features [C,total_points] or NULL.
 * ---------------------------------------------------------------------------------------------- */
int64_t recmv_rasterize_points_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t total_points, float radius);
int recmv_rasterize_points(const float* points, const int64_t* cloud_first_point, const int64_t* cloud_num_points,
                           int64_t N, int64_t total_points, int64_t max_points_per_cloud, int64_t H, int64_t W,
                           float radius, int points_per_pixel, int32_t* idx, float* zbuf, float* dists,
                           void* workspace, int64_t workspace_bytes, void* stream);
int recmv_rasterize_points_backward(const float* points, const int64_t* cloud_first_point,
                                    const int64_t* cloud_num_points, const int32_t* idx, const float* grad_dists,
                                    const float* grad_zbuf, int64_t N, int64_t total_points,
                                    int64_t max_points_per_cloud, int64_t H, int64_t W, float radius,
                                    int points_per_pixel, float* grad_points, void* stream);
int recmv_alpha_composite_forward(const int32_t* idx, const float* alphas, const float* features, int64_t N,
                                  int64_t H, int64_t W, int points_per_pixel, int64_t C, int64_t total_points,
                                  float radius2, float* images, void* stream);
int recmv_alpha_composite_backward(const int32_t* idx, const float* alphas, const float* features,
                                   const float* grad_images, int64_t N, int64_t H, int64_t W, int points_per_pixel,
                                   int64_t C, int64_t total_points, float radius2, float* grad_alphas,
                                   float* grad_features, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused linear-blend skinning on ray points and the root finder's per-ray step (csrc/lbs_fused.hip).
 *   replaces LBSkinner.forward with batch_inds (model/Deformer.py:405-445) and its input gradient, and the
 *   energy / update arithmetic of OptimizeGarmentSurfacePs (utils/FindSurfacePs.py:316-351), no autograd.
 * grid: the skinning-weight volume, CHANNELS-LAST [D,H,W,24] f32 (the memory of a torch channels_last_3d
 * [1,24,D,H,W] tensor), sampled trilinearly at (p - center) * scale (scale = 2 / bbox extent) with the
 * GridSamplerMine semantics (border padding, align_corners=False).
 *   recmv_lbs_forward   : d = (sum_j w_j A[frame,j]) [p;1] + trans[frame];  A [B,24,4,4], trans [B,3].
 *                         With rays [P,3] and cam [3] (device): loss2 = |(d-c) x v| / |d-c|, angle (degrees) and
 *                         g_d = d loss2 / d d.  rays == NULL skips those outputs.
 *   recmv_lbs_vjp_input : g_p [P,3] = J_d(p)^T g_d.
 *   recmv_rootfind_update: unfinished &= !(|f| < dthreshold && angle < athreshold); on unfinished rays (and
 *                         do_update != 0) p -= E grad / |grad|^2 with E = w1|f| + w2 loss2,
 *                         grad = w1 sign(f) gf + w2 gd; *counter += number of unfinished rays.
 * ---------------------------------------------------------------------------------------------- */
typedef struct recmv_lbs_grid {
  const float* volume;
  int64_t D, H, W;
  float center[3];
  float scale[3];
} recmv_lbs_grid;

int recmv_lbs_forward(const float* ps, const int64_t* frame, int64_t P, const float* A, const float* trans,
                      int64_t B, const recmv_lbs_grid* grid, const float* cam, const float* rays, float* d,
                      float* loss2, float* angle, float* g_d, void* stream);
int recmv_lbs_vjp_input(const float* ps, const int64_t* frame, int64_t P, const float* A, int64_t B,
                        const recmv_lbs_grid* grid, const float* g_d, float* g_p, void* stream);
/* Parameter side of the skinning VJP, staged for two fixed-order reductions: W [P,24] (sampled blend weights),
 * Q [P, B*12] = g_d (x) [p;1] in the point's frame block (zeros elsewhere), Gs [P, B*3] = g_d likewise.  Then
 * gA[b,j,i,k] = recmv_gemm_tn(W, Q)[j, b*12 + 4i + k] and gtrans[b,i] = recmv_colsum(Gs)[b*3 + i]. */
int recmv_lbs_vjp_params_stage(const float* ps, const int64_t* frame, int64_t P, int64_t B,
                               const recmv_lbs_grid* grid, const float* g_d, float* W, float* Q, float* Gs,
                               void* stream);
/* The skinning stage as a jet (ABI v9): v [P,3] as recmv_lbs_forward and J [P,9] = d v / d p (row i = gradient of v_i) in one
 * launch — replaces LBSkinner.forward + the three create_graph autograd.grad calls of utils/utils.py:133-156 (compute_Jacobian) at
 * the converged ray points — and its first-order reverse: g_p [P,3] per point, the parameter side staged as W4 [4P,24],
 * Q4 [4P, B*12], Gs [P, B*3] for gA[b,j,i,k] = recmv_gemm_tn(W4, Q4)[j, b*12 + 4i + k], gtrans = recmv_colsum(Gs).  gv / gJ: the
 * cotangents of v and J, either may be NULL (zeros). */
int recmv_lbs_jet_forward(const float* ps, const int64_t* frame, int64_t P, const float* A, const float* trans,
                          int64_t B, const recmv_lbs_grid* grid, float* v, float* J, void* stream);
int recmv_lbs_jet_backward_stage(const float* ps, const int64_t* frame, int64_t P, const float* A, int64_t B,
                                 const recmv_lbs_grid* grid, const float* gv, const float* gJ, float* g_p, float* W4,
                                 float* Q4, float* Gs, void* stream);
int recmv_rootfind_update(float* p, const float* f, const float* gf, const float* loss2, const float* angle,
                          const float* gd, uint8_t* unfinished, int32_t* counter, int64_t P, float dthreshold,
                          float athreshold, float w1, float w2, int do_update, void* stream);
/* The same step with the step index on the device, so that all steps of an iteration are one and the same launch (a
 * step captured in a hipGraph can be replayed).  counters / marks: int32 [times + 2], zero-initialised by the caller;
 * state: int32 [1] = index of the step to run (0 at the start).  Runs the update with do_update = (step < times),
 * counters[step] += unfinished rays; then marks[step] = counters[step] + 1 and state[0] = step + 1.  A host that polls an
 * asynchronous copy of `marks` reads 0 for "not run yet" and 1 for "no unfinished ray" (utils/FindSurfacePs.py:311-313). */
int recmv_rootfind_step(float* p, const float* f, const float* gf, const float* loss2, const float* angle,
                        const float* gd, uint8_t* unfinished, int32_t* counters, int32_t* marks, int32_t* state,
                        int64_t P, float dthreshold, float athreshold, float w1, float w2, int times, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of one fused layer y = act(x W^T + b) in one call (csrc/linear_bwd.hip) — autograd of nn.Linear +
 * activation in the reference (model/network.py:98-111 etc.).  y, gy [M,N]; x [M,K]; Wt = W^T [K,N];
 * outputs gx [M,K] = (gy . act') W, gW [N,K] = (gy . act')^T x, gb [N] = column sums; any output may be NULL.
 * recmv_colsum: out[c] = sum_r g[r*ld + c], fixed summation order (deterministic).
 * ---------------------------------------------------------------------------------------------- */
int64_t recmv_colsum_workspace_bytes(int64_t rows, int64_t cols);
int recmv_colsum(const float* g, int64_t ld, int64_t rows, int64_t cols, float* out, void* workspace,
                 int64_t workspace_bytes, void* stream);
int64_t recmv_linear_backward_workspace_bytes(int64_t M, int64_t N, int64_t K);
int recmv_linear_backward(const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                          const float* Wt, int64_t ldwt, int64_t M, int64_t N, int64_t K, int act, float act_param,
                          float* gx, int64_t ldgx, float* gW, float* gb, void* workspace, int64_t workspace_bytes,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * MLP "jet" pass (csrc/mlp_jet.hip): value and input-Jacobian in one forward sweep (three tangent rows per point
 * carried through the layers), with an explicit first-order reverse sweep.  Replaces autograd's double backward for
 * the terms of the loss that differentiate grad_x SDF (model/network.py:121-133, OptimGarmentNetwork.py:1108-1119,
 * :1169-1172) and the Jacobian of the offset MLP (utils/utils.py:133-156, OptimGarmentNetwork.py:1135-1155).
 *   forward : y [P, rows_last] (residual nets: x + mlp), tang [3P, n_j] with tang[k*P + p, j] = d y_j / d x_k of the
 *             MLP part (the residual's identity is NOT included).
 *   backward: cotangents gy [P, rows_last] / gtang [3P, n_j] (NULL = zeros) -> gW[l] [rows[l], dims[l]], gb[l],
 *             g_in [4P, pad4(dims[0])] (cotangent of the stacked layer-0 input; rows [0,P) x columns [3+6L, dims[0])
 *             are the per-point cotangents of the gathered code) and gx [P,3]; each may be NULL.
 * eye3: 9 floats on the device = the 3x3 identity.  The workspace carries the activations from forward to backward.
 * recmv_gather_rows: out[r, 0:cols] = table[index[r] (or 0), 0:cols], columns [cols, fill) zeroed.
 * ---------------------------------------------------------------------------------------------- */
int64_t recmv_mlp_jet_workspace_bytes(const recmv_mlp* m, int64_t P);
int recmv_mlp_jet_forward(const recmv_mlp* m, const float* x, const float* cond, int64_t ld_cond,
                          const int64_t* cond_index, const float* eye3, int64_t P, int n_j, float* y, int64_t ldy,
                          float* tang, void* workspace, int64_t workspace_bytes, void* stream);
int recmv_mlp_jet_backward(const recmv_mlp* m, const float* x, const float* eye3, int64_t P, int n_j, const float* gy,
                           int64_t ldgy, const float* gtang, float* const* gW, float* const* gb, float* g_in,
                           float* gx, void* workspace, int64_t workspace_bytes, void* stream);
int recmv_gather_rows(const float* table, int64_t ldt, const int64_t* index, float* out, int64_t ldo, int64_t rows,
                      int64_t cols, int64_t fill, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-launch HIP-event timing of the MFMA kernels, for the bench's roofline object.  Between
 * recmv_profile_begin() and recmv_profile_end() every gemm_nt / gemm_tn launch (from Python or from inside the
 * launch chains) is bracketed by events recorded on its launch stream.  recmv_profile_end fills, per kernel variant
 * v (0..7: gemm_nt_kernel<T, FAST, AMUL>, v = (T-1) + 2*FAST + 4*AMUL; 8: gemm_tn_occ_kernel / gemm_tn_kernel, the product
 * alone without its split-K reduction pass; 9 / 10: gemm_nt_occ_kernel<false, ...> with 128x128 / 64x128 tiles; 11:
 * gemm_nt_occ_kernel<true, ...>;
 * 12 / 13: gemm_nt_narrow_kernel<true, false, .> / its other instantiations — a caller that passes fewer than 14 slots gets
 * 9..13 folded into 3 / 3 / 7 / 2 / 6, the slots that carried those launches before):
 * out[5v] = timed launches, out[5v+1] = their summed duration [s], out[5v+2] = their summed algorithmic FLOP
 * (2 M N K), out[5v+3] / out[5v+4] = launches / FLOP of the launches smaller than `min_flops`, which are counted
 * but not bracketed (out must hold 5 * n_variants doubles, n_variants >= 9).
 * ---------------------------------------------------------------------------------------------- */
int recmv_profile_begin(double min_flops);
int recmv_profile_end(double* out, int n_variants);
/* Algorithmic bytes, 4 (M K + N K + M N), of the launches the last recmv_profile_end bracketed, per variant (same slots as its `out`:
 * out[v]).  With their summed duration this is the kernel's HBM-side roofline beside its MFMA one.  (ABI v7) */
int recmv_profile_bytes(double* out, int n_variants);
/* The launches of >= 4 GFLOP among the bracketed ones: out[3v] launches, out[3v+1] seconds, out[3v+2] FLOP.  (ABI v7) */
int recmv_profile_large(double* out, int n_variants);
/* After recmv_profile_end: out2[0] = seconds in which at least ONE bracketed launch was running (union of the event intervals of all
 * streams on one time axis), out2[1] = seconds from the first bracketed start to the last bracketed end. */
int recmv_profile_busy(double* out2);

#ifdef __cplusplus
}
#endif
#endif /* RECMV_HIP_H_ */
