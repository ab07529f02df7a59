/*
 * recmv_oracle.c — CPU ORACLE for the REC-MV hot-path kernels.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may build, load or call it.  Nothing under rec-mv_amd/ (the product) imports or links it; the product
 * path fails loudly when librecmv_hip.so is missing and has no CPU fallback.
 *
 * It restates, in plain C, the algorithms of the reference's CUDA extensions, function by function:
 *   oracle_inv3x3_*      <- FastMinv/Matrix3x3InvKernels.cu:22-104
 *   oracle_gs3d_*        <- MCAcc/cuda/GridSamplerMineKernel.cu:162-328, 333-570, 575-914
 *   oracle_interp2x_*    <- MCAcc/cuda/interp2x_boundary3d_kernel.cu:11-239
 *   oracle_mc_*          <- MCGpu/CudaKernels.cu:304-521 (tables :4-298, re-encoded in mc_tables.inc)
 *
 * Why a restatement and not the reference itself: the reference's extensions are CUDA-only (every binding asserts a
 * CUDA tensor: FastMinv/M3x3Inv.cpp:4-6, MCGpu/MCGpu.cpp:3-5, interp2x_boundary3d.cpp:12-14), their host side
 * launches with `<<<...>>>` and this environment has no nvcc.  Where the KERNEL BODIES are plain C they are compiled
 * for the host as they lie in the reference tree and used to pin this file: oracle/_ref (oracle/Makefile `ref`,
 * oracle/ref/*.cpp) holds MCGpu/CudaKernels.cu:4-521 and FastMinv/Matrix3x3InvKernels.cu:18-104 run serially through
 * a CUDA-spelling shim; tests/test_oracle_ref.py asserts bit equality with oracle_mc / oracle_inv3x3_*.
 *
 * Pinning (tests/test_oracle_pins.py, tests/test_oracle_ref.py):
 *   inv3x3   — PINNED TO THE REFERENCE KERNELS (oracle/_ref, f32 and f64, forward + backward, bit-exact, incl. det=0
 *              and |det| straddling 1e-4); also inv*m == I on randn(10000,3,3) as FastMinv/check.py:7-20 does;
 *              torch.linalg.inv (f64); backward vs autograd of torch.linalg.inv.
 *   sampler  — equality with torch.nn.functional.grid_sample(bilinear, border, align_corners=False) and its
 *              autograd on the reference's own check shapes (MCAcc/check_grid_sampler_mine.py:5-9);
 *              gradcheck of the backward Function in f64 (ibid. :11-16) -> pins the double backward.
 *   interp2x — F.interpolate(trilinear, align_corners=True) + (0<valid<1), MCAcc/seg3d_lossless.py:273-282.
 *   MC       — PINNED TO THE REFERENCE KERNELS (oracle/_ref): the reference's d_mc_get_mesh_on_gpu /
 *              d_conver_ijkd_to_pindex / d_scale_vertices run on 13 volumes (white noise with shifted iso values,
 *              sphere, surface touching the box, exact ties with the iso value) x 2 spacings; their output,
 *              canonicalised by SURVEY.md §8a-E keys read from the reference's own edge->vertex table, equals
 *              oracle_mc bit for bit (vertex f32 bits, face ids incl. -1, winding, face order), also when the
 *              reference's ids are handed out in a scrambled order.  Plus the invariants (closed 2-manifold, Euler
 *              characteristic 2 on a sphere) and the table SHA-256 equal to the one generated from the reference.
 *
 * Floating point: compiled with -ffp-contract=off; the places where the reference's nvcc build contracts
 * a*b+c into an fma that matters for last-bit equality are written with explicit fma()/fmaf().
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define T float
#define SUF f32
#define FMA fmaf
#include "oracle_impl.inc"
#undef T
#undef SUF
#undef FMA

#define T double
#define SUF f64
#define FMA fma
#include "oracle_impl.inc"
#undef T
#undef SUF
#undef FMA

/* ---------------------------------------------------------------------------------------------
 * Marching cubes — follows MCGpu/CudaKernels.cu.
 * ------------------------------------------------------------------------------------------- */
#include "mc_tables.inc"

/* cube corner offsets (a2fVertexOffset :4-8), edge -> corner pair (a2iEdgeConnection :9-14) and edge
 * directions (a2fEdgeDirection :15-20) of the reference. */
static const float kVertexOffset[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0},
                                          {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
static const int kEdgeConnection[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                           {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
static const float kEdgeDirection[12][3] = {{1, 0, 0}, {0, 1, 0}, {-1, 0, 0}, {0, -1, 0}, {1, 0, 0}, {0, 1, 0},
                                            {-1, 0, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}};
/* triangle corner (edge id) -> (dx,dy,dz,direction) of the lattice edge that carries the vertex
 * (the if/else ladder at :385-456) */
static const int kEdgeIJKD[12][4] = {{0, 0, 0, 0}, {1, 0, 0, 1}, {0, 1, 0, 0}, {0, 0, 0, 1},
                                     {0, 0, 1, 0}, {1, 0, 1, 1}, {0, 1, 1, 0}, {0, 0, 1, 1},
                                     {0, 0, 0, 2}, {1, 0, 0, 2}, {1, 1, 0, 2}, {0, 1, 0, 2}};

static float mc_get_offset(float v1, float v2, float desired) { /* d_fGetOffset :304-313 */
  double delta = v2 - v1; /* float subtraction widened to double */
  if (delta == 0.0) return 0.5f;
  return (float)((desired - v1) / delta);
}

static int mc_tri_entry(int flag_index, int n) { /* a2iTriangleConnectionTable[flag_index*16+n] */
  int v = (int)((kMcTriTable[flag_index] >> (4 * n)) & 0xF);
  return v == 0xF ? -1 : v;
}

static int mc_cube(const float* sdf, int64_t NY, int64_t NZ, int64_t i, int64_t j, int64_t k, float iso,
                   float cube[8]) {
  int flag = 0;
  for (int v = 0; v < 8; ++v) {
    int64_t idx = (i + (int)kVertexOffset[v][0]) * NY * NZ + (j + (int)kVertexOffset[v][1]) * NZ +
                  (k + (int)kVertexOffset[v][2]);
    cube[v] = sdf[idx];
    if (cube[v] < iso) flag |= 1 << v; /* strict <, :346 */
  }
  return flag;
}

static int mc_edge_flags(int flag) { /* == aiCubeEdgeFlags[flag] (checked by tools/gen_mc_tables.py) */
  int f = 0;
  for (int e = 0; e < 12; ++e)
    if (((flag >> kEdgeConnection[e][0]) & 1) != ((flag >> kEdgeConnection[e][1]) & 1)) f |= 1 << e;
  return f;
}

/* Canonical deterministic ordering: vertices by edge key ((x*NY+y)*NZ+z)*3+dir,
 * faces by (voxel linear index, triangle number).  The reference's own order is atomic-nondeterministic.
 * Pass `vertices`/`faces` NULL to only count.  edge_state scratch is allocated here (3*N ints). */
int oracle_mc(const float* sdf, int64_t NX, int64_t NY, int64_t NZ, float iso, float xstep, float ystep,
              float zstep, float xmin, float ymin, float zmin, float* vertices, int64_t* faces,
              int64_t* counts /* [2] out: V, F */) {
  if (NX <= 0 || NY <= 0 || NZ <= 0) return -1;
  const int64_t total = NX * NY * NZ;
  int32_t* edge_state = NULL;
  if (faces) {
    edge_state = (int32_t*)malloc(sizeof(int32_t) * 3 * total);
    if (!edge_state) return -2;
    for (int64_t t = 0; t < 3 * total; ++t) edge_state[t] = -1; /* d_set_int(-1) :506-512 */
  }
  int64_t nv = 0, nf = 0;
  /* pass 1: vertices, created by the voxel that owns the edge (its local edges 0,3,8) :457-466 */
  for (int64_t index = 0; index < total; ++index) {
    int64_t i = index / (NY * NZ), j = (index - i * NY * NZ) / NZ, k = index - i * NY * NZ - j * NZ;
    if (!(i < NX - 1 && j < NY - 1 && k < NZ - 1)) continue;
    float cube[8];
    int flag = mc_cube(sdf, NY, NZ, i, j, k, iso, cube);
    int eflags = mc_edge_flags(flag);
    if (!eflags) continue;
    static const int owned[3] = {0, 3, 8}; /* -> direction 0,1,2 */
    float fX = (float)i, fY = (float)j, fZ = (float)k;
    for (int o = 0; o < 3; ++o) {
      int e = owned[o];
      if (!(eflags & (1 << e))) continue;
      if (vertices) {
        int c0 = kEdgeConnection[e][0], c1 = kEdgeConnection[e][1];
        float off = mc_get_offset(cube[c0], cube[c1], iso);
        float px = fX + (kVertexOffset[c0][0] + off * kEdgeDirection[e][0]); /* :361-366 */
        float py = fY + (kVertexOffset[c0][1] + off * kEdgeDirection[e][1]);
        float pz = fZ + (kVertexOffset[c0][2] + off * kEdgeDirection[e][2]);
        /* d_scale_vertices :513-521 — nvcc contracts v*step+min into one fma */
        vertices[3 * nv + 0] = fmaf(px, xstep, xmin);
        vertices[3 * nv + 1] = fmaf(py, ystep, ymin);
        vertices[3 * nv + 2] = fmaf(pz, zstep, zmin);
      }
      if (edge_state) edge_state[index * 3 + o] = (int32_t)nv;
      ++nv;
    }
  }
  /* pass 2: faces :372-384, 467-470 and d_conver_ijkd_to_pindex :492-505 (corner order reversed) */
  for (int64_t index = 0; index < total; ++index) {
    int64_t i = index / (NY * NZ), j = (index - i * NY * NZ) / NZ, k = index - i * NY * NZ - j * NZ;
    if (!(i < NX - 1 && j < NY - 1 && k < NZ - 1)) continue;
    float cube[8];
    int flag = mc_cube(sdf, NY, NZ, i, j, k, iso, cube);
    if (!mc_edge_flags(flag)) continue;
    for (int tri = 0; tri < 5; ++tri) {
      if (mc_tri_entry(flag, 3 * tri) < 0) break;
      if (faces) {
        for (int corner = 0; corner < 3; ++corner) {
          int e = mc_tri_entry(flag, 3 * tri + corner);
          int64_t bx = i + kEdgeIJKD[e][0], by = j + kEdgeIJKD[e][1], bz = k + kEdgeIJKD[e][2];
          faces[nf * 3 + (2 - corner)] = (int64_t)edge_state[bx * NY * NZ * 3 + by * NZ * 3 + bz * 3 + kEdgeIJKD[e][3]];
        }
      }
      ++nf;
    }
  }
  free(edge_state);
  counts[0] = nv;
  counts[1] = nf;
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Mesh rasteriser, one face per pixel — the fragments `maskRender` gives `utils.FindSurfacePs`
 * (engineer/networks/OptimGarmentNetwork.py:742-767, utils/FindSurfacePs.py:7-37).
 *
 * PARITY UNPINNED against the real thing: the reference calls pytorch3d 0.4.0
 * (`rasterize_meshes`, settings at OptimGarmentNetwork.py:2336-2347: faces_per_pixel=1, blur_radius=0,
 * perspective_correct=True, clip_barycentric_coords=False), which is not vendored in /root/reference and not
 * installed here.  This restates its published per-pixel algorithm (RasterizeMeshesNaive: every pixel loops over
 * the faces of its mesh in order and keeps the nearest; geometry helpers of csrc/utils/geometry_utils.h) with the
 * reference's pixel convention (model/CameraMine.py:132-142: pixel i is centred at NDC 1-(2i+1)/S).
 * One deliberate deviation: a candidate whose interpolated depth is NaN is dropped (upstream keeps it).
 * ------------------------------------------------------------------------------------------- */
#define RAST_EPS 1e-8f

static float rast_edge(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

static float rast_point_line(float px, float py, float ax, float ay, float bx, float by) {
  float bax = bx - ax, bay = by - ay;
  float l2 = bax * bax + bay * bay;
  if (l2 <= RAST_EPS) return (px - bx) * (px - bx) + (py - by) * (py - by);
  float t = (bax * (px - ax) + bay * (py - ay)) / l2;
  float tt = fminf(fmaxf(t, 0.f), 1.f);
  float qx = ax + tt * bax, qy = ay + tt * bay;
  return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

static float rast_min3(float a, float b, float c) { return fminf(fminf(a, b), c); }
static float rast_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

/* One (pixel centre, face) test of CheckPixelInsideFace; returns 0 when the face does not cover the pixel. */
static int rast_pair(const float* v, float px, float py, float blur_radius, int perspective_correct,
                     int cull_backfaces, float* out /* z, dist, b0, b1, b2 */) {
  const float br = sqrtf(blur_radius);
  const float x0 = v[0], y0 = v[1], z0 = v[2], x1 = v[3], y1 = v[4], z1 = v[5], x2 = v[6], y2 = v[7], z2 = v[8];
  const float xmin = rast_min3(x0, x1, x2) - br, xmax = rast_max3(x0, x1, x2) + br;
  const float ymin = rast_min3(y0, y1, y2) - br, ymax = rast_max3(y0, y1, y2) + br;
  const float zmax = rast_max3(z0, z1, z2);
  const int outside = (px < xmin || px > xmax || py < ymin || py > ymax);
  const float face_area = rast_edge(x0, y0, x1, y1, x2, y2);
  const int back_face = face_area < 0.f;
  const int zero_area = (face_area <= RAST_EPS && face_area >= -RAST_EPS);
  if (zmax < 0.f || (cull_backfaces && back_face) || outside || zero_area || face_area != face_area) return 0;
  const float area = rast_edge(x2, y2, x0, y0, x1, y1) + RAST_EPS;
  const float w0 = rast_edge(px, py, x1, y1, x2, y2) / area;
  const float w1 = rast_edge(px, py, x2, y2, x0, y0) / area;
  const float w2 = rast_edge(px, py, x0, y0, x1, y1) / area;
  float b0 = w0, b1 = w1, b2 = w2;
  if (perspective_correct) {
    const float t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
    const float denom = fmaxf(t0 + t1 + t2, RAST_EPS);
    b0 = t0 / denom;
    b1 = t1 / denom;
    b2 = t2 / denom;
  }
  const float pz = b0 * z0 + b1 * z1 + b2 * z2;
  if (pz < 0.f || pz != pz) return 0;
  const float e01 = rast_point_line(px, py, x0, y0, x1, y1);
  const float e02 = rast_point_line(px, py, x0, y0, x2, y2);
  const float e12 = rast_point_line(px, py, x1, y1, x2, y2);
  const float dist = fminf(fminf(e01, e02), e12);
  const int inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
  if (!inside && dist >= blur_radius) return 0;
  out[0] = pz + 0.f;
  out[1] = inside ? -dist : dist;
  out[2] = b0;
  out[3] = b1;
  out[4] = b2;
  return 1;
}

/* The checker: every pixel loops over every face of its mesh, in order (RasterizeMeshesNaive). */
int oracle_rasterize_meshes(const float* face_verts, const int64_t* mesh_first_face, const int64_t* mesh_num_faces,
                            int64_t N, int64_t H, int64_t W, float blur_radius, int perspective_correct,
                            int cull_backfaces, int64_t* pix_to_face, float* zbuf, float* bary, float* dists) {
  if (N < 0 || H <= 0 || W <= 0) return -1;
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < N * H * W; ++i) {
    const int64_t n = i / (H * W), pix = i % (H * W);
    /* upstream flips both axes: output pixel (r, c) looks through NDC (1-(2c+1)/W, 1-(2r+1)/H) */
    const int64_t r = pix / W, c = pix % W;
    const float px = 1.f - (2.f * (float)c + 1.f) / (float)W;
    const float py = 1.f - (2.f * (float)r + 1.f) / (float)H;
    int64_t best = -1;
    float q[5] = {-1.f, -1.f, -1.f, -1.f, -1.f}, cand[5];
    for (int64_t f = mesh_first_face[n]; f < mesh_first_face[n] + mesh_num_faces[n]; ++f) {
      if (!rast_pair(face_verts + 9 * f, px, py, blur_radius, perspective_correct, cull_backfaces, cand)) continue;
      /* K = 1 queue: the first candidate enters, a later one replaces it only when strictly nearer */
      if (best < 0 || cand[0] < q[0]) {
        best = f;
        memcpy(q, cand, sizeof(q));
      }
    }
    pix_to_face[i] = best;
    zbuf[i] = q[0];
    dists[i] = q[1];
    bary[3 * i] = q[2];
    bary[3 * i + 1] = q[3];
    bary[3 * i + 2] = q[4];
  }
  return 0;
}

/* The CPU port timed as `cpu_baseline` (bench.py): same pair test, but organised the way a CPU scan converter is —
 * every face visits only the pixel centres of its bounding box; threads own bands of rows, so no locking and the same
 * answer as the per-pixel loop (faces are visited in index order inside a band). */
int oracle_rasterize_meshes_scan(const float* face_verts, const int64_t* mesh_first_face,
                                 const int64_t* mesh_num_faces, int64_t N, int64_t H, int64_t W, float blur_radius,
                                 int perspective_correct, int cull_backfaces, int64_t* pix_to_face, float* zbuf,
                                 float* bary, float* dists) {
  if (N < 0 || H <= 0 || W <= 0) return -1;
  const float br = sqrtf(blur_radius);
  const int64_t band = 8, nbands = (H + band - 1) / band;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t bi = 0; bi < nbands; ++bi) {
      const int64_t rlo = bi * band, rhi = (rlo + band < H ? rlo + band : H) - 1;
      for (int64_t i = (n * H + rlo) * W; i < (n * H + rhi + 1) * W; ++i) {
        pix_to_face[i] = -1;
        zbuf[i] = dists[i] = -1.f;
        bary[3 * i] = bary[3 * i + 1] = bary[3 * i + 2] = -1.f;
      }
      for (int64_t f = mesh_first_face[n]; f < mesh_first_face[n] + mesh_num_faces[n]; ++f) {
        const float* v = face_verts + 9 * f;
        const float ylo = rast_min3(v[1], v[4], v[7]) - br, yhi = rast_max3(v[1], v[4], v[7]) + br;
        const float xlo = rast_min3(v[0], v[3], v[6]) - br, xhi = rast_max3(v[0], v[3], v[6]) + br;
        if (!(ylo == ylo && yhi == yhi && xlo == xlo && xhi == xhi)) continue;
        /* centre(i) = 1 - (2i+1)/S in [lo, hi]; widened by one pixel, the pair test decides */
        double r0d = floor(((double)H * (1.0 - yhi) - 1.0) * 0.5) - 1.0, r1d = ceil(((double)H * (1.0 - ylo) - 1.0) * 0.5) + 1.0;
        double c0d = floor(((double)W * (1.0 - xhi) - 1.0) * 0.5) - 1.0, c1d = ceil(((double)W * (1.0 - xlo) - 1.0) * 0.5) + 1.0;
        int64_t r0 = r0d < (double)rlo ? rlo : (r0d > (double)H ? H : (int64_t)r0d);
        int64_t r1 = r1d > (double)rhi ? rhi : (r1d < -1.0 ? -1 : (int64_t)r1d);
        int64_t c0 = c0d < 0.0 ? 0 : (c0d > (double)W ? W : (int64_t)c0d);
        int64_t c1 = c1d > (double)(W - 1) ? W - 1 : (c1d < -1.0 ? -1 : (int64_t)c1d);
        for (int64_t r = r0; r <= r1; ++r)
          for (int64_t c = c0; c <= c1; ++c) {
            const float px = 1.f - (2.f * (float)c + 1.f) / (float)W;
            const float py = 1.f - (2.f * (float)r + 1.f) / (float)H;
            float cand[5];
            if (!rast_pair(v, px, py, blur_radius, perspective_correct, cull_backfaces, cand)) continue;
            const int64_t i = (n * H + r) * W + c;
            if (pix_to_face[i] < 0 || cand[0] < zbuf[i]) {
              pix_to_face[i] = f;
              zbuf[i] = cand[0];
              dists[i] = cand[1];
              bary[3 * i] = cand[2];
              bary[3 * i + 1] = cand[3];
              bary[3 * i + 2] = cand[4];
            }
          }
      }
    }
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Point-cloud rasteriser + alpha compositor — the silhouette renderer of the mask loss
 * (engineer/networks/OptimGarmentNetwork.py:937, model/CameraMine.py:306-415, engineer/networks/OptimNetwork.py:87-100).
 *
 * PARITY UNPINNED against pytorch3d 0.4.0 itself (not vendored, not installed): restated from its published
 * algorithm — RasterizePointsNaive (every pixel loops over the points of its cloud in order, keeps the K nearest
 * in depth among those closer than `radius` to the pixel centre, sorted by depth), its backward
 * (d dist2 / d p = 2 (p - pixel)), and alpha_composite forward / backward.  Ties in depth are ordered by point index.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  float z;
  int64_t p;
} rast_pt;

static int rast_pt_less(float za, int64_t pa, float zb, int64_t pb) { return za < zb || (za == zb && pa < pb); }

/* insert (z,p) into the sorted list q[0..*n) of capacity K (keeps the K smallest) */
static void rast_pt_insert(rast_pt* q, int* n, int K, float z, int64_t p) {
  int pos = *n;
  if (*n == K) {
    if (!rast_pt_less(z, p, q[K - 1].z, q[K - 1].p)) return;
    pos = K - 1;
  } else {
    ++*n;
  }
  while (pos > 0 && rast_pt_less(z, p, q[pos - 1].z, q[pos - 1].p)) {
    q[pos] = q[pos - 1];
    --pos;
  }
  q[pos].z = z;
  q[pos].p = p;
}

static void rast_pt_write(const float* points, const rast_pt* q, int n, int K, float xf, float yf, int32_t* idx,
                          float* zbuf, float* dists) {
  for (int k = 0; k < K; ++k) {
    if (k < n) {
      const float dx = xf - points[3 * q[k].p], dy = yf - points[3 * q[k].p + 1];
      idx[k] = (int32_t)q[k].p;
      zbuf[k] = q[k].z;
      dists[k] = dx * dx + dy * dy;
    } else {
      idx[k] = -1;
      zbuf[k] = dists[k] = -1.f;
    }
  }
}

int oracle_rasterize_points(const float* points, const int64_t* cloud_first, const int64_t* cloud_num, int64_t N,
                            int64_t H, int64_t W, float radius, int K, int32_t* idx, float* zbuf, float* dists) {
  if (N < 0 || H <= 0 || W <= 0 || K <= 0 || K > 512) return -1;
  const float radius2 = radius * radius;
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < N * H * W; ++i) {
    const int64_t n = i / (H * W), pix = i % (H * W);
    const float xf = 1.f - (2.f * (float)(pix % W) + 1.f) / (float)W;
    const float yf = 1.f - (2.f * (float)(pix / W) + 1.f) / (float)H;
    rast_pt q[512];
    int nq = 0;
    for (int64_t p = cloud_first[n]; p < cloud_first[n] + cloud_num[n]; ++p) {
      const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
      if (!(pz >= 0.f)) continue;
      const float dx = xf - px, dy = yf - py;
      const float dist2 = dx * dx + dy * dy;
      if (dist2 < radius2) rast_pt_insert(q, &nq, K, pz + 0.f, p);
    }
    rast_pt_write(points, q, nq, K, xf, yf, idx + i * K, zbuf + i * K, dists + i * K);
  }
  return 0;
}

/* CPU port for `cpu_baseline`: points visit the pixel centres of their own disc; rows are banded over threads. */
int oracle_rasterize_points_scan(const float* points, const int64_t* cloud_first, const int64_t* cloud_num,
                                 int64_t N, int64_t H, int64_t W, float radius, int K, int32_t* idx, float* zbuf,
                                 float* dists) {
  if (N < 0 || H <= 0 || W <= 0 || K <= 0 || K > 512) return -1;
  const float radius2 = radius * radius;
  const int64_t band = 8, nbands = (H + band - 1) / band;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t bi = 0; bi < nbands; ++bi) {
      const int64_t rlo = bi * band, rhi = (rlo + band < H ? rlo + band : H) - 1;
      const int64_t rows = rhi - rlo + 1;
      rast_pt* q = (rast_pt*)malloc(sizeof(rast_pt) * (size_t)(rows * W * K));
      int* nq = (int*)calloc((size_t)(rows * W), sizeof(int));
      for (int64_t p = cloud_first[n]; p < cloud_first[n] + cloud_num[n]; ++p) {
        const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
        if (!(pz >= 0.f) || px != px || py != py) continue;
        double r0d = floor(((double)H * (1.0 - ((double)py + radius)) - 1.0) * 0.5) - 1.0;
        double r1d = ceil(((double)H * (1.0 - ((double)py - radius)) - 1.0) * 0.5) + 1.0;
        double c0d = floor(((double)W * (1.0 - ((double)px + radius)) - 1.0) * 0.5) - 1.0;
        double c1d = ceil(((double)W * (1.0 - ((double)px - radius)) - 1.0) * 0.5) + 1.0;
        int64_t r0 = r0d < (double)rlo ? rlo : (r0d > (double)H ? H : (int64_t)r0d);
        int64_t r1 = r1d > (double)rhi ? rhi : (r1d < -1.0 ? -1 : (int64_t)r1d);
        int64_t c0 = c0d < 0.0 ? 0 : (c0d > (double)W ? W : (int64_t)c0d);
        int64_t c1 = c1d > (double)(W - 1) ? W - 1 : (c1d < -1.0 ? -1 : (int64_t)c1d);
        for (int64_t r = r0; r <= r1; ++r)
          for (int64_t c = c0; c <= c1; ++c) {
            const float xf = 1.f - (2.f * (float)c + 1.f) / (float)W;
            const float yf = 1.f - (2.f * (float)r + 1.f) / (float)H;
            const float dx = xf - px, dy = yf - py;
            const float dist2 = dx * dx + dy * dy;
            if (dist2 < radius2) {
              const int64_t l = (r - rlo) * W + c;
              rast_pt_insert(q + l * K, nq + l, K, pz + 0.f, p);
            }
          }
      }
      for (int64_t r = rlo; r <= rhi; ++r)
        for (int64_t c = 0; c < W; ++c) {
          const int64_t l = (r - rlo) * W + c, i = (n * H + r) * W + c;
          const float xf = 1.f - (2.f * (float)c + 1.f) / (float)W;
          const float yf = 1.f - (2.f * (float)r + 1.f) / (float)H;
          rast_pt_write(points, q + l * K, nq[l], K, xf, yf, idx + i * K, zbuf + i * K, dists + i * K);
        }
      free(q);
      free(nq);
    }
  }
  return 0;
}

int oracle_rasterize_points_backward(const float* points, const int32_t* idx, const float* grad_dists,
                                     const float* grad_zbuf, int64_t N, int64_t total_points, int64_t H, int64_t W,
                                     int K, float* grad_points) {
  memset(grad_points, 0, sizeof(float) * 3 * (size_t)total_points);
  for (int64_t i = 0; i < N * H * W; ++i) {
    const int64_t pix = i % (H * W);
    const float xf = 1.f - (2.f * (float)(pix % W) + 1.f) / (float)W;
    const float yf = 1.f - (2.f * (float)(pix / W) + 1.f) / (float)H;
    for (int k = 0; k < K; ++k) {
      const int64_t p = idx[i * K + k];
      if (p < 0) continue;
      const float gd = grad_dists ? grad_dists[i * K + k] : 0.f;
      grad_points[3 * p] += 2.f * gd * (points[3 * p] - xf);
      grad_points[3 * p + 1] += 2.f * gd * (points[3 * p + 1] - yf);
      if (grad_zbuf) grad_points[3 * p + 2] += grad_zbuf[i * K + k];
    }
  }
  return 0;
}

/* alphas / idx [N,H,W,K]; features [C,P]; images [N,C,H,W] */
int oracle_alpha_composite_forward(const int32_t* idx, const float* alphas, const float* features, int64_t N,
                                   int64_t H, int64_t W, int K, int64_t C, int64_t P, float* images) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < N * H * W; ++i) {
    const int64_t n = i / (H * W), pix = i % (H * W);
    for (int64_t c = 0; c < C; ++c) {
      float cum_alpha = 1.f, result = 0.f;
      for (int k = 0; k < K; ++k) {
        const int64_t p = idx[i * K + k];
        if (p < 0) continue;
        const float alpha = alphas[i * K + k];
        result += cum_alpha * alpha * features[c * P + p];
        cum_alpha = cum_alpha * (1.f - alpha);
      }
      images[(n * C + c) * H * W + pix] = result;
    }
  }
  return 0;
}

int oracle_alpha_composite_backward(const int32_t* idx, const float* alphas, const float* features,
                                    const float* grad_images, int64_t N, int64_t H, int64_t W, int K, int64_t C,
                                    int64_t P, float* grad_alphas, float* grad_features) {
  if (grad_features) memset(grad_features, 0, sizeof(float) * (size_t)(C * P));
  for (int64_t i = 0; i < N * H * W; ++i) {
    const int64_t n = i / (H * W), pix = i % (H * W);
    for (int k = 0; k < K; ++k) grad_alphas[i * K + k] = 0.f;
    for (int64_t c = 0; c < C; ++c) {
      const float g = grad_images[(n * C + c) * H * W + pix];
      float cum_alpha = 1.f;
      for (int k = 0; k < K; ++k) {
        const int64_t p = idx[i * K + k];
        if (p < 0) continue;
        const float alpha = alphas[i * K + k];
        const float f = features[c * P + p];
        if (grad_features) grad_features[c * P + p] += cum_alpha * alpha * g;
        grad_alphas[i * K + k] += cum_alpha * f * g;
        for (int t = 0; t < k; ++t) {
          if (idx[i * K + t] < 0) continue;
          const float alpha_t = alphas[i * K + t];
          grad_alphas[i * K + t] += -g * f * cum_alpha * alpha / (1.f - alpha_t + 1e-9f); /* upstream kEpsilon */
        }
        cum_alpha = cum_alpha * (1.f - alpha);
      }
    }
  }
  return 0;
}
