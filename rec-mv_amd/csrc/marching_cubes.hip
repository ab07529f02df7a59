// Deterministic marching cubes on an x-major f32 volume — gfx950.
//
// Semantics follow MCGpu/CudaKernels.cu of the reference: cube index with strict `v < iso` (:346),
// interpolation d_fGetOffset with a double division and 0.5 on a zero delta (:304-313), vertices created
// only by the voxel that owns the lattice edge (its local edges 0,3,8 = +x,+y,+z) (:457-466), triangle
// corners resolved through the edge -> vertex-id map and written reversed as int64 (:492-505), vertices
// scaled v*step+min (:513-521).  Edges whose owner voxel lies outside the grid resolve to -1, as there.
//
// Design (not the reference's).  The reference fills a 3*NX*NY*NZ int edge-state array with -1 on every
// call (204 MB at 257^3), hands out vertex/face ids with global atomics (nondeterministic order), stages
// faces as (i,j,k,d) tuples and needs a second resolve kernel.  Here:
//   K1 classify : one wave64 per SEGMENT = 64 consecutive k of one (i,j) lattice row (k is the fastest
//                 index of the volume, so every corner load is a 256-byte coalesced read and the k+1
//                 corners come from a lane shuffle).  Wave ballots give three 64-bit "owned edge" masks
//                 (x,y,z edge cut & owner valid) per segment and the segment's vertex/triangle counts:
//                 32 bytes per 64 voxels instead of 768 bytes of edge state.
//   K2 scan     : exclusive scan of the per-segment counts in segment order == canonical key order
//                 ((x*NY+y)*NZ+z)*3+dir for vertices and (voxel, triangle#) for faces.  No inter-block
//                 communication: every block re-reduces the (tiny) per-run sums that precede it.
//   K3 emit     : only segments that have work re-read their corners; a vertex id anywhere in the grid is
//                 offset[segment] + popcount(masks below the lane) — wave-level bit arithmetic, no
//                 atomics, no id map in memory; faces are written directly as int64.
// Output order is therefore a pure function of the input (needed so that frame-sharded ranks keep
// identical vertex numbering).  Blocks are remapped so each XCD sweeps a contiguous slab
// range and re-finds its (i+1) rows in its own L2.
//
// Algorithmic bytes: 4*NX*NY*NZ (volume read once) + 12*V + 24*F.
#include "common.h"
#include "mc_tables.inc"

namespace recmv {
namespace {

constexpr int kBlk = 256;            // 4 waves
constexpr int kWavesPerBlk = kBlk / kWave;
constexpr int kRun = 16;             // segments per K1 block (4 per wave) -> one per-run sum
constexpr int kScanBlk = 1024;
constexpr int kScanChunk = 4096;     // segments scanned per K2 block

struct McLayout {
  int64_t nseg;       // NX*NY*S
  int64_t nrun;       // ceil(nseg / kRun)
  int S;              // segments per lattice row = ceil((NZ-1)/64)
  // byte offsets into the workspace
  int64_t off_masks;  // uint64 [3][nseg]
  int64_t off_counts; // uint2  [nseg]   (nvert, ntri)
  int64_t off_offs;   // uint2  [nseg]   exclusive prefix
  int64_t off_runsum; // uint2  [nrun]
  int64_t off_runact; // uint32 [nrun]   segments with work in the run
  int64_t off_active; // int32  [nseg]   compacted list of the segments with work, in segment order
  int64_t off_total;  // uint32 [3]      {vertices, faces, active segments}
  int64_t bytes;
};

inline McLayout make_layout(int64_t nx, int64_t ny, int64_t nz) {
  McLayout L;
  L.S = (int)ceil_div(nz > 1 ? nz - 1 : 0, kWave);
  L.nseg = nx * ny * L.S;
  L.nrun = ceil_div(L.nseg, kRun);
  int64_t o = 0;
  auto take = [&](int64_t bytes) {
    int64_t r = o;
    o += (bytes + 255) / 256 * 256;
    return r;
  };
  L.off_masks = take(3 * L.nseg * 8);
  L.off_counts = take(L.nseg * 8);
  L.off_offs = take(L.nseg * 8);
  L.off_runsum = take(L.nrun * 8);
  L.off_runact = take(L.nrun * 4);
  L.off_active = take(L.nseg * 4);
  L.off_total = take(16);
  L.bytes = o;
  return L;
}

// XCD-aware block remap: workgroup b runs on XCD b%8 (observed, speed only); give XCD x the contiguous
// range of logical blocks [x*per, (x+1)*per) so neighbouring slabs share an L2.
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t nb) {
  const int64_t per = nb / kNumXCD;
  const int64_t main = per * kNumXCD;
  if (b >= main) return b;  // tail blocks keep their id
  return (b % kNumXCD) * per + b / kNumXCD;
}

#pragma clang fp contract(off)

__device__ __forceinline__ float mc_offset(float v1, float v2, float iso) {
  const double delta = (double)(v2 - v1);  // float subtraction widened (:306)
  if (delta == 0.0) return 0.5f;
  return (float)((double)(iso - v1) / delta);
}

// Corner values of the 64 voxels of one segment.  Lane l <-> k = k0 + l.
struct Corners {
  float v[8];
  bool valid;  // voxel (i,j,k) is inside the grid of cubes
};

__device__ __forceinline__ Corners load_corners(const float* __restrict__ sdf, int NX, int NY, int NZ, int i,
                                                int j, int k0, int lane) {
  Corners c;
  const int k = k0 + lane;
  const bool row_ok = (i < NX - 1) && (j < NY - 1);
  c.valid = row_ok && (k < NZ - 1);
  const int64_t r00 = ((int64_t)i * NY + j) * NZ;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (row_ok && k < NZ) {
    a0 = sdf[r00 + k];
    a1 = sdf[r00 + (int64_t)NY * NZ + k];
    a2 = sdf[r00 + (int64_t)NY * NZ + NZ + k];
    a3 = sdf[r00 + NZ + k];
  }
  // k+1 corners come from the neighbouring lane; lane 63 fetches its own.
  float b0 = __shfl_down(a0, 1), b1 = __shfl_down(a1, 1), b2 = __shfl_down(a2, 1), b3 = __shfl_down(a3, 1);
  if (lane == kWave - 1 && row_ok && k + 1 < NZ) {
    b0 = sdf[r00 + k + 1];
    b1 = sdf[r00 + (int64_t)NY * NZ + k + 1];
    b2 = sdf[r00 + (int64_t)NY * NZ + NZ + k + 1];
    b3 = sdf[r00 + NZ + k + 1];
  }
  c.v[0] = a0; c.v[1] = a1; c.v[2] = a2; c.v[3] = a3;
  c.v[4] = b0; c.v[5] = b1; c.v[6] = b2; c.v[7] = b3;
  return c;
}

__device__ __forceinline__ int cube_index(const Corners& c, float iso) {
  int f = 0;
#pragma unroll
  for (int v = 0; v < 8; ++v) f |= (c.v[v] < iso) ? (1 << v) : 0;
  return f;
}

__device__ __forceinline__ int tri_count(unsigned long long word) {
  // number of non-0xF nibbles / 3; entries are packed from nibble 0 upwards
  const unsigned long long hi = word & (word >> 1) & (word >> 2) & (word >> 3) & 0x1111111111111111ull;
  return (16 - __popcll(hi)) / 3;
}

// ------------------------------------------------------------------------------------------- K1
__global__ __launch_bounds__(kBlk) void mc_classify_kernel(const float* __restrict__ sdf, int NX, int NY,
                                                           int NZ, float iso, int S, int64_t nseg,
                                                           int64_t nrun, unsigned long long* __restrict__ masks,
                                                           uint2* __restrict__ counts,
                                                           uint2* __restrict__ runsum,
                                                           unsigned int* __restrict__ runact) {
  __shared__ unsigned long long tri_lds[256];
  __shared__ unsigned int blk_v[kWavesPerBlk], blk_t[kWavesPerBlk], blk_a[kWavesPerBlk];
  tri_lds[threadIdx.x] = kMcTriTable[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  for (int64_t pb = blockIdx.x; pb < nrun; pb += gridDim.x) {
    const int64_t run = xcd_remap(pb, nrun);
    unsigned int sum_v = 0, sum_t = 0, sum_a = 0;
    // the run's 4 segments of this wave are loaded back to back (16 coalesced row loads in flight per wave) before
    // any of them is classified: the kernel is a pure stream over the volume and lives on memory-level parallelism
    Corners cs[kRun / kWavesPerBlk];
#pragma unroll
    for (int it = 0; it < kRun / kWavesPerBlk; ++it) {
      const int64_t seg = run * kRun + it * kWavesPerBlk + wave;
      const int64_t segc = seg < nseg ? seg : nseg - 1;
      const int64_t row = segc / S;
      cs[it] = load_corners(sdf, NX, NY, NZ, (int)(row / NY), (int)(row % NY), (int)(segc % S) * kWave, lane);
    }
#pragma unroll
    for (int it = 0; it < kRun / kWavesPerBlk; ++it) {
      const int64_t seg = run * kRun + it * kWavesPerBlk + wave;
      if (seg >= nseg) continue;
      const Corners& c = cs[it];
      const int flag = c.valid ? cube_index(c, iso) : 0;
      const bool in0 = flag & 1;
      const bool ex = c.valid && (in0 != (bool)(flag & 2));    // edge 0: corners 0-1 (+x)
      const bool ey = c.valid && (in0 != (bool)(flag & 8));    // edge 3: corners 3-0 (+y)
      const bool ez = c.valid && (in0 != (bool)(flag & 16));   // edge 8: corners 0-4 (+z)
      const int nt = (flag != 0 && flag != 255) ? tri_count(tri_lds[flag]) : 0;
      const unsigned long long mx = __ballot(ex), my = __ballot(ey), mz = __ballot(ez);
      const unsigned int nv = __popcll(mx) + __popcll(my) + __popcll(mz);
      const unsigned int ntw = __popcll(__ballot(nt & 1)) + 2 * __popcll(__ballot(nt & 2)) +
                               4 * __popcll(__ballot(nt & 4));
      if (lane == 0) {
        masks[seg] = mx;
        masks[nseg + seg] = my;
        masks[2 * nseg + seg] = mz;
        counts[seg] = make_uint2(nv, ntw);
      }
      sum_v += nv;
      sum_t += ntw;
      sum_a += (nv | ntw) ? 1u : 0u;
    }
    if (lane == 0) {
      blk_v[wave] = sum_v;
      blk_t[wave] = sum_t;
      blk_a[wave] = sum_a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned int a = 0, b = 0, c = 0;
#pragma unroll
      for (int w = 0; w < kWavesPerBlk; ++w) {
        a += blk_v[w];
        b += blk_t[w];
        c += blk_a[w];
      }
      runsum[run] = make_uint2(a, b);
      runact[run] = c;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------- K2
// Block q scans segments [q*kScanChunk, (q+1)*kScanChunk).  Its base = sum of runsum[0 .. q*kScanChunk/kRun).
__global__ __launch_bounds__(kScanBlk) void mc_scan_kernel(const uint2* __restrict__ counts,
                                                           const uint2* __restrict__ runsum,
                                                           const unsigned int* __restrict__ runact, int64_t nseg,
                                                           int64_t nrun, uint2* __restrict__ offs,
                                                           int* __restrict__ active, unsigned int* __restrict__ total) {
  __shared__ unsigned int sh_v[kScanBlk / kWave], sh_t[kScanBlk / kWave], sh_a[kScanBlk / kWave];
  __shared__ unsigned int base_v, base_t, base_a;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int64_t seg0 = (int64_t)blockIdx.x * kScanChunk;
  const int64_t runs_before = seg0 / kRun;  // kScanChunk % kRun == 0
  // (1) base offset: reduce the run sums before this chunk
  unsigned int av = 0, at = 0, aa = 0;
  for (int64_t r = tid; r < runs_before; r += kScanBlk) {
    const uint2 x = runsum[r];
    av += x.x;
    at += x.y;
    aa += runact[r];
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    av += __shfl_xor(av, o);
    at += __shfl_xor(at, o);
    aa += __shfl_xor(aa, o);
  }
  if (lane == 0) {
    sh_v[wave] = av;
    sh_t[wave] = at;
    sh_a[wave] = aa;
  }
  __syncthreads();
  if (tid == 0) {
    unsigned int a = 0, b = 0, c = 0;
    for (int w = 0; w < kScanBlk / kWave; ++w) {
      a += sh_v[w];
      b += sh_t[w];
      c += sh_a[w];
    }
    base_v = a;
    base_t = b;
    base_a = c;
  }
  __syncthreads();
  // (2) scan the chunk: each thread owns 4 consecutive segments
  constexpr int kPer = kScanChunk / kScanBlk;
  uint2 loc[kPer];
  unsigned int tv = 0, tt = 0, ta = 0;
#pragma unroll
  for (int e = 0; e < kPer; ++e) {
    const int64_t seg = seg0 + (int64_t)tid * kPer + e;
    loc[e] = seg < nseg ? counts[seg] : make_uint2(0, 0);
    tv += loc[e].x;
    tt += loc[e].y;
    ta += (loc[e].x | loc[e].y) ? 1u : 0u;
  }
  // inclusive wave scan of (tv, tt, ta)
  unsigned int iv = tv, itt = tt, ia = ta;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const unsigned int pv = __shfl_up(iv, o), pt = __shfl_up(itt, o), pa = __shfl_up(ia, o);
    if (lane >= o) {
      iv += pv;
      itt += pt;
      ia += pa;
    }
  }
  __syncthreads();  // base_* consumed below; sh_* reused
  if (lane == kWave - 1) {
    sh_v[wave] = iv;
    sh_t[wave] = itt;
    sh_a[wave] = ia;
  }
  __syncthreads();
  unsigned int wv = 0, wt = 0, wa = 0;
  for (int w = 0; w < wave; ++w) {
    wv += sh_v[w];
    wt += sh_t[w];
    wa += sh_a[w];
  }
  // exclusive prefixes of this thread
  unsigned int ev = base_v + wv + iv - tv, et = base_t + wt + itt - tt, ea = base_a + wa + ia - ta;
#pragma unroll
  for (int e = 0; e < kPer; ++e) {
    const int64_t seg = seg0 + (int64_t)tid * kPer + e;
    if (seg < nseg) {
      offs[seg] = make_uint2(ev, et);
      if (loc[e].x | loc[e].y) active[ea++] = (int)seg;     // compacted, in segment order
    }
    ev += loc[e].x;
    et += loc[e].y;
  }
  // (3) grand totals from the last block
  if (blockIdx.x == gridDim.x - 1 && tid == kScanBlk - 1) {
    total[0] = ev;
    total[1] = et;
    total[2] = ea;
  }
}

// ------------------------------------------------------------------------------------------- K3
struct SegRec {
  unsigned long long mx, my, mz;
  unsigned int voff;
};

__device__ __forceinline__ SegRec load_rec(const unsigned long long* __restrict__ masks,
                                           const uint2* __restrict__ offs, int64_t nseg, int64_t seg,
                                           bool ok) {
  SegRec r;
  r.mx = r.my = r.mz = 0ull;
  r.voff = 0;
  if (ok) {
    r.mx = masks[seg];
    r.my = masks[nseg + seg];
    r.mz = masks[2 * nseg + seg];
    r.voff = offs[seg].x;
  }
  return r;
}

// vertex id of the edge (lane l, direction dir) inside a segment record, or -1 if no vertex lives there
__device__ __forceinline__ long long vertex_id(const SegRec& r, int l, int dir) {
  const unsigned long long bit = 1ull << l;
  const unsigned long long m = dir == 0 ? r.mx : (dir == 1 ? r.my : r.mz);
  if (!(m & bit)) return -1;
  const unsigned long long lt = bit - 1ull;
  unsigned int rank = __popcll(r.mx & lt) + __popcll(r.my & lt) + __popcll(r.mz & lt);
  if (dir > 0) rank += (r.mx & bit) ? 1 : 0;
  if (dir > 1) rank += (r.my & bit) ? 1 : 0;
  return (long long)(r.voff + rank);
}

__global__ __launch_bounds__(kBlk) void mc_emit_kernel(const float* __restrict__ sdf, int NX, int NY, int NZ,
                                                       float iso, int S, int64_t nseg,
                                                       const unsigned long long* __restrict__ masks,
                                                       const uint2* __restrict__ counts,
                                                       const uint2* __restrict__ offs, float xstep,
                                                       float ystep, float zstep, float xmin, float ymin,
                                                       float zmin, float* __restrict__ vertices,
                                                       long long* __restrict__ faces,
                                                       const int* __restrict__ active, int64_t nactive) {
  __shared__ unsigned long long tri_lds[256];
  tri_lds[threadIdx.x] = kMcTriTable[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  // one wave per segment that has work (compacted list built by the scan): no idle waves, no serial skipping
  const int64_t nblk_seg = (nactive + kWavesPerBlk - 1) / kWavesPerBlk;
  for (int64_t pb = blockIdx.x; pb < nblk_seg; pb += gridDim.x) {
    const int64_t slot = xcd_remap(pb, nblk_seg) * kWavesPerBlk + threadIdx.x / kWave;
    if (slot >= nactive) continue;
    const int64_t seg = active[slot];
    const int s = (int)(seg % S);
    const int64_t row = seg / S;
    const int j = (int)(row % NY), i = (int)(row / NY);
    const int k = s * kWave + lane;
    // every load of this segment is issued up front (corners, counts, its own and the six neighbour records): they
    // depend only on `seg`, and a wave that waits for them one after the other is latency-bound
    const uint2 cnt = counts[seg];
    const unsigned int toff = offs[seg].y;
    const Corners c = load_corners(sdf, NX, NY, NZ, i, j, s * kWave, lane);
    const SegRec self = load_rec(masks, offs, nseg, seg, true);
    const bool i1 = i + 1 < NX, j1 = j + 1 < NY, s1 = s + 1 < S;
    const int64_t seg10 = seg + (int64_t)NY * S, seg01 = seg + S, seg11 = seg + (int64_t)NY * S + S;
    const SegRec r10 = load_rec(masks, offs, nseg, seg10, i1);
    const SegRec r01 = load_rec(masks, offs, nseg, seg01, j1);
    const SegRec r11 = load_rec(masks, offs, nseg, seg11, i1 && j1);
    // k+1 of lane 63 lives in the next segment of the same row
    const SegRec n00 = load_rec(masks, offs, nseg, seg + 1, s1);
    const SegRec n10 = load_rec(masks, offs, nseg, seg10 + 1, s1 && i1);
    const SegRec n01 = load_rec(masks, offs, nseg, seg01 + 1, s1 && j1);
    const int flag = c.valid ? cube_index(c, iso) : 0;
    const unsigned long long bit = 1ull << lane;

    // ---- vertices owned by this lattice point (canonical order: dir 0,1,2)
    if ((self.mx | self.my | self.mz) & bit) {
      const float fX = (float)i, fY = (float)j, fZ = (float)k;
      if (self.mx & bit) {  // edge 0: corner 0 -> 1, direction (+1,0,0)
        const float t = mc_offset(c.v[0], c.v[1], iso);
        const long long id = vertex_id(self, lane, 0);
        vertices[3 * id + 0] = fmaf(fX + (0.f + t * 1.f), xstep, xmin);
        vertices[3 * id + 1] = fmaf(fY + (0.f + t * 0.f), ystep, ymin);
        vertices[3 * id + 2] = fmaf(fZ + (0.f + t * 0.f), zstep, zmin);
      }
      if (self.my & bit) {  // edge 3: corner 3 -> 0, direction (0,-1,0), start offset (0,1,0)
        const float t = mc_offset(c.v[3], c.v[0], iso);
        const long long id = vertex_id(self, lane, 1);
        vertices[3 * id + 0] = fmaf(fX + (0.f + t * 0.f), xstep, xmin);
        vertices[3 * id + 1] = fmaf(fY + (1.f + t * -1.f), ystep, ymin);
        vertices[3 * id + 2] = fmaf(fZ + (0.f + t * 0.f), zstep, zmin);
      }
      if (self.mz & bit) {  // edge 8: corner 0 -> 4, direction (0,0,+1)
        const float t = mc_offset(c.v[0], c.v[4], iso);
        const long long id = vertex_id(self, lane, 2);
        vertices[3 * id + 0] = fmaf(fX + (0.f + t * 0.f), xstep, xmin);
        vertices[3 * id + 1] = fmaf(fY + (0.f + t * 0.f), ystep, ymin);
        vertices[3 * id + 2] = fmaf(fZ + (0.f + t * 1.f), zstep, zmin);
      }
    }

    // ---- faces
    if (cnt.y == 0) continue;  // wave-uniform
    const unsigned long long word = (flag != 0 && flag != 255) ? tri_lds[flag] : ~0ull;
    const int nt = tri_count(word);
    // exclusive prefix of nt across the wave
    int incl = nt;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int p = __shfl_up(incl, o);
      if (lane >= o) incl += p;
    }
    long long f = (long long)toff + (incl - nt);
    // r10/r01/r11/n00/n10/n01: records of the lattice rows a triangle corner can live on, (di,dj) in {0,1}^2,
    // segment s or s+1 (loaded above)
    if (nt > 0) {
      // edge id -> (di, dj, dk, dir)   (the if/else ladder at CudaKernels.cu:385-456)
      for (int t = 0; t < nt; ++t) {
#pragma unroll
        for (int corner = 0; corner < 3; ++corner) {
          const int e = (int)((word >> (4 * (3 * t + corner))) & 0xF);
          // packed lookup: di = bit0, dj = bit1, dk = bit2, dir = bits 3-4, per edge 5 bits
          // e:   0        1        2        3        4        5        6        7        8        9       10       11
          // di   0        1        0        0        0        1        0        0        0        1        1        0
          // dj   0        0        1        0        0        0        1        0        0        0        1        1
          // dk   0        0        0        0        1        1        1        1        0        0        0        0
          // dir  0        1        0        1        0        1        0        1        2        2        2        2
          const int di = (0x622 >> e) & 1;
          const int dj = (0xC44 >> e) & 1;
          const int dk = (0x0F0 >> e) & 1;
          const int dir = e >= 8 ? 2 : (e & 1);
          int l = lane + dk;
          const bool next = l >= kWave;
          l &= kWave - 1;
          long long id;
          if (!di && !dj) id = vertex_id(next ? n00 : self, l, dir);
          else if (di && !dj) id = vertex_id(next ? n10 : r10, l, dir);
          else if (!di && dj) id = vertex_id(next ? n01 : r01, l, dir);
          else id = vertex_id(r11, l, dir);  // edge 10 only (dk = 0)
          faces[(f + t) * 3 + (2 - corner)] = id;
        }
      }
    }
  }
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int64_t recmv_mc_workspace_bytes(int64_t nx, int64_t ny, int64_t nz) {
  if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
  return make_layout(nx, ny, nz).bytes;
}

static int mc_check(const char* who, const float* sdf, int64_t nx, int64_t ny, int64_t nz, const void* ws,
                    int64_t ws_bytes, McLayout* L) {
  RECMV_REQUIRE(nx > 0 && ny > 0 && nz > 0, "%s: empty volume (%lld,%lld,%lld)", who, (long long)nx,
                (long long)ny, (long long)nz);
  RECMV_REQUIRE(nx < (1 << 15) && ny < (1 << 15) && nz < (1 << 15) && nx * ny * nz < (1ll << 31),
                "%s: volume too large for 32-bit vertex/face counters", who);
  RECMV_REQUIRE(sdf && ws, "%s: NULL pointer", who);
  *L = make_layout(nx, ny, nz);
  if (ws_bytes < L->bytes) {
    set_error("%s: workspace %lld < %lld bytes", who, (long long)ws_bytes, (long long)L->bytes);
    return RECMV_ERR_WORKSPACE;
  }
  RECMV_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "%s: workspace must be 256-byte aligned", who);
  return RECMV_OK;
}

extern "C" int recmv_mc_count(const float* sdf, int64_t nx, int64_t ny, int64_t nz, float iso,
                              void* workspace, int64_t workspace_bytes, int32_t* counts_host,
                              void* stream) {
  McLayout L;
  int rc = mc_check("mc_count", sdf, nx, ny, nz, workspace, workspace_bytes, &L);
  if (rc) return rc;
  RECMV_REQUIRE(counts_host, "mc_count: NULL counts_host");
  hipStream_t s = (hipStream_t)stream;
  counts_host[0] = counts_host[1] = counts_host[2] = 0;
  if (L.nseg == 0) return RECMV_OK;
  char* ws = (char*)workspace;
  auto* masks = (unsigned long long*)(ws + L.off_masks);
  auto* counts = (uint2*)(ws + L.off_counts);
  auto* offs = (uint2*)(ws + L.off_offs);
  auto* runsum = (uint2*)(ws + L.off_runsum);
  auto* runact = (unsigned int*)(ws + L.off_runact);
  auto* active = (int*)(ws + L.off_active);
  auto* total = (unsigned int*)(ws + L.off_total);
  const int g1 = (int)(L.nrun < 4096 ? L.nrun : 4096);
  hipLaunchKernelGGL(mc_classify_kernel, dim3(g1), dim3(kBlk), 0, s, sdf, (int)nx, (int)ny, (int)nz, iso,
                     L.S, L.nseg, L.nrun, masks, counts, runsum, runact);
  rc = check_launch("mc_classify");
  if (rc) return rc;
  const int g2 = (int)ceil_div(L.nseg, kScanChunk);
  hipLaunchKernelGGL(mc_scan_kernel, dim3(g2), dim3(kScanBlk), 0, s, counts, runsum, runact, L.nseg, L.nrun, offs,
                     active, total);
  rc = check_launch("mc_scan");
  if (rc) return rc;
  uint32_t host[3] = {0, 0, 0};
  RECMV_HIP_TRY(hipMemcpyAsync(host, total, 12, hipMemcpyDeviceToHost, s));
  RECMV_HIP_TRY(hipStreamSynchronize(s));
  counts_host[0] = (int32_t)host[0];
  counts_host[1] = (int32_t)host[1];
  counts_host[2] = (int32_t)host[2];
  return RECMV_OK;
}

extern "C" int recmv_mc_emit(const float* sdf, int64_t nx, int64_t ny, int64_t nz, float iso, float xstep,
                             float ystep, float zstep, float xmin, float ymin, float zmin,
                             const void* workspace, int64_t workspace_bytes, int64_t n_active_segments,
                             float* vertices, int64_t* faces, void* stream) {
  McLayout L;
  int rc = mc_check("mc_emit", sdf, nx, ny, nz, workspace, workspace_bytes, &L);
  if (rc) return rc;
  if (L.nseg == 0 || n_active_segments <= 0) return RECMV_OK;
  RECMV_REQUIRE(n_active_segments <= L.nseg, "mc_emit: n_active_segments out of range");
  hipStream_t s = (hipStream_t)stream;
  const char* ws = (const char*)workspace;
  auto* masks = (const unsigned long long*)(ws + L.off_masks);
  auto* counts = (const uint2*)(ws + L.off_counts);
  auto* offs = (const uint2*)(ws + L.off_offs);
  auto* active = (const int*)(ws + L.off_active);
  const int64_t nblk_seg = ceil_div(n_active_segments, kWavesPerBlk);
  const int g3 = (int)(nblk_seg < 16384 ? nblk_seg : 16384);
  hipLaunchKernelGGL(mc_emit_kernel, dim3(g3), dim3(kBlk), 0, s, sdf, (int)nx, (int)ny, (int)nz, iso, L.S,
                     L.nseg, masks, counts, offs, xstep, ystep, zstep, xmin, ymin, zmin, vertices,
                     (long long*)faces, active, n_active_segments);
  return check_launch("mc_emit");
}
