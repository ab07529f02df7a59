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
// faces as (i,j,k,d) tuples and needs a second resolve kernel.  Here the volume is streamed ONCE, into one
// bit per lattice point, and everything after that is proportional to the surface:
//   K0 inside   : a pure stream over the flat volume: every wave instruction loads 64 consecutive floats (256
//                 coalesced bytes, many in flight per wave) and one ballot turns them into a 64-bit word of
//                 "value < iso" bits; a wave parks 64 such words in a register pair and stores them with one
//                 512-byte store.  4 bytes in -> 1 bit out; no LDS, no barrier.
//   K1 classify : one LANE per SEGMENT (64 consecutive voxels k = 64 s + l of one (i,j) voxel row).  The eight
//                 corner-bit words of the segment (4 lattice rows x {k, k+1}) are funnel-shifted out of the bit
//                 volume (2.6 MB at 257^3: L2-resident), and a segment without a sign change is decided by eight
//                 64-bit AND/ORs.  For the others the lane forms the three "owned edge" masks (x,y,z edge cut &
//                 owner valid) with XORs, walks the few cut voxels for their triangle counts, and writes a
//                 64-byte record.  Every segment leaves a packed count word (vertices, triangles, voxels with
//                 output); the workgroup also leaves the sums of its 4096 segments.
//   K2 scan     : exclusive scan of the count words in segment order == canonical key order
//                 ((x*NY+y)*NZ+z)*3+dir for vertices and (voxel, triangle#) for faces; writes the offsets
//                 into the records and expands the output masks into the list of ACTIVE VOXELS.  No
//                 inter-block waiting: a block's base is the sum of the chunk sums K1 left before its chunk.
//   K3 emit     : one LANE per active voxel (not one wave per segment: a surface crosses a 64-voxel segment
//                 in one or two places).  A vertex id anywhere in the grid is record.offset + popcount(masks
//                 below the lane) — bit arithmetic on the owner segment's record, no atomics, no id map;
//                 consecutive lanes write consecutive vertices / faces.
// Output order is therefore a pure function of the input (needed so that frame-sharded ranks keep
// identical vertex numbering).  recmv_mc_run enqueues K0-K3 without any host round trip
// (the reference reads its counters back between its two kernels, CudaKernels.cu:628): the caller passes
// output capacities, K3 never writes past them, and the true sizes arrive in a device counter.
//
// Algorithmic bytes: 4*NX*NY*NZ (volume read once) + 12*V + 24*F.
#include "common.h"
#include "mc_tables.inc"

namespace recmv {
namespace {

constexpr int kBlk = 256;            // K0, K3: 4 waves
constexpr int kInBatch = 16;         // K0: loads in flight per wave (16 x 256 B)
constexpr int kScanBlk = 256;        // K1, K2: 4 waves (many small workgroups: both passes are latency-bound chains)
constexpr int kScanPer = 4;
constexpr int kScanChunk = kScanBlk * kScanPer;   // segments scanned per K2 block = kScanPer K1 workgroups

// One per 64-voxel segment with output.  The first 32 bytes are what a NEIGHBOUR needs to number a vertex.
struct McRec {
  unsigned long long mx, my;       // bit l: voxel k = 64 s + l owns a vertex on its +x / +y edge
  unsigned long long mz;           // ... +z edge
  unsigned int voff, toff;         // exclusive prefixes of vertices / triangles (written by K2)
  unsigned long long t0, t1;       // triangle count of voxel l, bits 0 and 1
  unsigned long long t2, ma;       // bit 2; voxels with any output
};
static_assert(sizeof(McRec) == 64, "McRec is one 64-byte line");

// count word of a segment: vertices (<= 192) | triangles (<= 320) << 8 | active voxels (<= 64) << 17
__host__ __device__ inline unsigned int pack_cnt(unsigned int nv, unsigned int nt, unsigned int na) {
  return nv | (nt << 8) | (na << 17);
}

struct McLayout {
  int S;              // segments per voxel row = ceil((NZ-1)/64)
  int64_t nrow;       // (NX-1)*(NY-1) voxel rows
  int64_t nseg;       // nrow*S
  int64_t nvox;       // NX*NY*NZ lattice points
  int64_t nword;      // ceil(nvox/64) words of inside bits (+2 words of slack for the funnel shifts)
  int64_t nchunk;     // ceil(nseg / kScanChunk)
  int64_t npart;      // nchunk * kScanPer: one partial sum per K1 workgroup (kScanBlk segments)
  // byte offsets into the workspace
  int64_t off_bits;   // uint64 [nword + 2]       bit f = sdf[f] < iso, flat lattice index f
  int64_t off_cnt;    // uint32 [nchunk * kScanChunk]
  int64_t off_part;   // uint32 [3][npart]        sums per kScanBlk segments (vertices, triangles, active voxels)
  int64_t off_rec;    // McRec  [nseg]            (written for segments with output only)
  int64_t off_act;    // uint32 [nseg*64]         active voxels (seg << 6 | lane), canonical order
  int64_t off_total;  // uint32 [4]               {vertices, faces, active voxels, -}
  int64_t bytes;
};

inline void make_layout(int64_t nx, int64_t ny, int64_t nz, McLayout* L) {
  L->S = (int)ceil_div(nz > 1 ? nz - 1 : 0, kWave);
  L->nrow = (nx > 1 && ny > 1) ? (nx - 1) * (ny - 1) : 0;
  L->nseg = L->nrow * L->S;
  L->nvox = nx * ny * nz;
  L->nword = ceil_div(L->nvox, 64);
  L->nchunk = ceil_div(L->nseg > 0 ? L->nseg : 1, kScanChunk);
  L->npart = L->nchunk * kScanPer;
  int64_t o = 0;
  auto take = [&](int64_t bytes) {
    int64_t r = o;
    o += (bytes + 255) / 256 * 256;
    return r;
  };
  L->off_bits = take((L->nword + 2) * 8);
  L->off_cnt = take(L->nchunk * kScanChunk * 4);
  L->off_part = take(3 * L->npart * 4);
  L->off_rec = take(L->nseg * (int64_t)sizeof(McRec));
  L->off_act = take(L->nseg * 64 * 4);
  L->off_total = take(16);
  L->bytes = o;
}

#pragma clang fp contract(off)

__device__ __forceinline__ float mc_offset(float v1, float v2, float iso) {
  const double delta = (double)(v2 - v1);  // float subtraction widened (:306)
  if (delta == 0.0) return 0.5f;
  return (float)((double)(iso - v1) / delta);
}

__device__ __forceinline__ int tri_count(unsigned long long word) {
  // number of non-0xF nibbles / 3; entries are packed from nibble 0 upwards
  const unsigned long long hi = word & (word >> 1) & (word >> 2) & (word >> 3) & 0x1111111111111111ull;
  return (16 - __popcll(hi)) / 3;
}

// ------------------------------------------------------------------------------------------- K0
// bits[w] bit b = (sdf[64 w + b] < iso).  A wave turns 64 words (4096 floats) per trip.
__device__ __forceinline__ void mc_inside_body(const float* __restrict__ sdf, int64_t total, float iso,
                                                         unsigned long long* __restrict__ bits, int64_t nword) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = (int64_t)blockIdx.x * (kBlk / kWave) + threadIdx.x / kWave;
  const int64_t nwave = (int64_t)gridDim.x * (kBlk / kWave);
  for (int64_t w0 = wave * kWave; w0 < nword + 2; w0 += nwave * kWave) {
    unsigned long long mine = 0ull;                   // lane q parks word w0 + q
    const bool interior = (w0 + kWave) * 64 <= total;
#pragma unroll 1
    for (int b = 0; b < kWave; b += kInBatch) {       // not unrolled: 16 ballots live at a time (SGPR budget)
      float v[kInBatch];
      if (interior) {
#pragma unroll
        for (int u = 0; u < kInBatch; ++u) v[u] = sdf[(w0 + b + u) * 64 + lane];
      } else {
#pragma unroll
        for (int u = 0; u < kInBatch; ++u) {
          const int64_t idx = (w0 + b + u) * 64 + lane;
          v[u] = idx < total ? sdf[idx] : iso;           // past the end: not inside
        }
      }
#pragma unroll
      for (int u = 0; u < kInBatch; ++u) {
        const unsigned long long m = __ballot(v[u] < iso);
        mine = lane == b + u ? m : mine;
      }
    }
    if (w0 + lane < nword + 2) bits[w0 + lane] = mine;
  }
}

// ------------------------------------------------------------------------------------------- K1
__device__ __forceinline__ void unpack_add(unsigned int w, unsigned int& v, unsigned int& t, unsigned int& a) {
  v += w & 0xFFu;
  t += (w >> 8) & 0x1FFu;
  a += w >> 17;
}

// block-wide sums of three counters; result valid in thread 0
__device__ __forceinline__ void block_sum3(unsigned int& v, unsigned int& t, unsigned int& a, unsigned int* sh_v,
                                           unsigned int* sh_t, unsigned int* sh_a) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    v += __shfl_xor(v, o);
    t += __shfl_xor(t, o);
    a += __shfl_xor(a, o);
  }
  if (lane == 0) {
    sh_v[wave] = v;
    sh_t[wave] = t;
    sh_a[wave] = a;
  }
  __syncthreads();
  if (tid == 0) {
    unsigned int x = 0, y = 0, z = 0;
    for (int w = 0; w < kScanBlk / kWave; ++w) {
      x += sh_v[w];
      y += sh_t[w];
      z += sh_a[w];
    }
    v = x;
    t = y;
    a = z;
  }
}

// inside bits of the 65 lattice points p .. p+64 of the flat volume: x0 = bits p..p+63, x1 = bits p+1..p+64
__device__ __forceinline__ void row_bits(const unsigned long long* __restrict__ bits, int64_t p, unsigned long long& x0,
                                         unsigned long long& x1) {
  const int64_t q = p >> 6;
  const int sh = (int)(p & 63);
  const unsigned long long w0 = bits[q], w1 = bits[q + 1];
  x0 = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
  x1 = sh == 63 ? w1 : (w0 >> (sh + 1)) | (w1 << (63 - sh));
}

// Workgroup q classifies segments [q*kScanBlk, (q+1)*kScanBlk), one per lane, and leaves their sums in part[][q].
__device__ __forceinline__ void mc_classify_body(const unsigned long long* __restrict__ bits, int NX,
                                                               int NY, int NZ, int S, int64_t nseg, int64_t npart,
                                                               unsigned int* __restrict__ cnt, McRec* __restrict__ rec,
                                                               unsigned int* __restrict__ part) {
  __shared__ unsigned int sh_v[kScanBlk / kWave], sh_t[kScanBlk / kWave], sh_a[kScanBlk / kWave];
  __shared__ unsigned char tcnt[256];
  const int tid = threadIdx.x;
  if (tid < 256) tcnt[tid] = (unsigned char)tri_count(kMcTriTable[tid]);     // 0 for cases 0 and 255
  __syncthreads();
  const int64_t slab = (int64_t)NY * NZ;
  unsigned int sv = 0, st = 0, sa = 0;
  const int64_t seg = (int64_t)blockIdx.x * kScanBlk + tid;       // consecutive lanes, consecutive segments
  if (seg < nseg) {
    const int s = (int)(seg % S);
    const int64_t row = seg / S;
    const int j = (int)(row % (NY - 1)), i = (int)(row / (NY - 1));
    const int64_t p00 = ((int64_t)i * NY + j) * NZ + (int64_t)s * kWave;
    // corner v: 0=(i,j,k) A0  1=(i+1,j,k) B0  2=(i+1,j+1,k) C0  3=(i,j+1,k) D0, 4..7 the same at k+1
    unsigned long long A0, A1, B0, B1, C0, C1, D0, D1;
    row_bits(bits, p00, A0, A1);
    row_bits(bits, p00 + slab, B0, B1);
    row_bits(bits, p00 + slab + NZ, C0, C1);
    row_bits(bits, p00 + NZ, D0, D1);
    const int nbits = NZ - 1 - s * kWave;                           // voxels of this segment that exist (>= 1)
    const unsigned long long kmask = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
    const unsigned long long any = A0 | B0 | C0 | D0 | A1 | B1 | C1 | D1;
    const unsigned long long all = A0 & B0 & C0 & D0 & A1 & B1 & C1 & D1;
    unsigned long long cut = kmask & any & ~all;                    // voxels whose cube is cut
    unsigned int word = 0u;
    if (cut) {
      const unsigned long long mx = kmask & (A0 ^ B0);              // edge 0: corners 0-1 (+x)
      const unsigned long long my = kmask & (A0 ^ D0);              // edge 3: corners 3-0 (+y)
      const unsigned long long mz = kmask & (A0 ^ A1);              // edge 8: corners 0-4 (+z)
      unsigned long long t0 = 0ull, t1 = 0ull, t2 = 0ull;
      unsigned int ntw = 0;
      while (cut) {
        const int m = __builtin_ctzll(cut);
        cut &= cut - 1ull;
        const int flag = (int)((A0 >> m) & 1ull) | ((int)((B0 >> m) & 1ull) << 1) | ((int)((C0 >> m) & 1ull) << 2) |
                         ((int)((D0 >> m) & 1ull) << 3) | ((int)((A1 >> m) & 1ull) << 4) |
                         ((int)((B1 >> m) & 1ull) << 5) | ((int)((C1 >> m) & 1ull) << 6) |
                         ((int)((D1 >> m) & 1ull) << 7);
        const unsigned int nt = tcnt[flag];
        ntw += nt;
        t0 |= (unsigned long long)(nt & 1u) << m;
        t1 |= (unsigned long long)((nt >> 1) & 1u) << m;
        t2 |= (unsigned long long)((nt >> 2) & 1u) << m;
      }
      const unsigned long long ma = mx | my | mz | t0 | t1 | t2;
      word = pack_cnt(__popcll(mx) + __popcll(my) + __popcll(mz), ntw, __popcll(ma));
      ulonglong2* r = reinterpret_cast<ulonglong2*>(rec + seg);
      r[0] = make_ulonglong2(mx, my);
      r[1] = make_ulonglong2(mz, 0ull);                             // (voff, toff): K2 writes them
      r[2] = make_ulonglong2(t0, t1);
      r[3] = make_ulonglong2(t2, ma);
    }
    cnt[seg] = word;
    unpack_add(word, sv, st, sa);
  }
  block_sum3(sv, st, sa, sh_v, sh_t, sh_a);
  if (tid == 0) {
    part[blockIdx.x] = sv;
    part[npart + blockIdx.x] = st;
    part[2 * npart + blockIdx.x] = sa;
  }
}

// ------------------------------------------------------------------------------------------- K2
// Block q scans segments [q*kScanChunk, (q+1)*kScanChunk); its base = sum of the K1 partial sums before the chunk.
__device__ __forceinline__ void mc_scan_body(const unsigned int* __restrict__ cnt, int64_t nseg,
                                                           int64_t npart, const unsigned int* __restrict__ part,
                                                           McRec* __restrict__ rec, unsigned int* __restrict__ act,
                                                           unsigned int* __restrict__ total) {
  __shared__ unsigned int sh_v[kScanBlk / kWave], sh_t[kScanBlk / kWave], sh_a[kScanBlk / kWave];
  __shared__ unsigned int base_v, base_t, base_a;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int64_t seg0 = (int64_t)blockIdx.x * kScanChunk;
  // the chunk's count words (each thread owns kScanPer consecutive segments) and the output masks of its segments
  // with output are requested first: both are in flight under the base reduction's barriers
  const uint4 mine = reinterpret_cast<const uint4*>(cnt)[seg0 / 4 + tid];
  unsigned int w[kScanPer] = {mine.x, mine.y, mine.z, mine.w};
  unsigned long long ma[kScanPer];
  unsigned int tv = 0, tt = 0, ta = 0;
#pragma unroll
  for (int e = 0; e < kScanPer; ++e) {
    const int64_t seg = seg0 + (int64_t)tid * kScanPer + e;
    if (seg >= nseg) w[e] = 0u;
    ma[e] = (w[e] >> 17) ? rec[seg].ma : 0ull;
    unpack_add(w[e], tv, tt, ta);
  }
  // (1) base offset
  unsigned int av = 0, at = 0, aa = 0;
  for (int64_t q = tid; q < (int64_t)blockIdx.x * kScanPer; q += kScanBlk) {
    av += part[q];
    at += part[npart + q];
    aa += part[2 * npart + q];
  }
  block_sum3(av, at, aa, sh_v, sh_t, sh_a);
  if (tid == 0) {
    base_v = av;
    base_t = at;
    base_a = aa;
  }
  __syncthreads();
  // (2) scan the chunk
  unsigned int iv = tv, itt = tt, ia = ta;                        // inclusive wave scan
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const unsigned int pv = __shfl_up(iv, o), pt = __shfl_up(itt, o), pa = __shfl_up(ia, o);
    if (lane >= o) {
      iv += pv;
      itt += pt;
      ia += pa;
    }
  }
  __syncthreads();  // sh_* reused
  if (lane == kWave - 1) {
    sh_v[wave] = iv;
    sh_t[wave] = itt;
    sh_a[wave] = ia;
  }
  __syncthreads();
  unsigned int wv = 0, wt = 0, wa = 0;
  for (int x = 0; x < wave; ++x) {
    wv += sh_v[x];
    wt += sh_t[x];
    wa += sh_a[x];
  }
  unsigned int ev = base_v + wv + iv - tv, et = base_t + wt + itt - tt, ea = base_a + wa + ia - ta;
#pragma unroll
  for (int e = 0; e < kScanPer; ++e) {
    const int64_t seg = seg0 + (int64_t)tid * kScanPer + e;
    if (w[e] >> 17) {                                             // segment with output: offsets + active voxels
      *reinterpret_cast<uint2*>(&rec[seg].voff) = make_uint2(ev, et);
      unsigned long long m = ma[e];
      const unsigned int tag = (unsigned int)seg << 6;
      unsigned int p = ea;
      while (m) {
        act[p++] = tag | (unsigned int)__builtin_ctzll(m);
        m &= m - 1ull;
      }
    }
    unpack_add(w[e], ev, et, ea);
  }
  // (3) grand totals from the last block
  if (blockIdx.x == gridDim.x - 1 && tid == kScanBlk - 1) {
    total[0] = ev;
    total[1] = et;
    total[2] = ea;
  }
}

// ------------------------------------------------------------------------------------------- K3
struct NbrRec {                      // first 32 bytes of a McRec
  unsigned long long mx, my, mz;
  unsigned int voff;
};

__device__ __forceinline__ NbrRec load_nbr(const McRec* __restrict__ rec, int64_t seg) {
  const uint4 a = reinterpret_cast<const uint4*>(rec + seg)[0];
  const uint4 b = reinterpret_cast<const uint4*>(rec + seg)[1];
  NbrRec r;
  r.mx = (unsigned long long)a.x | ((unsigned long long)a.y << 32);
  r.my = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
  r.mz = (unsigned long long)b.x | ((unsigned long long)b.y << 32);
  r.voff = b.z;
  return r;
}

// vertex id of the edge (lane l, direction dir) inside a segment record, or -1 if no vertex lives there
__device__ __forceinline__ int vertex_id(const NbrRec& r, int l, int dir) {
  const unsigned long long bit = 1ull << l;
  const unsigned long long m = dir == 0 ? r.mx : (dir == 1 ? r.my : r.mz);
  if (!(m & bit)) return -1;
  const unsigned long long lt = bit - 1ull;
  unsigned int rank = __popcll(r.mx & lt) + __popcll(r.my & lt) + __popcll(r.mz & lt);
  if (dir > 0) rank += (r.mx & bit) ? 1 : 0;
  if (dir > 1) rank += (r.my & bit) ? 1 : 0;
  return (int)(r.voff + rank);
}

__device__ __forceinline__ void mc_emit_body(const float* __restrict__ sdf, int NX, int NY, int NZ,
                                                       float iso, int S, int64_t nseg,
                                                       const McRec* __restrict__ rec,
                                                       const unsigned int* __restrict__ act,
                                                       const unsigned int* __restrict__ total, float xstep,
                                                       float ystep, float zstep, float xmin, float ymin,
                                                       float zmin, float* __restrict__ vertices, int64_t vcap,
                                                       long long* __restrict__ faces, int64_t fcap) {
  __shared__ unsigned long long tri_lds[256];
  tri_lds[threadIdx.x] = kMcTriTable[threadIdx.x];
  __syncthreads();
  const unsigned int nact = total[2];
  const int64_t row_stride = (int64_t)(NY - 1) * S;               // segments per voxel slab
  for (unsigned int t = blockIdx.x * kBlk + threadIdx.x; t < nact; t += gridDim.x * kBlk) {
    const unsigned int e = act[t];
    const int64_t seg = e >> 6;
    const int l = (int)(e & 63u);
    const int s = (int)(seg % S);
    const int64_t row = seg / S;
    const int j = (int)(row % (NY - 1)), i = (int)(row / (NY - 1));
    const int k = s * kWave + l;
    const uint4* rp = reinterpret_cast<const uint4*>(rec + seg);
    const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
    NbrRec self;
    self.mx = (unsigned long long)r0.x | ((unsigned long long)r0.y << 32);
    self.my = (unsigned long long)r0.z | ((unsigned long long)r0.w << 32);
    self.mz = (unsigned long long)r1.x | ((unsigned long long)r1.y << 32);
    self.voff = r1.z;
    const unsigned int toff = r1.w;
    const unsigned long long t0 = (unsigned long long)r2.x | ((unsigned long long)r2.y << 32);
    const unsigned long long t1 = (unsigned long long)r2.z | ((unsigned long long)r2.w << 32);
    const unsigned long long t2 = (unsigned long long)r3.x | ((unsigned long long)r3.y << 32);
    // Records of the segments a triangle corner of this voxel can live in — rows (i+1,j), (i,j+1), (i+1,j+1), each at
    // k and (for lane 63) k+1 = first voxel of the next segment.  Loaded before anything is known about the voxel
    // (addresses depend on the list entry only), so that they fly together with the corner loads; a record is only
    // USED when the edge it answers for is cut, which is exactly when its segment wrote it.
    const int64_t last = nseg - 1;
    const int64_t nxt = (l + 1) >> 6;
    const int64_t seg10 = seg + row_stride, seg01 = seg + S, seg11 = seg10 + S;
    const NbrRec n00 = load_nbr(rec, min(seg + nxt, last));
    const NbrRec r10 = load_nbr(rec, min(seg10, last)), n10 = load_nbr(rec, min(seg10 + nxt, last));
    const NbrRec r01 = load_nbr(rec, min(seg01, last)), n01 = load_nbr(rec, min(seg01 + nxt, last));
    const NbrRec r11 = load_nbr(rec, min(seg11, last));
    // the 8 corners of the voxel (4 lattice rows x (k, k+1)); the voxel is valid, so every index exists
    const int64_t b00 = ((int64_t)i * NY + j) * NZ + k, b10 = b00 + (int64_t)NY * NZ;
    float v[8];
    v[0] = sdf[b00];           v[4] = sdf[b00 + 1];
    v[3] = sdf[b00 + NZ];      v[7] = sdf[b00 + NZ + 1];
    v[1] = sdf[b10];           v[5] = sdf[b10 + 1];
    v[2] = sdf[b10 + NZ];      v[6] = sdf[b10 + NZ + 1];
    const unsigned long long bit = 1ull << l, lt = bit - 1ull;

    // ---- vertices owned by this lattice point (canonical order: dir 0,1,2)
    if ((self.mx | self.my | self.mz) & bit) {
      int64_t id = (int64_t)self.voff + __popcll(self.mx & lt) + __popcll(self.my & lt) + __popcll(self.mz & lt);
      const float fX = (float)i, fY = (float)j, fZ = (float)k;
      if (self.mx & bit) {  // edge 0: corner 0 -> 1, direction (+1,0,0)
        const float tt = mc_offset(v[0], v[1], iso);
        if (id < vcap) {
          vertices[3 * id + 0] = fmaf(fX + (0.f + tt * 1.f), xstep, xmin);
          vertices[3 * id + 1] = fmaf(fY + (0.f + tt * 0.f), ystep, ymin);
          vertices[3 * id + 2] = fmaf(fZ + (0.f + tt * 0.f), zstep, zmin);
        }
        ++id;
      }
      if (self.my & bit) {  // edge 3: corner 3 -> 0, direction (0,-1,0), start offset (0,1,0)
        const float tt = mc_offset(v[3], v[0], iso);
        if (id < vcap) {
          vertices[3 * id + 0] = fmaf(fX + (0.f + tt * 0.f), xstep, xmin);
          vertices[3 * id + 1] = fmaf(fY + (1.f + tt * -1.f), ystep, ymin);
          vertices[3 * id + 2] = fmaf(fZ + (0.f + tt * 0.f), zstep, zmin);
        }
        ++id;
      }
      if (self.mz & bit) {  // edge 8: corner 0 -> 4, direction (0,0,+1)
        const float tt = mc_offset(v[0], v[4], iso);
        if (id < vcap) {
          vertices[3 * id + 0] = fmaf(fX + (0.f + tt * 0.f), xstep, xmin);
          vertices[3 * id + 1] = fmaf(fY + (0.f + tt * 0.f), ystep, ymin);
          vertices[3 * id + 2] = fmaf(fZ + (0.f + tt * 1.f), zstep, zmin);
        }
      }
    }

    // ---- faces
    const int nt = (int)((t0 >> l) & 1ull) | ((int)((t1 >> l) & 1ull) << 1) | ((int)((t2 >> l) & 1ull) << 2);
    if (nt == 0) continue;
    int flag = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) flag |= (v[c] < iso) ? (1 << c) : 0;
    const unsigned long long word = tri_lds[flag];
    const int64_t f0 = (int64_t)toff + __popcll(t0 & lt) + 2 * __popcll(t1 & lt) + 4 * __popcll(t2 & lt);
    // edges used by this voxel's triangles
    unsigned int used = 0;
    for (int n = 0; n < 3 * nt; ++n) used |= 1u << ((word >> (4 * n)) & 0xF);
    // edge id -> (di, dj, dk, dir)   (the if/else ladder at CudaKernels.cu:385-456)
    // e:   0        1        2        3        4        5        6        7        8        9       10       11
    // di   0        1        0        0        0        1        0        0        0        1        1        0
    // dj   0        0        1        0        0        0        1        0        0        0        1        1
    // dk   0        0        0        0        1        1        1        1        0        0        0        0
    // dir  0        1        0        1        0        1        0        1        2        2        2        2
    int ids[12];
#pragma unroll
    for (int ed = 0; ed < 12; ++ed) {
      const int di = (0x622 >> ed) & 1, dj = (0xC44 >> ed) & 1, dk = (0x0F0 >> ed) & 1;
      const int dir = ed >= 8 ? 2 : (ed & 1);
      int id = -1;
      if ((used >> ed) & 1u) {
        const int k2 = k + dk;
        // an edge whose owner voxel lies outside the grid of cubes carries no vertex (-1), as in the reference
        if (i + di < NX - 1 && j + dj < NY - 1 && k2 < NZ - 1) {
          const int l2 = (l + dk) & 63;
          const NbrRec& r = di ? (dj ? r11 : (dk ? n10 : r10)) : (dj ? (dk ? n01 : r01) : (dk ? n00 : self));
          id = vertex_id(r, l2, dir);
        }
      }
      ids[ed] = id;
    }
    for (int tr = 0; tr < nt; ++tr) {
      if (f0 + tr >= fcap) break;
#pragma unroll
      for (int corner = 0; corner < 3; ++corner) {
        const int ed = (int)((word >> (4 * (3 * tr + corner))) & 0xF);
        int id = ids[0];
#pragma unroll
        for (int q = 1; q < 12; ++q) id = ed == q ? ids[q] : id;
        faces[(f0 + tr) * 3 + (2 - corner)] = (long long)id;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------- launch forms
// One volume per launch (grid x), or up to kMcMaxBatch volumes of ONE lattice size per launch (grid y = volume): the three nets of a
// re-mesh (body + two garments, OptimGarmentNetwork.py:581-618) share every launch — the passes after K0 work on kilobytes and are
// launch-latency bound, so three volumes take hardly longer than one.
constexpr int kMcMaxBatch = 4;
struct McBatch {
  const float* sdf[kMcMaxBatch];
  char* ws[kMcMaxBatch];
  float* vertices[kMcMaxBatch];
  long long* faces[kMcMaxBatch];
  int64_t vcap[kMcMaxBatch], fcap[kMcMaxBatch];
};

__global__ __launch_bounds__(kBlk) void mc_inside_kernel(const float* __restrict__ sdf, int64_t total, float iso,
                                                         unsigned long long* __restrict__ bits, int64_t nword) {
  mc_inside_body(sdf, total, iso, bits, nword);
}
__global__ __launch_bounds__(kScanBlk) void mc_classify_kernel(const unsigned long long* __restrict__ bits, int NX, int NY, int NZ,
                                                               int S, int64_t nseg, int64_t npart, unsigned int* __restrict__ cnt,
                                                               McRec* __restrict__ rec, unsigned int* __restrict__ part) {
  mc_classify_body(bits, NX, NY, NZ, S, nseg, npart, cnt, rec, part);
}
__global__ __launch_bounds__(kScanBlk) void mc_scan_kernel(const unsigned int* __restrict__ cnt, int64_t nseg, int64_t npart,
                                                           const unsigned int* __restrict__ part, McRec* __restrict__ rec,
                                                           unsigned int* __restrict__ act, unsigned int* __restrict__ total) {
  mc_scan_body(cnt, nseg, npart, part, rec, act, total);
}
__global__ __launch_bounds__(kBlk) void mc_emit_kernel(const float* __restrict__ sdf, int NX, int NY, int NZ, float iso, int S,
                                                       int64_t nseg, const McRec* __restrict__ rec,
                                                       const unsigned int* __restrict__ act, const unsigned int* __restrict__ total,
                                                       float xstep, float ystep, float zstep, float xmin, float ymin, float zmin,
                                                       float* __restrict__ vertices, int64_t vcap, long long* __restrict__ faces,
                                                       int64_t fcap) {
  mc_emit_body(sdf, NX, NY, NZ, iso, S, nseg, rec, act, total, xstep, ystep, zstep, xmin, ymin, zmin, vertices, vcap, faces, fcap);
}

__global__ __launch_bounds__(kBlk) void mc_inside_batch_kernel(McBatch b, McLayout L, float iso) {
  const int v = blockIdx.y;
  mc_inside_body(b.sdf[v], L.nvox, iso, (unsigned long long*)(b.ws[v] + L.off_bits), L.nword);
}
__global__ __launch_bounds__(kScanBlk) void mc_classify_batch_kernel(McBatch b, McLayout L, int NX, int NY, int NZ) {
  char* ws = b.ws[blockIdx.y];
  mc_classify_body((const unsigned long long*)(ws + L.off_bits), NX, NY, NZ, L.S, L.nseg, L.npart, (unsigned int*)(ws + L.off_cnt),
                   (McRec*)(ws + L.off_rec), (unsigned int*)(ws + L.off_part));
}
__global__ __launch_bounds__(kScanBlk) void mc_scan_batch_kernel(McBatch b, McLayout L) {
  char* ws = b.ws[blockIdx.y];
  mc_scan_body((const unsigned int*)(ws + L.off_cnt), L.nseg, L.npart, (const unsigned int*)(ws + L.off_part), (McRec*)(ws + L.off_rec),
               (unsigned int*)(ws + L.off_act), (unsigned int*)(ws + L.off_total));
}
__global__ __launch_bounds__(kBlk) void mc_emit_batch_kernel(McBatch b, McLayout L, int NX, int NY, int NZ, float iso, float xstep,
                                                             float ystep, float zstep, float xmin, float ymin, float zmin) {
  const int v = blockIdx.y;
  const char* ws = b.ws[v];
  mc_emit_body(b.sdf[v], NX, NY, NZ, iso, L.S, L.nseg, (const McRec*)(ws + L.off_rec), (const unsigned int*)(ws + L.off_act),
               (const unsigned int*)(ws + L.off_total), xstep, ystep, zstep, xmin, ymin, zmin, b.vertices[v], b.vcap[v], b.faces[v],
               b.fcap[v]);
}

}  // namespace
}  // namespace recmv

using namespace recmv;

extern "C" int64_t recmv_mc_workspace_bytes(int64_t nx, int64_t ny, int64_t nz) {
  if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
  if (nx >= (1 << 15) || ny >= (1 << 15) || nz >= (1 << 15) || nx * ny * nz > (1ll << 28)) return 0;
  McLayout L;
  make_layout(nx, ny, nz, &L);
  return L.bytes;
}

static int mc_check(const char* who, const float* sdf, int64_t nx, int64_t ny, int64_t nz, const void* ws,
                    int64_t ws_bytes, McLayout* L) {
  RECMV_REQUIRE(nx > 0 && ny > 0 && nz > 0, "%s: empty volume (%lld,%lld,%lld)", who, (long long)nx,
                (long long)ny, (long long)nz);
  RECMV_REQUIRE(nx < (1 << 15) && ny < (1 << 15) && nz < (1 << 15) && nx * ny * nz <= (1ll << 28),
                "%s: volume too large for 32-bit vertex/face counters (at most 2^28 lattice points)", who);
  make_layout(nx, ny, nz, L);
  RECMV_REQUIRE(sdf && ws, "%s: NULL pointer", who);
  RECMV_REQUIRE((reinterpret_cast<uintptr_t>(sdf) & 3) == 0, "%s: volume must be 4-byte aligned", who);
  if (ws_bytes < L->bytes) {
    set_error("%s: workspace %lld < %lld bytes", who, (long long)ws_bytes, (long long)L->bytes);
    return RECMV_ERR_WORKSPACE;
  }
  RECMV_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "%s: workspace must be 256-byte aligned", who);
  return RECMV_OK;
}

// K1 + K2 on the stream; the totals stay in the workspace.
static int mc_launch_count(const float* sdf, int64_t nx, int64_t ny, int64_t nz, float iso, const McLayout& L,
                           char* ws, hipStream_t s) {
  auto* bits = (unsigned long long*)(ws + L.off_bits);
  auto* cnt = (unsigned int*)(ws + L.off_cnt);
  auto* rec = (McRec*)(ws + L.off_rec);
  auto* act = (unsigned int*)(ws + L.off_act);
  auto* part = (unsigned int*)(ws + L.off_part);
  auto* total = (unsigned int*)(ws + L.off_total);
  // K0: a wave turns 64 words per trip; enough waves for one trip each, at most 8 workgroups per CU
  int64_t g0 = ceil_div(ceil_div(L.nword + 2, kWave), kBlk / kWave);
  if (g0 > kNumCU * 8) g0 = kNumCU * 8;
  hipLaunchKernelGGL(mc_inside_kernel, dim3((int)g0), dim3(kBlk), 0, s, sdf, L.nvox, iso, bits, L.nword);
  int rc = check_launch("mc_inside");
  if (rc) return rc;
  hipLaunchKernelGGL(mc_classify_kernel, dim3((int)L.npart), dim3(kScanBlk), 0, s, bits, (int)nx, (int)ny, (int)nz,
                     L.S, L.nseg, L.npart, cnt, rec, part);
  rc = check_launch("mc_classify");
  if (rc) return rc;
  hipLaunchKernelGGL(mc_scan_kernel, dim3((int)L.nchunk), dim3(kScanBlk), 0, s, cnt, L.nseg, L.npart, part, rec, act,
                     total);
  return check_launch("mc_scan");
}

static int mc_launch_emit(const float* sdf, int64_t nx, int64_t ny, int64_t nz, float iso, float xstep, float ystep,
                          float zstep, float xmin, float ymin, float zmin, const McLayout& L, const char* ws,
                          int64_t n_active_bound, float* vertices, int64_t vcap, int64_t* faces, int64_t fcap,
                          hipStream_t s) {
  auto* rec = (const McRec*)(ws + L.off_rec);
  auto* act = (const unsigned int*)(ws + L.off_act);
  auto* total = (const unsigned int*)(ws + L.off_total);
  int64_t g3 = ceil_div(n_active_bound, kBlk);
  g3 = g3 < 1 ? 1 : (g3 > 4096 ? 4096 : g3);
  hipLaunchKernelGGL(mc_emit_kernel, dim3((int)g3), dim3(kBlk), 0, s, sdf, (int)nx, (int)ny, (int)nz, iso, L.S, L.nseg, rec,
                     act, total, xstep, ystep, zstep, xmin, ymin, zmin, vertices, vcap, (long long*)faces, fcap);
  return check_launch("mc_emit");
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
  rc = mc_launch_count(sdf, nx, ny, nz, iso, L, (char*)workspace, s);
  if (rc) return rc;
  uint32_t host[3] = {0, 0, 0};
  RECMV_HIP_TRY(hipMemcpyAsync(host, (char*)workspace + L.off_total, 12, hipMemcpyDeviceToHost, s));
  RECMV_HIP_TRY(hipStreamSynchronize(s));
  counts_host[0] = (int32_t)host[0];
  counts_host[1] = (int32_t)host[1];
  counts_host[2] = (int32_t)host[2];
  return RECMV_OK;
}

extern "C" int recmv_mc_emit(const float* sdf, int64_t nx, int64_t ny, int64_t nz, float iso, float xstep,
                             float ystep, float zstep, float xmin, float ymin, float zmin,
                             const void* workspace, int64_t workspace_bytes, int64_t n_active,
                             float* vertices, int64_t vertex_capacity, int64_t* faces, int64_t face_capacity,
                             void* stream) {
  McLayout L;
  int rc = mc_check("mc_emit", sdf, nx, ny, nz, workspace, workspace_bytes, &L);
  if (rc) return rc;
  if (L.nseg == 0 || n_active <= 0) return RECMV_OK;
  RECMV_REQUIRE(n_active <= L.nseg * 64, "mc_emit: n_active out of range");
  RECMV_REQUIRE(vertex_capacity >= 0 && face_capacity >= 0 && (vertices || vertex_capacity == 0) &&
                    (faces || face_capacity == 0), "mc_emit: bad output buffers");
  return mc_launch_emit(sdf, nx, ny, nz, iso, xstep, ystep, zstep, xmin, ymin, zmin, L, (const char*)workspace,
                        n_active, vertices, vertex_capacity, faces, face_capacity, (hipStream_t)stream);
}

extern "C" int recmv_mc_run(const float* sdf, int64_t nx, int64_t ny, int64_t nz, float iso, float xstep,
                            float ystep, float zstep, float xmin, float ymin, float zmin, void* workspace,
                            int64_t workspace_bytes, float* vertices, int64_t vertex_capacity, int64_t* faces,
                            int64_t face_capacity, int32_t* counts_device, void* stream) {
  McLayout L;
  int rc = mc_check("mc_run", sdf, nx, ny, nz, workspace, workspace_bytes, &L);
  if (rc) return rc;
  RECMV_REQUIRE(counts_device, "mc_run: NULL counts_device");
  RECMV_REQUIRE(vertex_capacity >= 0 && face_capacity >= 0 && (vertices || vertex_capacity == 0) &&
                    (faces || face_capacity == 0), "mc_run: bad output buffers");
  hipStream_t s = (hipStream_t)stream;
  if (L.nseg == 0) {
    RECMV_HIP_TRY(hipMemsetAsync(counts_device, 0, 12, s));
    return RECMV_OK;
  }
  rc = mc_launch_count(sdf, nx, ny, nz, iso, L, (char*)workspace, s);
  if (rc) return rc;
  // the number of active voxels is not known on the host: a grid sized for the capacity, lanes stride over the list
  const int64_t bound = face_capacity > vertex_capacity ? face_capacity : vertex_capacity;
  rc = mc_launch_emit(sdf, nx, ny, nz, iso, xstep, ystep, zstep, xmin, ymin, zmin, L, (const char*)workspace,
                      bound > 0 ? bound : 1, vertices, vertex_capacity, faces, face_capacity, s);
  if (rc) return rc;
  RECMV_HIP_TRY(hipMemcpyAsync(counts_device, (char*)workspace + L.off_total, 12, hipMemcpyDeviceToDevice, s));
  return RECMV_OK;
}

extern "C" int recmv_mc_run_batch(int n, const float* const* sdf, int64_t nx, int64_t ny, int64_t nz, float iso, float xstep,
                                  float ystep, float zstep, float xmin, float ymin, float zmin, void* const* workspaces,
                                  int64_t workspace_bytes, float* const* vertices, const int64_t* vertex_capacity,
                                  int64_t* const* faces, const int64_t* face_capacity, int32_t* const* counts_device,
                                  void* stream) {
  RECMV_REQUIRE(n >= 1 && n <= kMcMaxBatch, "mc_run_batch: 1..%d volumes per call, got %d", kMcMaxBatch, n);
  RECMV_REQUIRE(sdf && workspaces && vertices && vertex_capacity && faces && face_capacity && counts_device,
                "mc_run_batch: NULL argument array");
  McLayout L;
  McBatch b;
  memset(&b, 0, sizeof(b));
  int64_t bound = 1;
  for (int v = 0; v < n; ++v) {
    int rc = mc_check("mc_run_batch", sdf[v], nx, ny, nz, workspaces[v], workspace_bytes, &L);
    if (rc) return rc;
    RECMV_REQUIRE(counts_device[v], "mc_run_batch: NULL counts_device[%d]", v);
    RECMV_REQUIRE(vertex_capacity[v] >= 0 && face_capacity[v] >= 0 && (vertices[v] || vertex_capacity[v] == 0) &&
                      (faces[v] || face_capacity[v] == 0), "mc_run_batch: bad output buffers of volume %d", v);
    for (int u = 0; u < v; ++u) RECMV_REQUIRE(workspaces[u] != workspaces[v], "mc_run_batch: volumes %d and %d share a workspace", u, v);
    b.sdf[v] = sdf[v];
    b.ws[v] = (char*)workspaces[v];
    b.vertices[v] = vertices[v];
    b.faces[v] = (long long*)faces[v];
    b.vcap[v] = vertex_capacity[v];
    b.fcap[v] = face_capacity[v];
    const int64_t m = face_capacity[v] > vertex_capacity[v] ? face_capacity[v] : vertex_capacity[v];
    bound = m > bound ? m : bound;
  }
  hipStream_t s = (hipStream_t)stream;
  if (L.nseg == 0) {
    for (int v = 0; v < n; ++v) RECMV_HIP_TRY(hipMemsetAsync(counts_device[v], 0, 12, s));
    return RECMV_OK;
  }
  int64_t g0 = ceil_div(ceil_div(L.nword + 2, kWave), kBlk / kWave);
  if (g0 > kNumCU * 8 / n) g0 = kNumCU * 8 / n;
  hipLaunchKernelGGL(mc_inside_batch_kernel, dim3((int)g0, n), dim3(kBlk), 0, s, b, L, iso);
  int rc = check_launch("mc_inside(batch)");
  if (rc) return rc;
  hipLaunchKernelGGL(mc_classify_batch_kernel, dim3((int)L.npart, n), dim3(kScanBlk), 0, s, b, L, (int)nx, (int)ny, (int)nz);
  rc = check_launch("mc_classify(batch)");
  if (rc) return rc;
  hipLaunchKernelGGL(mc_scan_batch_kernel, dim3((int)L.nchunk, n), dim3(kScanBlk), 0, s, b, L);
  rc = check_launch("mc_scan(batch)");
  if (rc) return rc;
  int64_t g3 = ceil_div(bound, kBlk);
  g3 = g3 < 1 ? 1 : (g3 > 4096 ? 4096 : g3);
  hipLaunchKernelGGL(mc_emit_batch_kernel, dim3((int)g3, n), dim3(kBlk), 0, s, b, L, (int)nx, (int)ny, (int)nz, iso, xstep, ystep, zstep,
                     xmin, ymin, zmin);
  rc = check_launch("mc_emit(batch)");
  if (rc) return rc;
  for (int v = 0; v < n; ++v)
    RECMV_HIP_TRY(hipMemcpyAsync(counts_device[v], (char*)workspaces[v] + L.off_total, 12, hipMemcpyDeviceToDevice, s));
  return RECMV_OK;
}
