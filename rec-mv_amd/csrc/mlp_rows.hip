// Row-tile-persistent fused MLP for the ray path — gfx950 (CDNA4, wave64, f32-input MFMA).
//
// The surface root finder (utils/FindSurfacePs.py:273-353 of the reference) evaluates, up to 21 times per iteration and
// garment, the SDF net with its input gradient (model/network.py:98-133) and the deformer's offset MLP with a
// vector-Jacobian product to its input (model/Deformer.py:141-206) on a few thousand rays.  csrc/mlp_chain.hip enqueues
// those passes as one launch per layer: at 3 k rows a 512 x 512 layer is ONE round of workgroups, a third of its 22 us is
// launch ramp / first operand tile / epilogue / drain, and the activations travel through L2 between every two launches.
//
// Here ONE launch evaluates a whole pass.  A workgroup (8 waves) owns a tile of 16 rays for all layers:
//   * the tile's activations stay in LDS (two buffers of 16 x 520 floats, ping-pong; the row stride of 520 = 8 mod 64 makes
//     every ds_read_b128 of an A fragment conflict-free for the 16x16x4 lane map), positional encoding, per-frame code
//     gather, skip concatenation, bias, activation, residual and — in the reverse pass — the activation gradient and the
//     encoding's VJP are all done on the tile in place;
//   * the weights are streamed from L2 straight into MFMA B fragments: recmv_mlp_pack lays every layer out ONCE per weight
//     version in fragment order (tile of 16 outputs x chunk of 16 inputs = 64 lanes x 16 bytes = one fully coalesced 1 KB
//     wave load, zero-padded, so the loop has no guards), each wave owns an eighth of the layer's output columns and keeps
//     the next chunk's fragments in flight under the 16 MFMAs of the current one (two waves per SIMD take turns on the pipe);
//   * v_mfma_f32_16x16x4_f32: exact f32 products and accumulation (an fma chain) like the layer kernels of gemm_f32.hip;
//     the k order inside a chunk differs from theirs, so results agree to rounding, not bitwise.  Rows are independent: a
//     ray gets the same bits whatever tile it sits in.
// With 16-row tiles every weight element fetched feeds 16 rows — 8 FLOP per byte from L2 — and between two layers the workgroup
// runs its epilogue and a barrier with the matrix pipe idle (profiles/r04_mlp_rows_clock.txt: products at 77 % of the pipe, 63 %
// over a pass), which is why this form is for the few-thousand-row passes only; recmv_mlp_forward keeps the per-layer kernels
// above RECMV_MLP_ROWS_MAX rows.
#include "common.h"
// RECMV_LIBM_SOFTPLUS (an experiment build of tools/trajectory_seeds.py, never the product's): the activation through the
// correctly-rounded-to-an-ulp library functions instead of the hardware exp2 / log2 units.
#ifdef RECMV_LIBM_SOFTPLUS
#define RECMV_EXPF(x) expf(x)
#define RECMV_LOG1PF(t) log1pf(t)
#else
#define RECMV_EXPF(x) __expf(x)
#define RECMV_LOG1PF(t) __logf(1.f + (t))
#endif


namespace recmv {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;              // two waves per SIMD: one issues MFMAs while the other waits for its weight fragments
constexpr int kThreads = 64 * kWaves;  // each wave owns an eighth of a layer's output columns
constexpr int kLD = 520;               // LDS row stride in floats (= 8 mod 64, >= 512 + 8)
constexpr int kMaxWidth = 512;
constexpr int kPeLD = 52;              // gamma(x) rows kept for the skip connection (3 + 6 * 8 = 51 max)
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kSqrt2 = 1.41421356237309504880f;

// -DRECMV_ROWS_TIMING: thread 0 of workgroup 0 stamps the 100 MHz wall clock at the phase boundaries of a pass (read back
// through recmv_debug_rows_clock, tools/mlp_rows_clock.py).  Not part of the shipped build.
#ifdef RECMV_ROWS_TIMING
__device__ long long g_clk[512];
__device__ int g_clk_n;
#define ROWS_STAMP()                                                     \
  do {                                                                   \
    if (blockIdx.x == 0 && threadIdx.x == 0 && g_clk_n < 512) {          \
      g_clk[g_clk_n] = (long long)wall_clock64();                        \
      g_clk_n = g_clk_n + 1;                                             \
    }                                                                    \
  } while (0)
#else
#define ROWS_STAMP() do { } while (0)
#endif

struct RowsLayer {
  const float* Wp;       // packed weights: [tile][chunk][lane][4]
  const float* bias;     // forward only
  int32_t N;             // valid output columns
  int32_t KC;            // chunks of 16 inputs (= chunk stride of a tile in the packed array)
  int32_t TPW;           // 16-column tiles per wave (1 .. 4)
  int32_t pad_;
};

struct RowsArgs {
  RowsLayer fwd[RECMV_MLP_MAX_LAYERS];
  RowsLayer bwd[RECMV_MLP_MAX_LAYERS];
  const float* W_last;   // un-packed last weight (row 0 = the broadcast cotangent of a scalar output)
  int32_t n_layers, multires, cond_dim, skip_layer, hidden_act, residual;
  int32_t dims[RECMV_MLP_MAX_LAYERS + 1];
  int32_t rows[RECMV_MLP_MAX_LAYERS];
  float act_param;
  float pe_w[32];
};

// Softplus is the hot activation (SDF net, offset MLP): straight-line code, both branches of the small-t split evaluated and
// selected, so that the epilogue of a tile is one basic block.  The formulas are those of gemm_f32.hip's epilogue.
__device__ __forceinline__ float softplus_fwd(float z, float p, float inv_p) {
  const float zb = z * p;
  const float t = RECMV_EXPF(-fabsf(zb));
  const float series = t * (1.f - t * (0.5f - t * (0.33333334f - 0.25f * t)));
  #ifdef RECMV_LIBM_SOFTPLUS
  const float lg = log1pf(t);
#else
  const float lg = __log2f(1.f + t) * 0.69314718f;          // 1 + t in [1, 2]: the bare v_log_f32, no range fix-ups
#endif
  const float l = t < 0.015625f ? series : lg;
  const float y = (fmaxf(zb, 0.f) + l) * inv_p;
  return zb > 20.f ? z : y;
}

__device__ __forceinline__ float softplus_grad(float y, float p) {   // sigmoid(beta z) = 1 - exp(-beta y), series below 1/64
  const float t = p * y;
  const float series = t * (1.f - t * (0.5f - t * (0.16666667f - 0.041666668f * t)));
  const float e = 1.f - RECMV_EXPF(-t);
  return t < 0.015625f ? series : e;
}

// ACT: a compile-time activation (RECMV_ACT_SOFTPLUS) or -1 = look at the run-time value.
template <int ACT>
__device__ __forceinline__ float act_fwd(float z, int act, float p, float inv_p) {
  if (ACT == RECMV_ACT_SOFTPLUS) return softplus_fwd(z, p, inv_p);
  if (act == RECMV_ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == RECMV_ACT_SOFTPLUS) return softplus_fwd(z, p, inv_p);
  if (act == RECMV_ACT_TANH) return tanhf(z);
  return z;
}

template <int ACT>
__device__ __forceinline__ float act_grad(float y, int act, float p) {   // act'(z) through y = act(z) (mlp_chain.hip)
  if (ACT == RECMV_ACT_SOFTPLUS) return softplus_grad(y, p);
  switch (act) {
    case RECMV_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case RECMV_ACT_SOFTPLUS: return softplus_grad(y, p);
    case RECMV_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// acc[rt][t] += A_rt[16 x 16*KC] . B_t^T for the wave's TPW column tiles and the workgroup's RT row tiles.  A: LDS, row stride
// kLD, lane (r = l & 15, j = l >> 4) reads the four inputs 16c + 4j .. + 3 of row 16 rt + r with one ds_read_b128; the packed B
// chunk holds the same four inputs of output column 16t + r for that lane, and feeds RT MFMAs: with RT = 2 a byte of weights from
// L2 does 16 FLOP instead of 8.
template <int TPW, int RT>
struct Frag {
  f32x4 b[TPW];
  f32x4 a[RT];
};

template <int TPW, int RT>
__device__ __forceinline__ void chunk_load(const f32x4* __restrict__ wp, int64_t tile_stride, int c, const float* __restrict__ arow,
                                           Frag<TPW, RT>& f) {
#pragma unroll
  for (int t = 0; t < TPW; ++t) f.b[t] = wp[t * tile_stride + (int64_t)c * 64];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) f.a[rt] = *reinterpret_cast<const f32x4*>(arow + rt * 16 * kLD + 16 * c);
}

template <int TPW, int RT>
__device__ __forceinline__ void chunk_mma(const Frag<TPW, RT>& f, bool live, f32x4 (&acc)[RT][TPW]) {
  f32x4 a[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) a[rt] = live ? f.a[rt] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][s], f.b[t][s], acc[rt][t], 0, 0, 0);
    }
  }
}

// Two register sets, the loop unrolled by two: the fragments of chunk c + 1 are REQUESTED before the MFMAs of chunk c are issued
// (the scheduling barriers keep the compiler from sinking the requests behind the matrix instructions), so a set has a whole chunk
// (4 * TPW * RT MFMAs, 512 cycles at TPW = 4, while the SIMD's other wave issues its own) to arrive.  No exit between a request and its use (a load whose only use sits
// behind a branch gets sunk behind that branch): the trip count is rounded up to a pair, a chunk past the end re-requests the last
// one and multiplies it by a zero A fragment.  (A ring of four sets measured the same, profiles/r04_mlp_rows_bench_v2.txt.)
template <int TPW, int RT>
__device__ __forceinline__ void tile_mma(const float* __restrict__ A, int KC, const float* __restrict__ Wp, int KCs, int wave,
                                         int lane, f32x4 (&acc)[RT][TPW]) {
  const float* arow = A + (lane & 15) * kLD + 4 * (lane >> 4);
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((int64_t)wave * TPW * KCs) * 64 + lane;
  const int64_t ts = (int64_t)KCs * 64;
  const int last = KC - 1;
  Frag<TPW, RT> f0, f1;
  chunk_load<TPW, RT>(wp, ts, 0, arow, f0);
  for (int c = 0; c < KC; c += 2) {
    chunk_load<TPW, RT>(wp, ts, c + 1 < last ? c + 1 : last, arow, f1);
    __builtin_amdgcn_sched_barrier(0);
    chunk_mma<TPW, RT>(f0, true, acc);
    __builtin_amdgcn_sched_barrier(0);
    chunk_load<TPW, RT>(wp, ts, c + 2 < last ? c + 2 : last, arow, f0);
    __builtin_amdgcn_sched_barrier(0);
    chunk_mma<TPW, RT>(f1, c + 1 < KC, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------------------------------------ forward
// Epilogue of a hidden layer: bias + activation (+ 1/sqrt2 in front of the skip concatenation) from the accumulators into the
// other LDS buffer; columns >= N (tile padding) are written as zeros unless the encoding follows there.
template <int TPW, int RT, int ACT>
__device__ __forceinline__ void hidden_layer(const RowsLayer& L, const float* __restrict__ in, float* __restrict__ outb, int wave,
                                             int lane, int act, float p, float inv_p, float scale, bool zero_pad) {
  f32x4 acc[RT][TPW];
  float bv[TPW];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the bias travels under the products (one clamped, unconditional load per column tile)
  if (L.bias) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int n = (wave * TPW + t) * 16 + (lane & 15);
      bv[t] = L.bias[n < L.N ? n : L.N - 1];
    }
  } else {
#pragma unroll
    for (int t = 0; t < TPW; ++t) bv[t] = 0.f;
  }
  tile_mma<TPW, RT>(in, L.KC, L.Wp, L.KC, wave, lane, acc);
  ROWS_STAMP();
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int n = (wave * TPW + t) * 16 + (lane & 15);
    const bool ok = n < L.N;
    // (behind the skip concatenation the columns past N belong to the encoding, written by the caller)
    const bool store = ok || (zero_pad && n < kMaxWidth);
    float v[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float y = act_fwd<ACT>(acc[rt][t][i] + bv[t], act, p, inv_p) * scale;
        v[rt][i] = ok ? y : 0.f;
      }
    }
    if (store) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) outb[(16 * rt + 4 * (lane >> 4) + i) * kLD + n] = v[rt][i];
      }
    }
  }
}

template <int RT, int ACT>
__device__ __forceinline__ void run_hidden_act(const RowsLayer& L, const float* in, float* outb, int wave, int lane, int act,
                                               float p, float inv_p, float scale, bool zero_pad) {
  switch (L.TPW) {
    case 1: hidden_layer<1, RT, ACT>(L, in, outb, wave, lane, act, p, inv_p, scale, zero_pad); break;
    case 2: hidden_layer<2, RT, ACT>(L, in, outb, wave, lane, act, p, inv_p, scale, zero_pad); break;
    case 3: hidden_layer<3, RT, ACT>(L, in, outb, wave, lane, act, p, inv_p, scale, zero_pad); break;
    default: hidden_layer<4, RT, ACT>(L, in, outb, wave, lane, act, p, inv_p, scale, zero_pad); break;
  }
}

template <int RT>
__device__ __forceinline__ void run_hidden(const RowsLayer& L, const float* in, float* outb, int wave, int lane, int act, float p,
                                           float inv_p, float scale, bool zero_pad) {
  if (act == RECMV_ACT_SOFTPLUS) run_hidden_act<RT, RECMV_ACT_SOFTPLUS>(L, in, outb, wave, lane, act, p, inv_p, scale, zero_pad);
  else run_hidden_act<RT, -1>(L, in, outb, wave, lane, act, p, inv_p, scale, zero_pad);
}

// x [P,3] -> out [P,n_out] (n_out <= 16): positional encoding (+ per-frame code) -> hidden layers -> last layer, one launch.
// A workgroup owns kR = 16 RT rows.  keep: every hidden activation tile is also written to acts[l] ([P32, ld_act] per layer) for
// mlp_rows_vjp_kernel.
template <int RT>
__global__ __launch_bounds__(kThreads) void mlp_rows_fwd_kernel(RowsArgs a, const float* __restrict__ x,
                                                                const float* __restrict__ cond, int64_t ld_cond,
                                                                const int64_t* __restrict__ cond_index, int64_t P, int n_out,
                                                                float* __restrict__ out, int64_t ldo, float* __restrict__ acts,
                                                                int64_t ld_act, int64_t act_stride, int keep) {
  constexpr int kR = 16 * RT;
  extern __shared__ float smem[];
  float* buf0 = smem;
  float* buf1 = smem + kR * kLD;
  float* pe = smem + 2 * kR * kLD;               // [kR][kPeLD] weighted gamma(x), unscaled
  float* xs = pe + kR * kPeLD;                   // [kR][4]
  float* red = xs + kR * 4;                      // [kWaves][64 lanes][4] partial sums of the last layer
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * kR;
  const int L = a.multires, nf = 1 + 2 * L, d_pe = 3 * nf;
  const int n = a.n_layers;

  ROWS_STAMP();
  if (tid < kR * 3) {
    const int r = tid / 3, c = tid - 3 * r;
    xs[r * 4 + c] = (row0 + r < P) ? x[(row0 + r) * 3 + c] : 0.f;
  }
  __syncthreads();
  // ---- input tile: [gamma(x) | code[frame] | zeros up to the chunk boundary]
  for (int e = tid; e < kR * nf; e += kThreads) {
    const int r = e / nf, f = e - r * nf;
    const float x0 = xs[r * 4], x1 = xs[r * 4 + 1], x2 = xs[r * 4 + 2];
    float v0, v1, v2;
    if (f == 0) {
      v0 = x0; v1 = x1; v2 = x2;
    } else {
      const float freq = (float)(1 << ((f - 1) >> 1));
      const float wt = a.pe_w[f - 1];
      if ((f - 1) & 1) {
        v0 = wt * cosf(x0 * freq); v1 = wt * cosf(x1 * freq); v2 = wt * cosf(x2 * freq);
      } else {
        v0 = wt * sinf(x0 * freq); v1 = wt * sinf(x1 * freq); v2 = wt * sinf(x2 * freq);
      }
    }
    float* o = buf0 + r * kLD + 3 * f;
    o[0] = v0; o[1] = v1; o[2] = v2;
    float* q = pe + r * kPeLD + 3 * f;
    q[0] = v0; q[1] = v1; q[2] = v2;
  }
  {
    const int fill = a.fwd[0].KC * 16 - d_pe;     // code columns + zero padding
    for (int e = tid; e < kR * fill; e += kThreads) {
      const int r = e / fill, c = e - r * fill;
      float v = 0.f;
      if (c < a.cond_dim && row0 + r < P) v = cond[(cond_index ? cond_index[row0 + r] : 0) * ld_cond + c];
      buf0[r * kLD + d_pe + c] = v;
    }
  }
  __syncthreads();
  ROWS_STAMP();
  float* in = buf0;
  float* ob = buf1;
  const float p = a.act_param, inv_p = p != 0.f ? 1.f / p : 0.f;
  for (int l = 0; l + 1 < n; ++l) {
    const bool skip_next = (l + 1 == a.skip_layer);
    run_hidden<RT>(a.fwd[l], in, ob, wave, lane, a.hidden_act, p, inv_p, skip_next ? kInvSqrt2 : 1.f, !skip_next);
    ROWS_STAMP();
    const int width = a.dims[l + 1];              // = N (+ d_pe behind the skip)
    if (skip_next) {
      for (int e = tid; e < kR * d_pe; e += kThreads) {
        const int r = e / d_pe, c = e - r * d_pe;
        ob[r * kLD + a.fwd[l].N + c] = pe[r * kPeLD + c] * kInvSqrt2;
      }
    }
    {
      const int padded = a.fwd[l + 1].KC * 16;     // the next layer reads whole chunks: zero what the tiles did not cover
      const int covered = skip_next ? width : ((a.fwd[l].N + 15) / 16) * 16;
      const int extra = padded - covered;
      for (int e = tid; e < kR * extra; e += kThreads) {
        const int r = e / extra, c = e - r * extra;
        ob[r * kLD + covered + c] = 0.f;
      }
    }
    __syncthreads();
    ROWS_STAMP();
    if (keep) {
      float* dst = acts + (int64_t)l * act_stride + row0 * ld_act;
      const int w4 = (width + 3) / 4;
      for (int e = tid; e < kR * w4; e += kThreads) {
        const int r = e / w4, c = (e - r * w4) * 4;
        *reinterpret_cast<float4*>(dst + (int64_t)r * ld_act + c) = *reinterpret_cast<const float4*>(ob + r * kLD + c);
      }
    }
    ROWS_STAMP();
    float* t = in; in = ob; ob = t;
  }
  // ---- last layer, n_out <= 16 outputs: ONE column tile, its chunks dealt to the waves, partial sums through LDS
  const RowsLayer& Ll = a.fwd[n - 1];
  const int KC = Ll.KC, per = (KC + kWaves - 1) / kWaves;
  const int c0 = wave * per, c1 = (c0 + per < KC) ? c0 + per : KC;
  const f32x4* wpl = reinterpret_cast<const f32x4*>(Ll.Wp) + lane;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow = in + (16 * rt + (lane & 15)) * kLD + 4 * (lane >> 4);
    for (int c = c0; c < c1; ++c) {
      const f32x4 bv = wpl[(int64_t)c * 64];
      const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * c);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
    }
    if (rt) __syncthreads();                       // (the partial sums of the previous row tile have been read)
    *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (wave == 0) {
      const int nn = lane & 15;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 16 * rt + 4 * (lane >> 4) + i;
        float v = red[lane * 4 + i];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += red[(64 * w + lane) * 4 + i];
        if (nn < n_out && row0 + r < P) {
          if (Ll.bias) v += Ll.bias[nn];
          if (a.residual) v += xs[r * 4 + nn];
          out[(row0 + r) * ldo + nn] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ reverse pass
// g <- dZ . W for the wave's column tiles, then the NEXT (earlier) layer's activation gradient applied in the epilogue:
//   dZ_prev[r][c] = g[r][c] * act'(y_scale * y_prev[r][c]) * out_scale   for c < n_act (columns of the activation),
//   park[r][c - n_act] = g[r][c] / sqrt2                                  for n_act <= c < n_act + d_pe (skip connection),
// written to the other LDS buffer (zeros in the tile padding).  y_prev comes from the forward pass's workspace; it is requested
// before the product so that it travels under the MFMAs.
template <int TPW, int RT, int ACT>
__device__ __forceinline__ void reverse_layer(const RowsLayer& L, const float* __restrict__ in, float* __restrict__ outb,
                                              float* __restrict__ park, const float* __restrict__ yprev, int64_t ld_act, int n_act,
                                              int d_park, int wave, int lane, int act, float p, float y_scale, float out_scale) {
  f32x4 acc[RT][TPW];
  float yv[RT][TPW][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (yprev) {                                     // clamped, unconditional: one basic block of loads in front of the products
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int c = (wave * TPW + t) * 16 + (lane & 15);
        const int cc = c < n_act ? c : n_act - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) yv[rt][t][i] = yprev[(int64_t)(16 * rt + 4 * (lane >> 4) + i) * ld_act + cc];
      }
    }
  } else {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) yv[rt][t][i] = 0.f;
  }
  tile_mma<TPW, RT>(in, L.KC, L.Wp, L.KC, wave, lane, acc);
  const bool has_y = yprev != nullptr;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int c = (wave * TPW + t) * 16 + (lane & 15);
    const bool in_act = c < n_act;
    const bool in_park = !in_act && c < n_act + d_park;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 16 * rt + 4 * (lane >> 4) + i;
        const float g = acc[rt][t][i];
        const float d = has_y ? g * act_grad<ACT>(yv[rt][t][i] * y_scale, act, p) * out_scale : g;
        if (in_park) park[r * kPeLD + (c - n_act)] = g * kInvSqrt2;
        if (c < kMaxWidth) outb[r * kLD + c] = in_act ? d : 0.f;
      }
    }
  }
}

template <int RT, int ACT>
__device__ __forceinline__ void run_reverse_act(const RowsLayer& L, const float* in, float* outb, float* park, const float* yprev,
                                                int64_t ld_act, int n_act, int d_park, int wave, int lane, int act, float p,
                                                float y_scale, float out_scale) {
  switch (L.TPW) {
    case 1: reverse_layer<1, RT, ACT>(L, in, outb, park, yprev, ld_act, n_act, d_park, wave, lane, act, p, y_scale, out_scale); break;
    case 2: reverse_layer<2, RT, ACT>(L, in, outb, park, yprev, ld_act, n_act, d_park, wave, lane, act, p, y_scale, out_scale); break;
    case 3: reverse_layer<3, RT, ACT>(L, in, outb, park, yprev, ld_act, n_act, d_park, wave, lane, act, p, y_scale, out_scale); break;
    default: reverse_layer<4, RT, ACT>(L, in, outb, park, yprev, ld_act, n_act, d_park, wave, lane, act, p, y_scale, out_scale); break;
  }
}

template <int RT>
__device__ __forceinline__ void run_reverse(const RowsLayer& L, const float* in, float* outb, float* park, const float* yprev,
                                            int64_t ld_act, int n_act, int d_park, int wave, int lane, int act, float p,
                                            float y_scale, float out_scale) {
  if (act == RECMV_ACT_SOFTPLUS)
    run_reverse_act<RT, RECMV_ACT_SOFTPLUS>(L, in, outb, park, yprev, ld_act, n_act, d_park, wave, lane, act, p, y_scale, out_scale);
  else
    run_reverse_act<RT, -1>(L, in, outb, park, yprev, ld_act, n_act, d_park, wave, lane, act, p, y_scale, out_scale);
}

// gx [P,3] = J(x)^T g_out through the layers in reverse, from the activations mlp_rows_fwd_kernel(keep) left in `acts`.
// g_out NULL: ones on a scalar output (the cotangent of every ray is row 0 of the last weight).
template <int RT>
__global__ __launch_bounds__(kThreads) void mlp_rows_vjp_kernel(RowsArgs a, const float* __restrict__ x, int64_t P, int n_out,
                                                                const float* __restrict__ g_out, int64_t ldg,
                                                                float* __restrict__ gx, const float* __restrict__ acts,
                                                                int64_t ld_act, int64_t act_stride) {
  constexpr int kR = 16 * RT;
  extern __shared__ float smem[];
  float* buf0 = smem;
  float* buf1 = smem + kR * kLD;
  float* park = smem + 2 * kR * kLD;             // [kR][kPeLD] gradient of the encoding that entered through the skip
  float* xs = park + kR * kPeLD;                 // [kR][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * kR;
  const int L = a.multires, d_pe = 3 * (1 + 2 * L);
  const int n = a.n_layers;
  const float p = a.act_param;
  if (tid < kR * 3) {
    const int r = tid / 3, c = tid - 3 * r;
    xs[r * 4 + c] = (row0 + r < P) ? x[(row0 + r) * 3 + c] : 0.f;
  }
  for (int e = tid; e < kR * kPeLD; e += kThreads) park[e] = 0.f;
  const float* yt = acts + row0 * ld_act;        // this tile's rows of every stored activation
  float* in = buf0;
  float* ob = buf1;
  // ---- dZ of the last hidden layer: (cotangent of the last layer's input) (.) act'
  {
    const int l = n - 2;                           // layer whose output feeds the last layer (n >= 2 checked by the host)
    const bool skip_here = (l + 1 == a.skip_layer);
    const int n_act = a.rows[l];
    const float y_scale = skip_here ? kSqrt2 : 1.f, o_scale = skip_here ? kInvSqrt2 : 1.f;
    const float* yprev = yt + (int64_t)l * act_stride;
    if (!g_out) {
      const int width = (a.dims[n - 1] + 15) / 16 * 16;
      __syncthreads();                             // (park has been cleared)
      for (int e = tid; e < kR * width; e += kThreads) {
        const int r = e / width, c = e - r * width;
        float v = 0.f;
        if (c < a.dims[n - 1]) {
          const float g = a.W_last[c];
          if (c < n_act) v = g * act_grad<-1>(yprev[(int64_t)r * ld_act + c] * y_scale, a.hidden_act, p) * o_scale;
          else if (c < n_act + d_pe) park[r * kPeLD + (c - n_act)] = g * kInvSqrt2;
        }
        in[r * kLD + c] = v;
      }
      __syncthreads();
    } else {
      // stage the cotangent tile (zero-padded to one chunk per 16 outputs), multiply by the last weight
      const int kc = a.bwd[n - 1].KC;             // chunks of the packed last layer
      for (int e = tid; e < kR * kc * 16; e += kThreads) {
        const int r = e / (kc * 16), c = e - r * (kc * 16);
        ob[r * kLD + c] = (c < n_out && row0 + r < P) ? g_out[(row0 + r) * ldg + c] : 0.f;
      }
      __syncthreads();
      run_reverse<RT>(a.bwd[n - 1], ob, in, park, yprev, ld_act, n_act, skip_here ? d_pe : 0, wave, lane, a.hidden_act, p, y_scale,
                      o_scale);
      __syncthreads();
    }
  }
  // ---- hidden layers in reverse: in = dZ_l  ->  ob = dZ_{l-1}
  for (int l = n - 2; l >= 1; --l) {
    const bool skip_here = (l == a.skip_layer);   // layer l's input = [act(z_{l-1}) / sqrt2 | gamma / sqrt2]
    const int n_act = a.rows[l - 1];
    const float* yprev = yt + (int64_t)(l - 1) * act_stride;
    run_reverse<RT>(a.bwd[l], in, ob, park, yprev, ld_act, n_act, skip_here ? d_pe : 0, wave, lane, a.hidden_act, p,
                    skip_here ? kSqrt2 : 1.f, skip_here ? kInvSqrt2 : 1.f);
    __syncthreads();
    float* t = in; in = ob; ob = t;
  }
  // ---- first layer: gradient of [gamma(x) | code]; only the encoding's columns flow to x
  run_reverse<RT>(a.bwd[0], in, ob, park, nullptr, ld_act, d_pe, 0, wave, lane, RECMV_ACT_NONE, 0.f, 1.f, 1.f);
  __syncthreads();
  if (tid < kR * 3) {
    const int r = tid / 3, c = tid - 3 * r;
    if (row0 + r < P) {
      const float xc = xs[r * 4 + c];
      const float* gp = ob + r * kLD;
      const float* sp = park + r * kPeLD;
      float acc = gp[c] + sp[c];
      for (int b = 0; b < L; ++b) {
        const float f = (float)(1 << b);
        float s, co;
        sincosf(xc * f, &s, &co);
        const float gs = gp[3 + 6 * b + c] + sp[3 + 6 * b + c], gc = gp[3 + 6 * b + 3 + c] + sp[3 + 6 * b + 3 + c];
        acc += f * (a.pe_w[2 * b] * co * gs - a.pe_w[2 * b + 1] * s * gc);
      }
      if (a.residual) acc += g_out[(row0 + r) * ldg + c];
      gx[(row0 + r) * 3 + c] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------ packing
// out[((t * KC + c) * 64 + l) * 4 + s] = B[n = 16 t + (l & 15)][k = 16 c + 4 (l >> 4) + s], zero outside [N) x [K);
// B[n][k] = W[n * sn + k * sk]: (ldw, 1) packs W for the forward product, (1, ldw) packs W^T for the reverse one.
__global__ __launch_bounds__(kThreads) void mlp_pack_kernel(const float* __restrict__ W, int64_t sn, int64_t sk, int N, int K,
                                                            int KC, int64_t total4, float* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total4; e += (int64_t)gridDim.x * kThreads) {
    const int l = (int)(e & 63);
    const int64_t blk = e >> 6;
    const int c = (int)(blk % KC);
    const int t = (int)(blk / KC);
    const int nn = 16 * t + (l & 15);
    const int k0 = 16 * c + 4 * (l >> 4);
    float4 v;
    v.x = (nn < N && k0 + 0 < K) ? W[nn * sn + (k0 + 0) * sk] : 0.f;
    v.y = (nn < N && k0 + 1 < K) ? W[nn * sn + (k0 + 1) * sk] : 0.f;
    v.z = (nn < N && k0 + 2 < K) ? W[nn * sn + (k0 + 2) * sk] : 0.f;
    v.w = (nn < N && k0 + 3 < K) ? W[nn * sn + (k0 + 3) * sk] : 0.f;
    reinterpret_cast<float4*>(out)[e] = v;
  }
}

inline int tpw_for(int N) {
  const int tiles = (N + 15) / 16;
  const int per = (tiles + kWaves - 1) / kWaves;
  return per < 1 ? 1 : (per > 4 ? 4 : per);
}

struct PackPlan {
  int64_t off_fwd[RECMV_MLP_MAX_LAYERS], off_bwd[RECMV_MLP_MAX_LAYERS];   // float offsets into the packed buffer
  int N_fwd[RECMV_MLP_MAX_LAYERS], K_fwd[RECMV_MLP_MAX_LAYERS], N_bwd[RECMV_MLP_MAX_LAYERS], K_bwd[RECMV_MLP_MAX_LAYERS];
  int64_t floats;
};

inline int64_t packed_floats(int N, int K, bool whole_tiles_only) {
  const int tiles = whole_tiles_only ? (N + 15) / 16 : tpw_for(N) * kWaves;
  return (int64_t)tiles * ((K + 15) / 16) * 256;
}

// forward: layer l maps dims[l] inputs to rows[l] outputs (the last layer is packed tile by tile, it is evaluated one tile at a
// time); reverse: layer l maps rows[l] cotangents to dims[l] input gradients — only the encoding's columns for layer 0.
PackPlan make_plan(const recmv_mlp* m) {
  PackPlan pl;
  int64_t o = 0;
  const int n = m->n_layers, d_pe = 3 + 6 * m->multires;
  for (int l = 0; l < n; ++l) {
    pl.N_fwd[l] = m->rows[l];
    pl.K_fwd[l] = m->dims[l];
    pl.off_fwd[l] = o;
    o += packed_floats(pl.N_fwd[l], pl.K_fwd[l], l == n - 1);
    pl.N_bwd[l] = l == 0 ? d_pe : m->dims[l];
    pl.K_bwd[l] = m->rows[l];
    pl.off_bwd[l] = o;
    o += packed_floats(pl.N_bwd[l], pl.K_bwd[l], false);
  }
  pl.floats = o;
  return pl;
}

int rows_supported(const recmv_mlp* m) {
  if (!m || m->n_layers < 2 || m->n_layers > RECMV_MLP_MAX_LAYERS) return 0;
  if (m->multires < 0 || m->multires > 8) return 0;
  if (m->split_row > 0 && m->W2[0]) return 0;
  for (int l = 0; l <= m->n_layers; ++l)
    if (m->dims[l] > kMaxWidth && l < m->n_layers) return 0;
  for (int l = 0; l + 1 < m->n_layers; ++l)
    if (m->rows[l] > kMaxWidth) return 0;
  if (m->skip_layer == 0 || m->skip_layer >= m->n_layers) return 0;
  return 1;
}

void fill_args(const recmv_mlp* m, const float* packed, RowsArgs* a) {
  const PackPlan pl = make_plan(m);
  const int n = m->n_layers;
  memset(a, 0, sizeof(*a));
  for (int l = 0; l < n; ++l) {
    a->fwd[l].Wp = packed + pl.off_fwd[l];
    a->fwd[l].bias = m->bias[l];
    a->fwd[l].N = pl.N_fwd[l];
    a->fwd[l].KC = (pl.K_fwd[l] + 15) / 16;
    a->fwd[l].TPW = tpw_for(pl.N_fwd[l]);
    a->bwd[l].Wp = packed + pl.off_bwd[l];
    a->bwd[l].bias = nullptr;
    a->bwd[l].N = pl.N_bwd[l];
    a->bwd[l].KC = (pl.K_bwd[l] + 15) / 16;
    a->bwd[l].TPW = tpw_for(pl.N_bwd[l]);
    a->rows[l] = m->rows[l];
  }
  for (int l = 0; l <= n; ++l) a->dims[l] = m->dims[l];
  a->W_last = m->W[n - 1];
  a->n_layers = n;
  a->multires = m->multires;
  a->cond_dim = m->cond_dim;
  a->skip_layer = m->skip_layer;
  a->hidden_act = m->hidden_act;
  a->residual = m->residual;
  a->act_param = m->act_param;
  for (int i = 0; i < 32; ++i) a->pe_w[i] = m->pe_weights[i];
}

inline size_t fwd_lds(int rt) { return (size_t)(2 * 16 * rt * kLD + 16 * rt * kPeLD + 16 * rt * 4 + kWaves * 64 * 4) * sizeof(float); }
inline size_t vjp_lds(int rt) { return (size_t)(2 * 16 * rt * kLD + 16 * rt * kPeLD + 16 * rt * 4) * sizeof(float); }

inline int64_t pad16(int64_t v) { return (v + 15) / 16 * 16; }
inline int64_t pad32(int64_t v) { return (v + 31) / 32 * 32; }

// Row tiles per workgroup: 16 rows while one round of workgroups covers the rows (lowest latency), 32 rows past that — or whatever
// recmv_set_mlp_rows_tile says (2 = every weight byte from L2 feeds twice the FLOP and a pass takes half the CUs).
int g_rows_rt = 0;       // 0 = by row count
inline int pick_rt(int64_t P) {
  if (g_rows_rt == 1 || g_rows_rt == 2) return g_rows_rt;
  return P <= (int64_t)16 * kNumCU ? 1 : 2;
}

struct RowsLayout {
  int64_t ld_act, act_stride, bytes;
};

RowsLayout rows_layout(const recmv_mlp* m, int64_t P) {
  RowsLayout L;
  int64_t maxw = 16;
  for (int l = 1; l < m->n_layers; ++l) maxw = m->dims[l] > maxw ? m->dims[l] : maxw;
  L.ld_act = pad16(maxw);
  L.act_stride = pad32(P) * L.ld_act;
  L.bytes = (int64_t)(m->n_layers - 1) * L.act_stride * 4;
  return L;
}

}  // namespace
}  // namespace recmv

using namespace recmv;

#ifdef RECMV_ROWS_TIMING
extern "C" int recmv_debug_rows_clock(long long* out, int cap, int reset) {
  int n = 0;
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_clk_n), sizeof(int));
  if (n > cap) n = cap;
  if (n > 0) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk), sizeof(long long) * n);
  if (reset) {
    const int zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_clk_n), &zero, sizeof(int));
  }
  return n;
}
#endif

extern "C" int recmv_mlp_rows_supported(const recmv_mlp* m) { return rows_supported(m); }

extern "C" int recmv_set_mlp_rows_tile(int row_tiles) {
  RECMV_REQUIRE(row_tiles >= 0 && row_tiles <= 2, "set_mlp_rows_tile: 0 (by row count), 1 (16 rows) or 2 (32 rows)");
  g_rows_rt = row_tiles;
  return RECMV_OK;
}

extern "C" int64_t recmv_mlp_pack_bytes(const recmv_mlp* m) {
  if (!rows_supported(m)) return 0;
  return make_plan(m).floats * 4;
}

extern "C" int recmv_mlp_pack(const recmv_mlp* m, void* packed, int64_t packed_bytes, void* stream) {
  RECMV_REQUIRE(rows_supported(m), "mlp_pack: this net does not fit the row-tile kernels");
  const PackPlan pl = make_plan(m);
  RECMV_REQUIRE(packed && packed_bytes >= pl.floats * 4, "mlp_pack: buffer %lld < %lld bytes", (long long)packed_bytes,
                (long long)(pl.floats * 4));
  float* out = (float*)packed;
  const int n = m->n_layers;
  for (int l = 0; l < n; ++l) {
    RECMV_REQUIRE(m->W[l], "mlp_pack: layer %d has no weight", l);
    const int ldw = m->dims[l];
    {
      const int KC = (pl.K_fwd[l] + 15) / 16;
      const int64_t total4 = packed_floats(pl.N_fwd[l], pl.K_fwd[l], l == n - 1) / 4;
      hipLaunchKernelGGL(mlp_pack_kernel, dim3(stream_grid(total4, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, m->W[l],
                         (int64_t)ldw, (int64_t)1, pl.N_fwd[l], pl.K_fwd[l], KC, total4, out + pl.off_fwd[l]);
    }
    {
      const int KC = (pl.K_bwd[l] + 15) / 16;
      const int64_t total4 = packed_floats(pl.N_bwd[l], pl.K_bwd[l], false) / 4;
      hipLaunchKernelGGL(mlp_pack_kernel, dim3(stream_grid(total4, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, m->W[l],
                         (int64_t)1, (int64_t)ldw, pl.N_bwd[l], pl.K_bwd[l], KC, total4, out + pl.off_bwd[l]);
    }
  }
  return check_launch("mlp_pack");
}

extern "C" int64_t recmv_mlp_rows_workspace_bytes(const recmv_mlp* m, int64_t P) {
  if (!rows_supported(m) || P <= 0) return 0;
  return rows_layout(m, P).bytes;
}

extern "C" int recmv_mlp_rows_forward(const recmv_mlp* m, const void* packed, const float* x, const float* cond, int64_t ld_cond,
                                      const int64_t* cond_index, int64_t P, int n_out, float* out, int64_t ldo, void* workspace,
                                      int64_t workspace_bytes, int keep, void* stream) {
  RECMV_REQUIRE(rows_supported(m), "mlp_rows_forward: this net does not fit the row-tile kernels");
  RECMV_REQUIRE(P >= 0, "mlp_rows_forward: negative P");
  if (P == 0) return RECMV_OK;
  RECMV_REQUIRE(packed && x && out, "mlp_rows_forward: NULL pointer");
  RECMV_REQUIRE(n_out >= 1 && n_out <= 16 && n_out <= m->rows[m->n_layers - 1] && ldo >= n_out, "mlp_rows_forward: bad n_out %d", n_out);
  RECMV_REQUIRE(m->cond_dim == 0 || cond, "mlp_rows_forward: the net takes a per-frame code but cond is NULL");
  RECMV_REQUIRE(!m->residual || n_out == 3, "mlp_rows_forward: residual nets are 3-d");
  const RowsLayout L = rows_layout(m, P);
  if (keep) {
    RECMV_REQUIRE(workspace, "mlp_rows_forward: keep needs a workspace");
    if (workspace_bytes < L.bytes) {
      set_error("mlp_rows_forward: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.bytes);
      return RECMV_ERR_WORKSPACE;
    }
  }
  RowsArgs a;
  fill_args(m, (const float*)packed, &a);
  static bool attr_set = false;
  if (!attr_set) {
    RECMV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_rows_fwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)fwd_lds(1)));
    RECMV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_rows_fwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)fwd_lds(2)));
    attr_set = true;
  }
  if (pick_rt(P) == 1)
    hipLaunchKernelGGL(mlp_rows_fwd_kernel<1>, dim3((unsigned)ceil_div(P, 16)), dim3(kThreads), fwd_lds(1), (hipStream_t)stream, a, x,
                       cond, ld_cond, cond_index, P, n_out, out, ldo, (float*)workspace, L.ld_act, L.act_stride, keep);
  else
    hipLaunchKernelGGL(mlp_rows_fwd_kernel<2>, dim3((unsigned)ceil_div(P, 32)), dim3(kThreads), fwd_lds(2), (hipStream_t)stream, a, x,
                       cond, ld_cond, cond_index, P, n_out, out, ldo, (float*)workspace, L.ld_act, L.act_stride, keep);
  return check_launch("mlp_rows_forward");
}

extern "C" int recmv_mlp_rows_vjp_input(const recmv_mlp* m, const void* packed, const float* x, int64_t P, int n_out,
                                        const float* g_out, int64_t ldg, float* gx, const void* workspace, int64_t workspace_bytes,
                                        void* stream) {
  RECMV_REQUIRE(rows_supported(m), "mlp_rows_vjp_input: this net does not fit the row-tile kernels");
  RECMV_REQUIRE(P >= 0, "mlp_rows_vjp_input: negative P");
  if (P == 0) return RECMV_OK;
  RECMV_REQUIRE(packed && x && gx && workspace, "mlp_rows_vjp_input: NULL pointer");
  RECMV_REQUIRE(n_out >= 1 && n_out <= 16 && n_out <= m->rows[m->n_layers - 1], "mlp_rows_vjp_input: bad n_out %d", n_out);
  RECMV_REQUIRE(g_out || n_out == 1, "mlp_rows_vjp_input: a NULL cotangent means ones and needs n_out == 1");
  RECMV_REQUIRE(!m->residual || (g_out && n_out == 3), "mlp_rows_vjp_input: residual nets need their 3-d cotangent");
  const RowsLayout L = rows_layout(m, P);
  if (workspace_bytes < L.bytes) {
    set_error("mlp_rows_vjp_input: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.bytes);
    return RECMV_ERR_WORKSPACE;
  }
  RowsArgs a;
  fill_args(m, (const float*)packed, &a);
  static bool attr_set = false;
  if (!attr_set) {
    RECMV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_rows_vjp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)vjp_lds(1)));
    RECMV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_rows_vjp_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)vjp_lds(2)));
    attr_set = true;
  }
  if (pick_rt(P) == 1)
    hipLaunchKernelGGL(mlp_rows_vjp_kernel<1>, dim3((unsigned)ceil_div(P, 16)), dim3(kThreads), vjp_lds(1), (hipStream_t)stream, a, x, P,
                       n_out, g_out, ldg, gx, (const float*)workspace, L.ld_act, L.act_stride);
  else
    hipLaunchKernelGGL(mlp_rows_vjp_kernel<2>, dim3((unsigned)ceil_div(P, 32)), dim3(kThreads), vjp_lds(2), (hipStream_t)stream, a, x, P,
                       n_out, g_out, ldg, gx, (const float*)workspace, L.ld_act, L.act_stride);
  return check_launch("mlp_rows_vjp_input");
}
