// Whole-MLP launch chains for the graph-free passes of the hot path — gfx950.
//
// The surface root finder (utils/FindSurfacePs.py:273-353 of the reference) evaluates, up to 20 times per
// iteration and garment, the SDF net with its input gradient (model/network.py:98-133) and the offset MLP of the
// deformer with a vector-Jacobian product to its input (model/Deformer.py:141-206) on a few thousand rays.  Each
// of those is ~30 kernels of 5-20 us; issued one by one from Python they are host-bound.  Here ONE C call
// enqueues the whole chain on the caller's stream:
//
//   recmv_mlp_forward   : x [P,3] -> gamma(x) (+ per-frame code gathered by frame id) -> L fused MFMA layers
//                         (recmv_gemm_nt: GEMM + bias + activation [+ 1/sqrt(2) skip scale])  -> out
//   recmv_mlp_vjp_input : J(x)^T g through the same layers in reverse (activation-gradient kernel + MFMA product
//                         with the cached W^T per layer, skip split, positional-encoding VJP), no parameter grads.
//
// Activations live in a caller-provided workspace (recmv_mlp_workspace_bytes); nothing is allocated here.
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kSqrt2 = 1.41421356237309504880f;

__device__ __forceinline__ float dact(float y, int act, float p) {
  switch (act) {
    case RECMV_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case RECMV_ACT_SOFTPLUS: return -expm1f(-p * y);
    case RECMV_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// out[r,c] = out_scale * gy[r*ldg + c] * act'(z) with y = act(z) = y_scale * ybuf[r*ldy + c]   (ldg may be 0:
// one cotangent row broadcast to every point)
__global__ __launch_bounds__(kBlk) void act_grad_2d_kernel(const float* __restrict__ gy, int64_t ldg,
                                                           const float* __restrict__ y, int64_t ldy,
                                                           float* __restrict__ out, int64_t ldo, int64_t rows,
                                                           int cols, int act, float p, float y_scale,
                                                           float out_scale) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlk) {
    const int64_t r = e / cols;
    const int c = (int)(e - r * cols);
    out[r * ldo + c] = out_scale * gy[r * ldg + c] * dact(y[r * ldy + c] * y_scale, act, p);
  }
}

// the same on float4 columns (cols4 = cols / 4; strides and pointers 16-byte aligned)
__global__ __launch_bounds__(kBlk) void act_grad_2d_vec4_kernel(const float* __restrict__ gy, int64_t ldg,
                                                                const float* __restrict__ y, int64_t ldy,
                                                                float* __restrict__ out, int64_t ldo, int64_t rows,
                                                                int cols4, int act, float p, float y_scale,
                                                                float out_scale) {
  const int64_t total = rows * cols4;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlk) {
    const int64_t r = e / cols4;
    const int c = (int)(e - r * cols4) * 4;
    const float4 g = *reinterpret_cast<const float4*>(gy + r * ldg + c);
    const float4 v = *reinterpret_cast<const float4*>(y + r * ldy + c);
    float4 o;
    o.x = out_scale * g.x * dact(v.x * y_scale, act, p);
    o.y = out_scale * g.y * dact(v.y * y_scale, act, p);
    o.z = out_scale * g.z * dact(v.z * y_scale, act, p);
    o.w = out_scale * g.w * dact(v.w * y_scale, act, p);
    *reinterpret_cast<float4*>(out + r * ldo + c) = o;
  }
}

// out[r,c] = a[r*lda + c] + s * b[r*ldb + c]
__global__ __launch_bounds__(kBlk) void add_scaled_2d_kernel(const float* __restrict__ a, int64_t lda,
                                                             const float* __restrict__ b, int64_t ldb, float s,
                                                             float* __restrict__ out, int64_t ldo, int64_t rows,
                                                             int cols) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlk) {
    const int64_t r = e / cols;
    const int c = (int)(e - r * cols);
    out[r * ldo + c] = a[r * lda + c] + s * b[r * ldb + c];
  }
}

// out[r, 0:cols] = table[index[r], 0:cols]; columns [cols, fill) of out are zeroed
__global__ __launch_bounds__(kBlk) void gather_rows_kernel(const float* __restrict__ table, int64_t ldt,
                                                           const int64_t* __restrict__ index,
                                                           float* __restrict__ out, int64_t ldo, int64_t rows,
                                                           int cols, int fill) {
  const int64_t total = rows * fill;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlk) {
    const int64_t r = e / fill;
    const int c = (int)(e - r * fill);
    out[r * ldo + c] = c < cols ? table[(index ? index[r] : 0) * ldt + c] : 0.f;
  }
}

inline int64_t pad4(int64_t v) { return (v + 3) / 4 * 4; }

struct Layout {
  int64_t ld_in;          // row stride of the input buffer
  int64_t ld_act;         // row stride of every activation / gradient buffer
  int64_t off_in, off_act[RECMV_MLP_MAX_LAYERS], off_g[3];
  int64_t bytes;
};

Layout make_layout(const recmv_mlp* m, int64_t P, int keep) {
  Layout L;
  int64_t maxw = 4;
  for (int l = 0; l <= m->n_layers; ++l) maxw = m->dims[l] > maxw ? m->dims[l] : maxw;
  L.ld_in = pad4(m->dims[0]);
  L.ld_act = pad4(maxw);
  int64_t o = 0;
  auto take = [&](int64_t floats) {
    int64_t r = o;
    o += (floats * 4 + 255) / 256 * 256;
    return r;
  };
  L.off_in = take(P * L.ld_in);
  const int nact = m->n_layers - 1;
  for (int l = 0; l < nact; ++l) L.off_act[l] = (keep || l < 2) ? take(P * L.ld_act) : L.off_act[l & 1];
  L.off_g[0] = keep ? take(P * L.ld_act) : 0;
  L.off_g[1] = keep ? take(P * L.ld_act) : 0;
  L.off_g[2] = keep ? take(P * L.ld_act) : 0;      // dZ of the layer being differentiated
  L.bytes = o;
  return L;
}

int check_desc(const recmv_mlp* m) {
  RECMV_REQUIRE(m, "mlp: NULL descriptor");
  RECMV_REQUIRE(m->n_layers >= 1 && m->n_layers <= RECMV_MLP_MAX_LAYERS, "mlp: bad layer count %d", m->n_layers);
  RECMV_REQUIRE(m->multires >= 0 && m->multires <= 16 && m->cond_dim >= 0, "mlp: bad encoding");
  RECMV_REQUIRE(m->dims[0] == 3 + 6 * m->multires + m->cond_dim, "mlp: dims[0] != 3+6L+cond_dim");
  for (int l = 0; l < m->n_layers; ++l) {
    RECMV_REQUIRE(m->W[l] && m->rows[l] > 0 && m->dims[l] > 0, "mlp: layer %d incomplete", l);
    const int expect = (l + 1 == m->skip_layer) ? m->dims[l + 1] - (3 + 6 * m->multires) : m->dims[l + 1];
    RECMV_REQUIRE(m->rows[l] == expect, "mlp: layer %d has %d rows, expected %d", l, m->rows[l], expect);
  }
  RECMV_REQUIRE(m->skip_layer < m->n_layers, "mlp: bad skip layer");
  if (m->split_row > 0 && m->W2[0]) {
    RECMV_REQUIRE(m->split_row % 128 == 0, "mlp: split_row %lld is not a multiple of 128", (long long)m->split_row);
    for (int l = 0; l < m->n_layers; ++l)
      RECMV_REQUIRE(m->W2[l] && (!m->bias[l] == !m->bias2[l]), "mlp: the second net's layer %d is incomplete", l);
  }
  return RECMV_OK;
}

// rows [split, P) of this call belong to the second net (0: there is none)
inline int64_t second_net_from(const recmv_mlp* m, int64_t P) {
  return (m->split_row > 0 && m->W2[0] && m->split_row < P) ? m->split_row : 0;
}

}  // namespace
}  // namespace recmv

using namespace recmv;

#define RECMV_TRY(expr)            \
  do {                             \
    int rc__ = (expr);             \
    if (rc__ != RECMV_OK) return rc__; \
  } while (0)

extern "C" int recmv_act_grad_2d(const float* gy, int64_t ldg, const float* y, int64_t ldy, float* out, int64_t ldo,
                                 int64_t rows, int64_t cols, int act, float act_param, float y_scale,
                                 float out_scale, void* stream) {
  RECMV_REQUIRE(rows >= 0 && cols >= 0 && cols < (1 << 30), "act_grad_2d: bad size");
  if (rows == 0 || cols == 0) return RECMV_OK;
  RECMV_REQUIRE(gy && y && out, "act_grad_2d: NULL pointer");
  const bool vec = cols % 4 == 0 && ldg % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (vec) {
    hipLaunchKernelGGL(act_grad_2d_vec4_kernel, dim3(stream_grid(rows * (cols / 4), kBlk)), dim3(kBlk), 0,
                       (hipStream_t)stream, gy, ldg, y, ldy, out, ldo, rows, (int)(cols / 4), act, act_param, y_scale,
                       out_scale);
    return check_launch("act_grad_2d(vec4)");
  }
  hipLaunchKernelGGL(act_grad_2d_kernel, dim3(stream_grid(rows * cols, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, gy,
                     ldg, y, ldy, out, ldo, rows, (int)cols, act, act_param, y_scale, out_scale);
  return check_launch("act_grad_2d");
}

extern "C" int recmv_add_scaled_2d(const float* a, int64_t lda, const float* b, int64_t ldb, float s, float* out,
                                   int64_t ldo, int64_t rows, int64_t cols, void* stream) {
  RECMV_REQUIRE(rows >= 0 && cols >= 0 && cols < (1 << 30), "add_scaled_2d: bad size");
  if (rows == 0 || cols == 0) return RECMV_OK;
  RECMV_REQUIRE(a && b && out, "add_scaled_2d: NULL pointer");
  hipLaunchKernelGGL(add_scaled_2d_kernel, dim3(stream_grid(rows * cols, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, a,
                     lda, b, ldb, s, out, ldo, rows, (int)cols);
  return check_launch("add_scaled_2d");
}

extern "C" int recmv_gather_rows(const float* table, int64_t ldt, const int64_t* index, float* out, int64_t ldo,
                                 int64_t rows, int64_t cols, int64_t fill, void* stream) {
  RECMV_REQUIRE(rows >= 0 && cols >= 0 && fill >= cols && fill < (1 << 30), "gather_rows: bad size");
  if (rows == 0 || fill == 0) return RECMV_OK;
  RECMV_REQUIRE(out && (cols == 0 || table), "gather_rows: NULL pointer");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(stream_grid(rows * fill, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, table,
                     ldt, index, out, ldo, rows, (int)cols, (int)fill);
  return check_launch("gather_rows");
}

extern "C" int64_t recmv_mlp_workspace_bytes(const recmv_mlp* m, int64_t P, int keep) {
  if (!m || P <= 0 || m->n_layers < 1 || m->n_layers > RECMV_MLP_MAX_LAYERS) return 0;
  return make_layout(m, P, keep).bytes;
}

extern "C" int recmv_mlp_forward(const recmv_mlp* m, const float* x, const float* cond, int64_t ld_cond,
                                 const int64_t* cond_index, int64_t P, int n_out, float* out, int64_t ldo,
                                 void* workspace, int64_t workspace_bytes, int keep, void* stream) {
  RECMV_TRY(check_desc(m));
  RECMV_REQUIRE(P >= 0, "mlp_forward: negative P");
  if (P == 0) return RECMV_OK;
  const int n = m->n_layers;
  RECMV_REQUIRE(x && out && workspace, "mlp_forward: NULL pointer");
  RECMV_REQUIRE(n_out >= 1 && n_out <= m->rows[n - 1] && ldo >= n_out, "mlp_forward: bad n_out");
  RECMV_REQUIRE(m->cond_dim == 0 || cond, "mlp_forward: the net takes a per-frame code but cond is NULL");
  RECMV_REQUIRE(!m->residual || n_out == 3, "mlp_forward: residual nets are 3-d");
  const Layout L = make_layout(m, P, keep);
  if (workspace_bytes < L.bytes) {
    set_error("mlp_forward: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.bytes);
    return RECMV_ERR_WORKSPACE;
  }
  float* base = (float*)workspace;
  float* in = base + L.off_in / 4;
  const int d_pe = 3 + 6 * m->multires;
  // input = [gamma(x) | code[frame] | zero pad]
  RECMV_TRY(recmv_posenc_forward(x, 3, in, L.ld_in, m->cond_dim ? d_pe : (int)L.ld_in, P, m->multires, m->pe_weights,
                                 1.f, stream));
  if (m->cond_dim) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(stream_grid(P * (L.ld_in - d_pe), kBlk)), dim3(kBlk), 0,
                       (hipStream_t)stream, cond, ld_cond, cond_index, in + d_pe, L.ld_in, P, m->cond_dim,
                       (int)(L.ld_in - d_pe));
    RECMV_TRY(check_launch("mlp_forward/gather"));
  }
  const float* h = in;
  int64_t ldh = L.ld_in;
  const int64_t split = second_net_from(m, P);
  for (int l = 0; l < n; ++l) {
    const bool last = l == n - 1;
    const float* W2 = split ? m->W2[l] : nullptr;
    const float* b2 = split ? m->bias2[l] : nullptr;
    if (last) {
      RECMV_TRY(recmv_gemm_nt_seg(h, ldh, m->W[l], m->dims[l], m->bias[l], W2, b2, split, out, ldo, P, n_out, m->dims[l],
                                  RECMV_ACT_NONE, 0.f, 1.f, stream));
      if (m->residual) RECMV_TRY(recmv_add_scaled_2d(out, ldo, x, 3, 1.f, out, ldo, P, 3, stream));
      break;
    }
    float* y = base + L.off_act[l] / 4;
    const bool skip_next = l + 1 == m->skip_layer;
    RECMV_TRY(recmv_gemm_nt_seg(h, ldh, m->W[l], m->dims[l], m->bias[l], W2, b2, split, y, L.ld_act, P, m->rows[l], m->dims[l],
                                m->hidden_act, m->act_param, skip_next ? kInvSqrt2 : 1.f, stream));
    if (skip_next)
      RECMV_TRY(recmv_posenc_forward(x, 3, y + m->rows[l], L.ld_act, d_pe, P, m->multires, m->pe_weights, kInvSqrt2,
                                     stream));
    h = y;
    ldh = L.ld_act;
  }
  return RECMV_OK;
}

extern "C" int recmv_mlp_vjp_input(const recmv_mlp* m, const float* x, int64_t P, int n_out, const float* g_out,
                                   int64_t ldg, float* gx, void* workspace, int64_t workspace_bytes, void* stream) {
  RECMV_TRY(check_desc(m));
  RECMV_REQUIRE(P >= 0, "mlp_vjp_input: negative P");
  if (P == 0) return RECMV_OK;
  const int n = m->n_layers;
  RECMV_REQUIRE(x && gx && workspace, "mlp_vjp_input: NULL pointer");
  RECMV_REQUIRE(n_out >= 1 && n_out <= m->rows[n - 1], "mlp_vjp_input: bad n_out");
  RECMV_REQUIRE(g_out || n_out == 1, "mlp_vjp_input: a NULL cotangent means ones and needs n_out == 1");
  for (int l = 0; l < n; ++l) RECMV_REQUIRE(m->Wt[l], "mlp_vjp_input: layer %d has no transposed weight", l);
  const int64_t split = second_net_from(m, P);
  if (split)
    for (int l = 0; l < n; ++l) RECMV_REQUIRE(m->Wt2[l], "mlp_vjp_input: the second net's layer %d has no transposed weight", l);
  const Layout L = make_layout(m, P, 1);
  if (workspace_bytes < L.bytes) {
    set_error("mlp_vjp_input: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.bytes);
    return RECMV_ERR_WORKSPACE;
  }
  float* base = (float*)workspace;
  float* gbuf[2] = {base + L.off_g[0] / 4, base + L.off_g[1] / 4};
  const int d_pe = 3 + 6 * m->multires;
  // g = cotangent of the last layer's output -> gradient wrt its input
  const float* g;      // gradient wrt the OUTPUT of layer l-1 (= input of layer l), [P, dims[l]]
  int64_t ld;
  int cur = 0;
  const float* skip_g = nullptr;   // gradient wrt the encoded input that entered through the skip connection
  int64_t skip_ld = 0;
  if (!g_out) {
    g = m->W[n - 1];               // d out_0 / d h = row 0 of the last weight, the same for every point
    ld = 0;
  } else {
    // [P,n_out] x [n_out, dims] : Wt[n-1] is [dims[n-1], rows[n-1]] row-major, use its first n_out columns
    RECMV_TRY(recmv_gemm_nt_seg(g_out, ldg, m->Wt[n - 1], m->rows[n - 1], nullptr, split ? m->Wt2[n - 1] : nullptr, nullptr, split,
                                gbuf[cur], L.ld_act, P, m->dims[n - 1], n_out, RECMV_ACT_NONE, 0.f, 1.f, stream));
    g = gbuf[cur];
    ld = L.ld_act;
    cur ^= 1;
  }
  bool have_dz = false;            // g already is dZ of the layer about to be differentiated (fused into the product before)
  for (int l = n - 2; l >= 0; --l) {
    const float* y = base + L.off_act[l] / 4;
    const bool skip_next = l + 1 == m->skip_layer;
    float* gin = gbuf[cur];
    RECMV_REQUIRE(!(split && ld == 0 && skip_next), "mlp_vjp_input: two nets with the skip connection on the last layer");
    if (skip_next) {
      // y = [act(z)/sqrt2 | gamma/sqrt2]: left part through the activation, right part to the encoding.  The
      // right part is parked FIRST (the product below overwrites the buffer it may live in two steps later).
      float* park = base + L.off_in / 4;
      RECMV_TRY(recmv_add_scaled_2d(g + m->rows[l], ld, g + m->rows[l], ld, kInvSqrt2 - 1.f, park, L.ld_in, P, d_pe,
                                    stream));
      skip_g = park;
      skip_ld = L.ld_in;
    }
    // dZ = g (.) act'(z) once per element, then gin = dZ W.  (Fusing the activation gradient into the product's operand
    // staging, recmv_gemm_nt_actgrad, recomputes it in every column tile — 16x with the 64x32 tiles these row counts
    // get.)  Where the next layer's input is this product's output as it stands, ITS activation gradient is applied in
    // this product's epilogue (recmv_gemm_nt_mulgrad): one launch per layer.
    const float* dzp = g;
    int64_t lddz = ld;
    if (!have_dz) {
      float* dz = base + L.off_g[2] / 4;
      if (split && ld == 0) {
        // the broadcast cotangent row (row 0 of the last weight) differs between the two nets: one pass per row range
        RECMV_TRY(recmv_act_grad_2d(g, 0, y, L.ld_act, dz, L.ld_act, split, m->rows[l], m->hidden_act, m->act_param,
                                    skip_next ? kSqrt2 : 1.f, skip_next ? kInvSqrt2 : 1.f, stream));
        RECMV_TRY(recmv_act_grad_2d(m->W2[n - 1], 0, y + split * L.ld_act, L.ld_act, dz + split * L.ld_act, L.ld_act, P - split,
                                    m->rows[l], m->hidden_act, m->act_param, skip_next ? kSqrt2 : 1.f,
                                    skip_next ? kInvSqrt2 : 1.f, stream));
      } else {
        RECMV_TRY(recmv_act_grad_2d(g, ld, y, L.ld_act, dz, L.ld_act, P, m->rows[l], m->hidden_act, m->act_param,
                                    skip_next ? kSqrt2 : 1.f, skip_next ? kInvSqrt2 : 1.f, stream));
      }
      dzp = dz;
      lddz = L.ld_act;
    }
    const bool fuse = l >= 1 && l != m->skip_layer;       // layer l-1's output is layer l's whole input
    if (fuse) {
      RECMV_TRY(recmv_gemm_nt_mulgrad_seg(dzp, lddz, m->Wt[l], split ? m->Wt2[l] : nullptr, split, m->rows[l], gin, L.ld_act, P,
                                          m->dims[l], m->rows[l], base + L.off_act[l - 1] / 4, L.ld_act, m->hidden_act,
                                          m->act_param, 1.f, 1.f, stream));
    } else {
      RECMV_TRY(recmv_gemm_nt_seg(dzp, lddz, m->Wt[l], m->rows[l], nullptr, split ? m->Wt2[l] : nullptr, nullptr, split, gin,
                                  L.ld_act, P, m->dims[l], m->rows[l], RECMV_ACT_NONE, 0.f, 1.f, stream));
    }
    have_dz = fuse;
    cur ^= 1;
    g = gin;
    ld = L.ld_act;
  }
  // g: gradient wrt [gamma(x) | code]; only the encoding part flows to x
  const float* gpe = g;
  int64_t ldpe = ld;
  if (n == 1) {
    RECMV_REQUIRE(ld != 0, "mlp_vjp_input: single-layer nets need an explicit cotangent");
  }
  if (skip_g) {
    float* sum = gbuf[cur];
    RECMV_TRY(recmv_add_scaled_2d(g, ld, skip_g, skip_ld, 1.f, sum, L.ld_act, P, d_pe, stream));
    gpe = sum;
    ldpe = L.ld_act;
  }
  RECMV_TRY(recmv_posenc_vjp(x, 3, gpe, ldpe, nullptr, 0, gx, P, m->multires, m->pe_weights, stream));
  if (m->residual) RECMV_TRY(recmv_add_scaled_2d(gx, 3, g_out, ldg, 1.f, gx, 3, P, 3, stream));
  return RECMV_OK;
}
