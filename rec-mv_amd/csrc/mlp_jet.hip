// MLP "jet" pass: value AND input-Jacobian in one forward sweep, with an explicit first-order reverse — gfx950.
//
// The loss of the hot path differentiates input derivatives of its MLPs: the eikonal term and the surface normal
// use grad_x SDF(x) (model/network.py:121-133 with create_graph=True, OptimGarmentNetwork.py:1108-1119,
// :1169-1172) and the deformation regulariser / cardinal rays / normal loss use the 3x3 Jacobian of the offset MLP
// (utils/utils.py:133-156 with create_graph=True, OptimGarmentNetwork.py:1135-1155, :1176, :1191-1217).  The
// reference gets the parameter gradients of those terms by autograd's double backward (hundreds of small launches
// per term).  Here the derivative is carried FORWARD as three tangent rows per point,
//
//   layer:  z = h W^T + b,  u_k = t_k W^T (k = x,y,z);   y = phi(z),  s_k = phi'(z) u_k
//   input:  h = [gamma(x) | code],  t_k = [d gamma / d x_k | 0];   skip layer: [y | gamma]/sqrt2, [s_k | t_k]/sqrt2
//
// as ONE MFMA product over the stacked rows [h; t_x; t_y; t_z] (4P x K) per layer plus one element-wise kernel,
// and the reverse sweep is plain first-order arithmetic:
//
//   zbar = ybar phi'(z) + phi''(z) sum_k sbar_k u_k,   ubar_k = sbar_k phi'(z)
//   Wbar = [zbar; ubar]^T [h; t]   (one split-K MFMA product over 4P rows),  bbar = colsum(zbar)
//   [hbar; tbar] = [zbar; ubar] W
//
// down to the encoding (second-derivative kernel of gamma for the tangent rows).  Two C calls per term.
#include "common.h"

namespace recmv {
namespace {

constexpr int kBlk = 256;
constexpr float kInvSqrt2 = 0.70710678118654752440f;

struct ActD {
  float f, d1, d2;
};

// phi, phi', phi'' at z.  Softplus follows torch (threshold 20 on beta*z).
__device__ __forceinline__ ActD act_jet(float z, int act, float p) {
  ActD r;
  switch (act) {
    case RECMV_ACT_RELU:
      r.f = z > 0.f ? z : 0.f;
      r.d1 = z > 0.f ? 1.f : 0.f;
      r.d2 = 0.f;
      break;
    case RECMV_ACT_SOFTPLUS: {
      const float zb = z * p;
      if (zb > 20.f) {
        r.f = z;
        r.d1 = 1.f;
        r.d2 = 0.f;
      } else {
        const float e = expf(-fabsf(zb));                  // in (0,1]
        const float sig = zb >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        r.f = (fmaxf(zb, 0.f) + log1pf(e)) / p;
        r.d1 = sig;
        r.d2 = p * sig * (1.f - sig);
      }
      break;
    }
    case RECMV_ACT_TANH: {
      const float t = tanhf(z);
      r.f = t;
      r.d1 = 1.f - t * t;
      r.d2 = -2.f * t * r.d1;
      break;
    }
    default:
      r.f = z;
      r.d1 = 1.f;
      r.d2 = 0.f;
  }
  return r;
}

// Z: [4P, ldz] pre-activations (value rows WITHOUT bias on entry; the biased z is written back), Y: [4P, ldy].
__global__ __launch_bounds__(kBlk) void jet_act_forward_kernel(float* __restrict__ Z, int64_t ldz,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ Y, int64_t ldy, int64_t P, int N,
                                                               int act, float p, float scale) {
  const int64_t total = P * N;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlk) {
    const int64_t r = e / N;
    const int c = (int)(e - r * N);
    const float z = Z[r * ldz + c] + (bias ? bias[c] : 0.f);
    Z[r * ldz + c] = z;
    const ActD a = act_jet(z, act, p);
    Y[r * ldy + c] = a.f * scale;
#pragma unroll
    for (int k = 1; k <= 3; ++k) Y[(k * P + r) * ldy + c] = a.d1 * Z[(k * P + r) * ldz + c] * scale;
  }
}

// Ybar: [4P, ldyb] cotangents of [y; s_k] (times `scale`), Z as saved by the forward; Zbar: [4P, ldzb] (may alias Ybar).
__global__ __launch_bounds__(kBlk) void jet_act_backward_kernel(const float* Ybar, int64_t ldyb,
                                                                const float* __restrict__ Z, int64_t ldz,
                                                                float* Zbar, int64_t ldzb, int64_t P, int N, int act,
                                                                float p, float scale) {
  const int64_t total = P * N;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlk) {
    const int64_t r = e / N;
    const int c = (int)(e - r * N);
    const ActD a = act_jet(Z[r * ldz + c], act, p);
    const float yb = Ybar[r * ldyb + c] * scale;
    float sb[3], acc = 0.f;
#pragma unroll
    for (int k = 1; k <= 3; ++k) {
      sb[k - 1] = Ybar[(k * P + r) * ldyb + c] * scale;
      acc += sb[k - 1] * Z[(k * P + r) * ldz + c];
    }
    Zbar[r * ldzb + c] = yb * a.d1 + a.d2 * acc;
#pragma unroll
    for (int k = 1; k <= 3; ++k) Zbar[(k * P + r) * ldzb + c] = sb[k - 1] * a.d1;
  }
}

inline int64_t pad4(int64_t v) { return (v + 3) / 4 * 4; }

struct JetLayout {
  int64_t R, ld_in, ld_act;
  int64_t off_in, off_z[RECMV_MLP_MAX_LAYERS], off_h[RECMV_MLP_MAX_LAYERS], off_ga, off_gb, off_park, off_tn;
  int64_t tn_bytes, bytes;
};

JetLayout jet_layout(const recmv_mlp* m, int64_t P) {
  JetLayout L;
  int64_t maxw = 4;
  for (int l = 0; l <= m->n_layers; ++l) maxw = m->dims[l] > maxw ? m->dims[l] : maxw;
  L.R = 4 * P;
  L.ld_in = pad4(m->dims[0]);
  L.ld_act = pad4(maxw);
  int64_t o = 0;
  auto take = [&](int64_t bytes) {
    int64_t r = o;
    o += (bytes + 255) / 256 * 256;
    return r;
  };
  L.off_in = take(L.R * L.ld_in * 4);
  for (int l = 0; l + 1 < m->n_layers; ++l) {
    L.off_z[l] = take(L.R * L.ld_act * 4);
    L.off_h[l] = take(L.R * L.ld_act * 4);
  }
  L.off_ga = take(L.R * L.ld_act * 4);
  L.off_gb = take(L.R * L.ld_act * 4);
  L.off_park = take(L.R * L.ld_in * 4);
  L.tn_bytes = (recmv_gemm_tn_workspace_bytes(L.ld_act, L.ld_act, L.R) + 255) / 256 * 256 +
               recmv_colsum_workspace_bytes(P, L.ld_act) + 512;
  L.off_tn = take(L.tn_bytes);
  L.bytes = o;
  return L;
}

int jet_check(const recmv_mlp* m) {
  RECMV_REQUIRE(m, "mlp_jet: NULL descriptor");
  RECMV_REQUIRE(m->n_layers >= 1 && m->n_layers <= RECMV_MLP_MAX_LAYERS, "mlp_jet: bad layer count");
  RECMV_REQUIRE(m->dims[0] == 3 + 6 * m->multires + m->cond_dim, "mlp_jet: dims[0] != 3+6L+cond_dim");
  for (int l = 0; l < m->n_layers; ++l) {
    RECMV_REQUIRE(m->W[l] && m->rows[l] > 0, "mlp_jet: layer %d incomplete", l);
    const int expect = (l + 1 == m->skip_layer) ? m->dims[l + 1] - (3 + 6 * m->multires) : m->dims[l + 1];
    RECMV_REQUIRE(m->rows[l] == expect, "mlp_jet: layer %d has %d rows, expected %d", l, m->rows[l], expect);
  }
  return RECMV_OK;
}

// The tangent rows' code / padding columns are zeroed by this kernel instead of a hipMemsetAsync over the whole tangent block (whose
// first d_pe columns posenc_jvp overwrites anyway): fewer bytes, and no runtime fill between this stream's kernels — in the bf16x6
// matrix mode, with the two garments' chains on two streams, about one run of the loop in four parted from the others in the rows
// this fill covers; with the kernel 8 runs of 8 were identical (profiles/r04_b3_presplit.txt).  RECMV_JET_FILL_KERNEL=0:
// the hipMemsetAsync (A/B; bit-identical results when both work).
__global__ __launch_bounds__(kBlk) void jet_zero_cols_kernel(float* __restrict__ p, int64_t ld, int64_t rows, int c0, int width) {
  const int64_t total = rows * width;
  for (int64_t e = (int64_t)blockIdx.x * kBlk + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlk) {
    const int64_t r = e / width;
    p[r * ld + c0 + (int)(e - r * width)] = 0.f;
  }
}

int g_jet_fill = -1;      // -1: not set yet (RECMV_JET_FILL_KERNEL decides at the first pass), 0: hipMemsetAsync, 1: kernel
inline bool jet_fill_kernel() {
  if (g_jet_fill < 0) {
    const char* e = getenv("RECMV_JET_FILL_KERNEL");
    g_jet_fill = !(e && e[0] == '0');
  }
  return g_jet_fill != 0;
}

}  // namespace
}  // namespace recmv

using namespace recmv;

#define RECMV_TRY(expr)                \
  do {                                 \
    int rc__ = (expr);                 \
    if (rc__ != RECMV_OK) return rc__; \
  } while (0)

extern "C" int recmv_set_jet_fill(int use_kernel) {
  const int prev = jet_fill_kernel() ? 1 : 0;
  g_jet_fill = use_kernel ? 1 : 0;
  return prev;
}

extern "C" int64_t recmv_mlp_jet_workspace_bytes(const recmv_mlp* m, int64_t P) {
  if (!m || P <= 0 || m->n_layers < 1 || m->n_layers > RECMV_MLP_MAX_LAYERS) return 0;
  return jet_layout(m, P).bytes;
}

extern "C" int recmv_mlp_jet_forward(const recmv_mlp* m, const float* x, const float* cond, int64_t ld_cond,
                                     const int64_t* cond_index, const float* eye3, int64_t P, int n_j, float* y,
                                     int64_t ldy, float* tang, void* workspace, int64_t workspace_bytes,
                                     void* stream) {
  RECMV_TRY(jet_check(m));
  RECMV_REQUIRE(P >= 0, "mlp_jet_forward: negative P");
  if (P == 0) return RECMV_OK;
  const int n = m->n_layers;
  const int n_out = m->rows[n - 1];
  RECMV_REQUIRE(x && y && tang && eye3 && workspace, "mlp_jet_forward: NULL pointer");
  RECMV_REQUIRE(n_j >= 1 && n_j <= n_out && ldy >= n_out, "mlp_jet_forward: bad n_j");
  RECMV_REQUIRE(m->cond_dim == 0 || cond, "mlp_jet_forward: the net takes a per-frame code but cond is NULL");
  RECMV_REQUIRE(!m->residual || n_out == 3, "mlp_jet_forward: residual nets are 3-d");
  const JetLayout L = jet_layout(m, P);
  if (workspace_bytes < L.bytes) {
    set_error("mlp_jet_forward: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.bytes);
    return RECMV_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)workspace;
  float* in = (float*)(base + L.off_in);
  const int d_pe = 3 + 6 * m->multires;
  // value rows: [gamma(x) | code[frame] | 0]; tangent rows k: [d gamma / d x_k | 0]
  RECMV_TRY(recmv_posenc_forward(x, 3, in, L.ld_in, m->cond_dim ? d_pe : (int)L.ld_in, P, m->multires, m->pe_weights,
                                 1.f, stream));
  if (m->cond_dim)
    RECMV_TRY(recmv_gather_rows(cond, ld_cond, cond_index, in + d_pe, L.ld_in, P, m->cond_dim, L.ld_in - d_pe, stream));
  if (jet_fill_kernel() && L.ld_in > d_pe) {
    hipLaunchKernelGGL(jet_zero_cols_kernel, dim3(stream_grid(3 * P * (L.ld_in - d_pe), kBlk)), dim3(kBlk), 0, s, in + P * L.ld_in,
                       L.ld_in, 3 * P, d_pe, (int)(L.ld_in - d_pe));
    RECMV_TRY(check_launch("mlp_jet_forward/zero"));
  } else {
    RECMV_HIP_TRY(hipMemsetAsync(in + P * L.ld_in, 0, (size_t)3 * P * L.ld_in * 4, s));
  }
  for (int k = 0; k < 3; ++k)
    RECMV_TRY(recmv_posenc_jvp(x, 3, eye3 + 3 * k, 0, in + (int64_t)(k + 1) * P * L.ld_in, L.ld_in, P, m->multires,
                               m->pe_weights, stream));
  const float* h = in;
  int64_t ldh = L.ld_in;
  for (int l = 0; l < n; ++l) {
    if (l == n - 1) {
      RECMV_TRY(recmv_gemm_nt(h, ldh, m->W[l], m->dims[l], m->bias[l], y, ldy, P, n_out, m->dims[l], RECMV_ACT_NONE,
                              0.f, 1.f, stream));
      RECMV_TRY(recmv_gemm_nt(h + P * ldh, ldh, m->W[l], m->dims[l], nullptr, tang, n_j, 3 * P, n_j, m->dims[l],
                              RECMV_ACT_NONE, 0.f, 1.f, stream));
      if (m->residual) RECMV_TRY(recmv_add_scaled_2d(y, ldy, x, 3, 1.f, y, ldy, P, 3, stream));
      break;
    }
    float* Z = (float*)(base + L.off_z[l]);
    float* Hn = (float*)(base + L.off_h[l]);
    const bool skip_next = l + 1 == m->skip_layer;
    RECMV_TRY(recmv_gemm_nt(h, ldh, m->W[l], m->dims[l], nullptr, Z, L.ld_act, L.R, m->rows[l], m->dims[l],
                            RECMV_ACT_NONE, 0.f, 1.f, stream));
    hipLaunchKernelGGL(jet_act_forward_kernel, dim3(stream_grid(P * m->rows[l], kBlk)), dim3(kBlk), 0, s, Z, L.ld_act,
                       m->bias[l], Hn, L.ld_act, P, m->rows[l], m->hidden_act, m->act_param,
                       skip_next ? kInvSqrt2 : 1.f);
    RECMV_TRY(check_launch("mlp_jet_forward/act"));
    if (skip_next)   // [.. | gamma / sqrt2] for the value rows and [.. | d gamma / sqrt2] for the tangent rows
      RECMV_TRY(recmv_add_scaled_2d(in, L.ld_in, in, L.ld_in, kInvSqrt2 - 1.f, Hn + m->rows[l], L.ld_act, L.R, d_pe,
                                    stream));
    h = Hn;
    ldh = L.ld_act;
  }
  return RECMV_OK;
}

// gy [P, n_out] (row stride ldgy) and gtang [3P, n_j] are the cotangents of the forward's outputs; either may be NULL
// (= zeros).  gW[l] ([rows[l], dims[l]]) / gb[l] ([rows[l]]) may be NULL individually.  g_in [4P, ld_in] receives the
// cotangent of the stacked layer-0 input (value rows: [gamma | code]); gx [P,3] (may be NULL) the cotangent of x.
extern "C" int recmv_mlp_jet_backward(const recmv_mlp* m, const float* x, const float* eye3, int64_t P, int n_j,
                                      const float* gy, int64_t ldgy, const float* gtang, float* const* gW,
                                      float* const* gb, float* g_in, float* gx, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  RECMV_TRY(jet_check(m));
  RECMV_REQUIRE(P >= 0, "mlp_jet_backward: negative P");
  if (P == 0) return RECMV_OK;
  const int n = m->n_layers;
  const int n_out = m->rows[n - 1];
  RECMV_REQUIRE(x && eye3 && workspace && gW && gb, "mlp_jet_backward: NULL pointer");
  RECMV_REQUIRE(n_j >= 1 && n_j <= n_out, "mlp_jet_backward: bad n_j");
  for (int l = 0; l < n; ++l) RECMV_REQUIRE(m->Wt[l], "mlp_jet_backward: layer %d has no transposed weight", l);
  const JetLayout L = jet_layout(m, P);
  if (workspace_bytes < L.bytes) {
    set_error("mlp_jet_backward: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)L.bytes);
    return RECMV_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)workspace;
  float* in = (float*)(base + L.off_in);
  float* G[2] = {(float*)(base + L.off_ga), (float*)(base + L.off_gb)};
  float* park = (float*)(base + L.off_park);
  char* tnws = base + L.off_tn;
  const int64_t cs_bytes = recmv_colsum_workspace_bytes(P, L.ld_act) + 256;
  char* csws = tnws + (L.tn_bytes - cs_bytes);
  const int64_t tn_bytes = L.tn_bytes - cs_bytes;
  const int d_pe = 3 + 6 * m->multires;
  bool parked = false;
  int cur = 0;
  // ---- last layer (no activation)
  {
    const int l = n - 1;
    const float* H = l == 0 ? in : (float*)(base + L.off_h[l - 1]);
    const int64_t ldh = l == 0 ? L.ld_in : L.ld_act;
    const int K = m->dims[l];
    if (gW[l]) {
      if (gy)
        RECMV_TRY(recmv_gemm_tn(gy, ldgy, H, ldh, gW[l], K, n_out, K, P, tnws, tn_bytes, stream));
      else
        RECMV_HIP_TRY(hipMemsetAsync(gW[l], 0, (size_t)n_out * K * 4, s));
      if (gtang) {
        float* tmp = G[cur];     // [n_j, K]
        RECMV_TRY(recmv_gemm_tn(gtang, n_j, H + P * ldh, ldh, tmp, K, n_j, K, 3 * P, tnws, tn_bytes, stream));
        RECMV_TRY(recmv_add_scaled_2d(gW[l], K, tmp, K, 1.f, gW[l], K, n_j, K, stream));
      }
    }
    if (gb[l]) {
      if (gy)
        RECMV_TRY(recmv_colsum(gy, ldgy, P, n_out, gb[l], csws, cs_bytes, stream));
      else
        RECMV_HIP_TRY(hipMemsetAsync(gb[l], 0, (size_t)n_out * 4, s));
    }
    float* Hbar = G[cur ^ 1];
    if (gy)
      RECMV_TRY(recmv_gemm_nt(gy, ldgy, m->Wt[l], n_out, nullptr, Hbar, L.ld_act, P, K, n_out, RECMV_ACT_NONE, 0.f, 1.f,
                              stream));
    else
      RECMV_HIP_TRY(hipMemsetAsync(Hbar, 0, (size_t)P * L.ld_act * 4, s));
    if (gtang)
      RECMV_TRY(recmv_gemm_nt(gtang, n_j, m->Wt[l], n_out, nullptr, Hbar + P * L.ld_act, L.ld_act, 3 * P, K, n_j,
                              RECMV_ACT_NONE, 0.f, 1.f, stream));
    else
      RECMV_HIP_TRY(hipMemsetAsync(Hbar + P * L.ld_act, 0, (size_t)3 * P * L.ld_act * 4, s));
    cur ^= 1;     // G[cur] now holds the cotangent of the last layer's input
  }
  // ---- hidden layers
  for (int l = n - 2; l >= 0; --l) {
    float* Ybar = G[cur];
    const float* Z = (const float*)(base + L.off_z[l]);
    const bool skip_next = l + 1 == m->skip_layer;
    if (skip_next) {
      RECMV_TRY(recmv_add_scaled_2d(Ybar + m->rows[l], L.ld_act, Ybar + m->rows[l], L.ld_act, kInvSqrt2 - 1.f, park,
                                    L.ld_in, L.R, d_pe, stream));
      parked = true;
    }
    hipLaunchKernelGGL(jet_act_backward_kernel, dim3(stream_grid(P * m->rows[l], kBlk)), dim3(kBlk), 0, s, Ybar,
                       L.ld_act, Z, L.ld_act, Ybar, L.ld_act, P, m->rows[l], m->hidden_act, m->act_param,
                       skip_next ? kInvSqrt2 : 1.f);
    RECMV_TRY(check_launch("mlp_jet_backward/act"));
    const float* Zbar = Ybar;
    const float* H = l == 0 ? in : (const float*)(base + L.off_h[l - 1]);
    const int64_t ldh = l == 0 ? L.ld_in : L.ld_act;
    const int K = m->dims[l], N = m->rows[l];
    if (gb[l]) RECMV_TRY(recmv_colsum(Zbar, L.ld_act, P, N, gb[l], csws, cs_bytes, stream));
    if (gW[l]) RECMV_TRY(recmv_gemm_tn(Zbar, L.ld_act, H, ldh, gW[l], K, N, K, L.R, tnws, tn_bytes, stream));
    float* Hbar = G[cur ^ 1];
    if (l > 0 || g_in || gx) {
      RECMV_TRY(recmv_gemm_nt(Zbar, L.ld_act, m->Wt[l], N, nullptr, Hbar, L.ld_act, L.R, K, N, RECMV_ACT_NONE, 0.f, 1.f,
                              stream));
      cur ^= 1;
    }
  }
  // ---- layer-0 input cotangent: add what entered through the skip connection, hand it out, push it through gamma
  if (g_in || gx) {
    float* Hbar = G[cur];
    if (parked) RECMV_TRY(recmv_add_scaled_2d(Hbar, L.ld_act, park, L.ld_in, 1.f, Hbar, L.ld_act, L.R, d_pe, stream));
    if (g_in) RECMV_TRY(recmv_add_scaled_2d(Hbar, L.ld_act, Hbar, L.ld_act, 0.f, g_in, L.ld_in, L.R, m->dims[0], stream));
    if (gx) {
      float* tmp = G[cur ^ 1];
      RECMV_TRY(recmv_posenc_vjp(x, 3, Hbar, L.ld_act, nullptr, 0, gx, P, m->multires, m->pe_weights, stream));
      for (int k = 0; k < 3; ++k) {
        RECMV_TRY(recmv_posenc_vjp(x, 3, Hbar + (int64_t)(k + 1) * P * L.ld_act, L.ld_act, eye3 + 3 * k, 0, tmp, P,
                                   m->multires, m->pe_weights, stream));
        RECMV_TRY(recmv_add_scaled_2d(gx, 3, tmp, 3, 1.f, gx, 3, P, 3, stream));
      }
      if (m->residual && gy) RECMV_TRY(recmv_add_scaled_2d(gx, 3, gy, ldgy, 1.f, gx, 3, P, 3, stream));
    }
  }
  return RECMV_OK;
}
