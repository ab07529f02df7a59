"""Tensor-level wrappers and autograd Functions over the dense kernels of librecmv_hip.so.

The three MLPs of the hot path (SDF model/network.py:98-111, deformer model/Deformer.py:194-199, colour
model/RenderNet.py:83-94) are chains of y = act(x W^T + b).  In the reference every layer is an
nn.Linear (cuBLAS) + activation and the second-order terms of the loss (eikonal, normals, deformer
Jacobian — all built with create_graph=True) come from torch's autograd of those ops.  Here:

  * `linear_act`  — ONE fused MFMA kernel per layer (bias + activation + output scale in the epilogue);
  * its backward is written with the same kernels (`matmul_nt`, `matmul_tn`) wrapped as autograd
    Functions whose own backward is again made of them, so any order of differentiation stays on the
    hand-written kernels — no torch.matmul / rocBLAS anywhere on the path.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch

from . import _lib as L

ACT_NONE, ACT_RELU, ACT_SOFTPLUS, ACT_TANH = L.ACT_NONE, L.ACT_RELU, L.ACT_SOFTPLUS, L.ACT_TANH

_tn_ws = {}      # scratch per (device, stream): kernels of concurrent streams must not share a workspace
_lb_ws = {}


def _ws_key(dev):
    return (dev.index, L.raw_stream(dev))


def _forget_split(ptr):
    try:
        L.lib().recmv_b3_forget(ptr)
    except Exception:                      # interpreter shutdown
        pass


def presplit(W):
    """bf16x6 matrix mode only: split the weight matrix W [N,K] (contiguous, a fresh tensor per weight version — the output of the
    weight normalisation or its transpose) into its three bf16 planes ONCE (recmv_b3_split) so that the large products read the
    pieces instead of splitting W's tile again in every row tile of every launch.  The planes live on W (`_recmv_b3`) and the
    registration ends with W (weakref.finalize -> recmv_b3_forget).  RECMV_B3_PRESPLIT=0 keeps the in-loop split (A/B)."""
    lib = L.lib()
    if (not W.is_cuda or W.dtype != torch.float32 or W.dim() != 2 or lib.recmv_get_gemm_mode() != 1
            or os.environ.get('RECMV_B3_PRESPLIT', '1') == '0'):
        return W
    N, K = W.shape
    if K % 8 != 0 or N < 64 or not W.is_contiguous() or W.data_ptr() % 16 != 0 or getattr(W, '_recmv_b3', None) is not None:
        return W
    planes = L.scratch(int(lib.recmv_b3_planes_bytes(N, K)), torch.uint8, W.device)
    with L.device_guard(W.device):
        L.check(lib.recmv_b3_split(L.ptr(W), K, N, K, L.ptr(planes), planes.numel(), L.stream_ptr(W.device)), "b3_split")
    try:
        W._recmv_b3 = planes
        weakref.finalize(W, _forget_split, W.data_ptr())
    except Exception:
        lib.recmv_b3_forget(L.ptr(W))
    return W


def transposed(W):
    """Contiguous W^T, cached on the tensor object for as long as its data is unchanged.  The cache is shared by every
    stream that differentiates through W in an iteration (mask loss, curve branch, render loss): it keeps the event
    recorded behind the transpose, and a hit from another stream waits for it."""
    ent = getattr(W, "_recmv_entry", None)
    if ent is not None and ent[1] is not None and ent[1].data_ptr() == W.data_ptr() and ent[1]._version == W._version:
        # the alias of a shared weight-normed matrix (weight_norm_shared): its transpose is made once per optimiser step
        if ent[2] is None:
            L.acquire(ent[3])
            ent[2] = presplit(ent[1].t().contiguous())
            ent[4] = L.publish(W.device)
        else:
            L.acquire(ent[4])
        return ent[2]
    hit = getattr(W, "_recmv_t", None)
    if hit is not None and hit[0] == W._version:
        if hit[2] is not None and L.raw_stream(W.device) != hit[3]:
            torch.cuda.current_stream(W.device).wait_event(hit[2])
        return hit[1]
    Wt = presplit(W.detach().t().contiguous())
    ev = sid = None
    if W.is_cuda:
        ev = torch.cuda.Event()
        ev.record()
        sid = L.raw_stream(W.device)
    try:
        W._recmv_t = (W._version, Wt, ev, sid)
    except Exception:
        pass
    return Wt


def linear_backward(gy, y, x, W, act, act_param, need_gx=True, need_gW=True, need_gb=True):
    """(gx, gW, gb) of y = act(x W^T + b) from ONE C call (recmv_linear_backward); no autograd."""
    gy = _rowmajor(gy)
    x = _rowmajor(x.detach())
    M, N = gy.shape
    K = x.shape[1]
    dev = gy.device
    lib = L.lib()
    need = int(lib.recmv_linear_backward_workspace_bytes(M, N, K))
    ws = _lb_ws.get(_ws_key(dev))
    if ws is None or ws.numel() < need:
        ws = L.scratch(need, torch.uint8, dev)
        _lb_ws[_ws_key(dev)] = ws
    gx = L.scratch((M, K), torch.float32, dev) if need_gx else None
    gW = L.scratch((N, K), torch.float32, dev) if need_gW else None
    gb = L.scratch((N,), torch.float32, dev) if need_gb else None
    Wt = transposed(W) if need_gx else None
    yd = y.detach() if y is not None else None
    with L.device_guard(dev):
        L.check(lib.recmv_linear_backward(L.ptr(gy), gy.stride(0) if M > 1 else N, L.ptr(yd),
                                          (yd.stride(0) if M > 1 else N) if yd is not None else 0, L.ptr(x),
                                          x.stride(0) if M > 1 else K, L.ptr(Wt), N, M, N, K, act, float(act_param),
                                          L.ptr(gx), K, L.ptr(gW), L.ptr(gb), L.ptr(ws), ws.numel(),
                                          L.stream_ptr(dev)), "linear_backward")
    return gx, gW, gb


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    """2-D f32 CUDA tensor with unit inner stride (row stride may exceed the width)."""
    if t.dim() != 2:
        raise RuntimeError("recmv.ops: expected a 2-D tensor")
    if t.dtype != torch.float32:
        raise RuntimeError("recmv.ops: expected float32")
    L.require_cuda(t, "operand")
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def gemm_nt(A, B, bias=None, act=ACT_NONE, act_param=0.0, out_scale=1.0, out=None):
    """C[M,N] = act(A[M,K] @ B[N,K]^T + bias) * out_scale — recmv_gemm_nt."""
    A, B = _rowmajor(A), _rowmajor(B)
    M, K = A.shape
    N, K2 = B.shape
    if K != K2:
        raise RuntimeError(f"gemm_nt: inner dimensions differ ({K} vs {K2})")
    if out is None:
        out = L.scratch((M, N), torch.float32, A.device)
    else:
        assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    if bias is not None:
        bias = bias.contiguous()
        assert bias.numel() == N and bias.dtype == torch.float32
    lda = A.stride(0) if M > 1 else max(K, 1)
    ldb = B.stride(0) if N > 1 else max(K, 1)
    ldc = out.stride(0) if M > 1 else max(N, 1)
    with L.device_guard(A.device):
        L.check(L.lib().recmv_gemm_nt(L.ptr(A), lda, L.ptr(B), ldb, L.ptr(bias), L.ptr(out), ldc, M, N, K, act,
                                      float(act_param), float(out_scale), L.stream_ptr(A.device)), "gemm_nt")
    return out


def gemm_tn(A, B):
    """C[M,N] = A[K,M]^T @ B[K,N] — recmv_gemm_tn (deterministic split-K)."""
    A, B = _rowmajor(A), _rowmajor(B)
    K, M = A.shape
    K2, N = B.shape
    if K != K2:
        raise RuntimeError(f"gemm_tn: reduction dimensions differ ({K} vs {K2})")
    out = L.scratch((M, N), torch.float32, A.device)
    lib = L.lib()
    with L.device_guard(A.device):
        need = int(lib.recmv_gemm_tn_workspace_bytes(M, N, K))
        key = _ws_key(A.device)
        ws = _tn_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = L.scratch(max(need, 256), torch.uint8, A.device)
            _tn_ws[key] = ws
        lda = A.stride(0) if K > 1 else max(M, 1)
        ldb = B.stride(0) if K > 1 else max(N, 1)
        L.check(lib.recmv_gemm_tn(L.ptr(A), lda, L.ptr(B), ldb, L.ptr(out), max(N, 1), M, N, K, L.ptr(ws),
                                  ws.numel(), L.stream_ptr(A.device)), "gemm_tn")
    return out


def posenc(x, multires, weights=None, out_scale=1.0, out=None, ld_fill=None):
    """Positional encoding of x [P,3] -> [P, 3+6L] (model/Embedder.py:4-65) — recmv_posenc_forward.

    `out` may be a wider pre-allocated row-major buffer (e.g. the 512-wide skip-layer input); columns
    [3+6L, ld_fill) are zero-filled."""
    L.require_cuda(x, "x")
    assert x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 3
    if x.stride(1) != 1:
        x = x.contiguous()
    P = x.shape[0]
    width = 3 + 6 * multires
    if out is None:
        out = L.scratch((P, width), torch.float32, x.device)
    assert out.stride(1) == 1 and out.shape[0] == P and out.shape[1] >= width
    ldo = out.stride(0) if P > 1 else out.shape[1]
    fill = width if ld_fill is None else ld_fill
    wbuf = None
    if weights is not None:
        assert len(weights) == 2 * multires
        wbuf = (C.c_float * (2 * multires))(*[float(w) for w in weights])
    with L.device_guard(x.device):
        L.check(L.lib().recmv_posenc_forward(L.ptr(x), x.stride(0) if P > 1 else 3, L.ptr(out), ldo, fill, P,
                                             multires, C.cast(wbuf, C.c_void_p) if wbuf is not None else None,
                                             float(out_scale), L.stream_ptr(x.device)), "posenc")
    return out


# --------------------------------------------------------------------------------------------------
# autograd: products closed under differentiation
# --------------------------------------------------------------------------------------------------
class MatmulNT(torch.autograd.Function):
    """C = A @ B^T   (A [M,K], B [N,K])."""

    @staticmethod
    def forward(ctx, A, B):
        ctx.save_for_backward(A, B)
        return gemm_nt(A.detach(), B.detach())

    @staticmethod
    def backward(ctx, gC):
        A, B = ctx.saved_tensors
        gA = gB = None
        if not torch.is_grad_enabled():                        # first-order execution: raw kernels, no sub-graph
            if ctx.needs_input_grad[0]:
                gA = gemm_nt(gC, transposed(B))
            if ctx.needs_input_grad[1]:
                gB = gemm_tn(gC, A)
            return gA, gB
        if ctx.needs_input_grad[0]:
            gA = MatmulNT.apply(gC, B.t().contiguous())        # gC [M,N] @ B [N,K]
        if ctx.needs_input_grad[1]:
            gB = MatmulTN.apply(gC, A)                         # gC^T [N,M] @ A [M,K]
        return gA, gB


class MatmulTN(torch.autograd.Function):
    """C = A^T @ B   (A [K,M], B [K,N])."""

    @staticmethod
    def forward(ctx, A, B):
        ctx.save_for_backward(A, B)
        return gemm_tn(A.detach(), B.detach())

    @staticmethod
    def backward(ctx, gC):
        A, B = ctx.saved_tensors
        gA = gB = None
        if not torch.is_grad_enabled():
            if ctx.needs_input_grad[0]:
                gA = gemm_nt(B, gC)
            if ctx.needs_input_grad[1]:
                gB = gemm_nt(A, gC.t().contiguous())
            return gA, gB
        if ctx.needs_input_grad[0]:
            gA = MatmulNT.apply(B, gC)                         # B [K,N] @ gC^T [N,M]
        if ctx.needs_input_grad[1]:
            gB = MatmulNT.apply(A, gC.t().contiguous())        # A [K,M] @ gC [M,N]
        return gA, gB


def rows_sum_by_index(g, index, n):
    """out[k] = sum of the rows of `g` [P,C] whose `index` [P] is k, k < n — as a one-hot product on the split-K MFMA
    kernel: a FIXED summation order.  torch's `index_add_` / the backward of `index_select` add with float atomics in
    scheduling order; with a handful of frames and thousands of rays per frame that makes the per-frame gradients (and,
    through the optimiser, every later iteration) differ from run to run.  Differentiable (MatmulTN) when a graph is
    being built."""
    onehot = (index.view(-1, 1) == torch.arange(n, device=g.device).view(1, -1)).to(g.dtype)
    g2 = g.reshape(g.shape[0], -1)
    if g2.shape[0] == 0:
        return torch.zeros((n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
    if not g.is_cuda:
        out = torch.zeros(n, g2.shape[1], dtype=g.dtype, device=g.device).index_add(0, index, g2)   # sequential on the host
    elif torch.is_grad_enabled() and g.requires_grad:
        out = MatmulTN.apply(onehot, g2.contiguous())
    else:
        out = gemm_tn(onehot, g2.detach().contiguous())
    return out.view((n,) + tuple(g.shape[1:]))


class GatherRows(torch.autograd.Function):
    """rows[index] of a small per-frame table (per-frame codes, translations) with a deterministic backward
    (rows_sum_by_index) — the `cond[batch_inds]` of model/Deformer.py:190."""

    @staticmethod
    def forward(ctx, table, index):
        ctx.save_for_backward(index)
        ctx.n = table.shape[0]
        return table.index_select(0, index)

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        return rows_sum_by_index(g, index, ctx.n), None


def gather_rows(table, index):
    return GatherRows.apply(table, index)


def _dact_from_output(y, act, act_param):
    """act'(z) expressed through y = act(z) with differentiable torch ops."""
    if act == ACT_NONE:
        return None
    if act == ACT_RELU:
        return (y > 0).to(y.dtype)
    if act == ACT_SOFTPLUS:
        # y = log(1+e^{bz})/b  ->  sigmoid(bz) = 1 - e^{-by}
        return -torch.expm1(-act_param * y)
    if act == ACT_TANH:
        return 1.0 - y * y
    raise RuntimeError("unknown activation")


def _wbuf(weights, multires):
    if weights is None:
        return None
    assert len(weights) == 2 * multires
    return (C.c_float * (2 * multires))(*[float(w) for w in weights])


def _cptr(buf):
    return C.cast(buf, C.c_void_p) if buf is not None else None


# --------------------------------------------------------------------------------------------------
# autograd: positional encoding with first and second derivative, one launch each
# --------------------------------------------------------------------------------------------------
def _pe_vjp(x, g, t, multires, weights):
    x = x.contiguous()
    g = g if g.stride(1) == 1 else g.contiguous()
    out = L.scratch((x.shape[0], 3), torch.float32, x.device)
    if t is not None:
        t = t.contiguous()
    P = x.shape[0]
    with L.device_guard(x.device):
        L.check(L.lib().recmv_posenc_vjp(L.ptr(x), 3, L.ptr(g), g.stride(0) if P > 1 else g.shape[1], L.ptr(t), 3,
                                         L.ptr(out), P, multires, _cptr(_wbuf(weights, multires)),
                                         L.stream_ptr(x.device)), "posenc_vjp")
    return out


def _pe_jvp(x, t, multires, weights):
    x, t = x.contiguous(), t.contiguous()
    P = x.shape[0]
    out = L.scratch((P, 3 + 6 * multires), torch.float32, x.device)
    with L.device_guard(x.device):
        L.check(L.lib().recmv_posenc_jvp(L.ptr(x), 3, L.ptr(t), 3, L.ptr(out), out.shape[1], P, multires,
                                         _cptr(_wbuf(weights, multires)), L.stream_ptr(x.device)), "posenc_jvp")
    return out


class PosEnc(torch.autograd.Function):
    """gamma(x) (model/Embedder.py:4-65); differentiable to second order on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, multires, weights):
        ctx.save_for_backward(x)
        ctx.cfg = (multires, weights)
        return posenc(x.detach(), multires, weights)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if not torch.is_grad_enabled():
            return _pe_vjp(x.detach(), g, None, *ctx.cfg), None, None
        return PosEncVjp.apply(x, g, *ctx.cfg), None, None


class PosEncVjp(torch.autograd.Function):
    """gx = J(x)^T g."""

    @staticmethod
    def forward(ctx, x, g, multires, weights):
        ctx.save_for_backward(x, g)
        ctx.cfg = (multires, weights)
        return _pe_vjp(x.detach(), g.detach(), None, multires, weights)

    @staticmethod
    def backward(ctx, ggx):
        x, g = ctx.saved_tensors
        if not torch.is_grad_enabled():
            gx = _pe_vjp(x.detach(), g.detach(), ggx, *ctx.cfg) if ctx.needs_input_grad[0] else None
            gg = _pe_jvp(x.detach(), ggx, *ctx.cfg) if ctx.needs_input_grad[1] else None
            return gx, gg, None, None
        gx = PosEncVjp2.apply(x, g, ggx, *ctx.cfg) if ctx.needs_input_grad[0] else None
        gg = PosEncJvp.apply(x, ggx, *ctx.cfg) if ctx.needs_input_grad[1] else None
        return gx, gg, None, None


class PosEncJvp(torch.autograd.Function):
    """J(x) t."""

    @staticmethod
    def forward(ctx, x, t, multires, weights):
        ctx.save_for_backward(x, t)
        ctx.cfg = (multires, weights)
        return _pe_jvp(x.detach(), t.detach(), multires, weights)

    @staticmethod
    def backward(ctx, go):
        x, t = ctx.saved_tensors
        if not torch.is_grad_enabled():
            gx = _pe_vjp(x.detach(), go, t.detach(), *ctx.cfg) if ctx.needs_input_grad[0] else None
            gt = _pe_vjp(x.detach(), go, None, *ctx.cfg) if ctx.needs_input_grad[1] else None
            return gx, gt, None, None
        gx = PosEncVjp2.apply(x, go, t, *ctx.cfg) if ctx.needs_input_grad[0] else None
        gt = PosEncVjp.apply(x, go, *ctx.cfg) if ctx.needs_input_grad[1] else None
        return gx, gt, None, None


class PosEncVjp2(torch.autograd.Function):
    """t * d/dx <g, J(x) 1>  (second-derivative term; third order is never needed by the loss)."""

    @staticmethod
    def forward(ctx, x, g, t, multires, weights):
        return _pe_vjp(x.detach(), g.detach(), t.detach(), multires, weights)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        raise RuntimeError("recmv: third-order derivative of the positional encoding is not implemented")


# --------------------------------------------------------------------------------------------------
# autograd: activation-gradient step and weight norm
# --------------------------------------------------------------------------------------------------
def act_grad(gy, y, act, act_param):
    """gy * act'(z) through y = act(z), no autograd."""
    gy, y = gy.contiguous(), y.contiguous()
    out = L.scratch_like(gy)
    with L.device_guard(gy.device):
        L.check(L.lib().recmv_act_grad(L.ptr(gy), L.ptr(y), L.ptr(out), out.numel(), act, float(act_param),
                                       L.stream_ptr(gy.device)), "act_grad")
    return out


class ActGrad(torch.autograd.Function):
    """gz = gy * act'(z), written through y = act(z); one launch; differentiable once more."""

    @staticmethod
    def forward(ctx, gy, y, act, act_param):
        ctx.save_for_backward(gy, y)
        ctx.cfg = (act, act_param)
        gy_c, y_c = gy.detach().contiguous(), y.detach().contiguous()
        out = L.scratch_like(gy_c)
        with L.device_guard(gy.device):
            L.check(L.lib().recmv_act_grad(L.ptr(gy_c), L.ptr(y_c), L.ptr(out), out.numel(), act, float(act_param),
                                           L.stream_ptr(gy.device)), "act_grad")
        return out

    @staticmethod
    def backward(ctx, ggz):
        gy, y = ctx.saved_tensors
        act, p = ctx.cfg
        if not torch.is_grad_enabled():
            g_gy = act_grad(ggz, y.detach(), act, p) if ctx.needs_input_grad[0] else None
            g_y = None
            if ctx.needs_input_grad[1] and act in (ACT_SOFTPLUS, ACT_TANH):
                g_y = act_grad2(ggz, gy.detach(), y.detach(), act, p)
            return g_gy, g_y, None, None
        g_gy = ActGrad.apply(ggz, y, act, p) if ctx.needs_input_grad[0] else None
        g_y = None
        if ctx.needs_input_grad[1] and act in (ACT_SOFTPLUS, ACT_TANH):
            g_y = ActGrad2.apply(ggz, gy, y, act, p)
        return g_gy, g_y, None, None


def act_grad2(a, b, y, act, act_param):
    """a * b * d(act')/dy, no autograd."""
    a_c, b_c, y_c = a.detach().contiguous(), b.detach().contiguous(), y.detach().contiguous()
    out = L.scratch_like(a_c)
    with L.device_guard(a.device):
        L.check(L.lib().recmv_act_grad2(L.ptr(a_c), L.ptr(b_c), L.ptr(y_c), L.ptr(out), out.numel(), act,
                                        float(act_param), L.stream_ptr(a.device)), "act_grad2")
    return out


class ActGrad2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, y, act, act_param):
        return act_grad2(a, b, y, act, act_param)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        raise RuntimeError("recmv: third-order derivative through an activation is not implemented")


class WeightNorm(torch.autograd.Function):
    """W = g * v / ||v||_row (nn.utils.weight_norm, dim 0), one launch forward, one backward."""

    @staticmethod
    def forward(ctx, v, g):
        v_c, g_c = v.detach().contiguous(), g.detach().contiguous()
        rows, cols = v_c.shape
        W = L.scratch_like(v_c)
        norms = L.scratch(rows, torch.float32, v.device)
        with L.device_guard(v.device):
            L.check(L.lib().recmv_weight_norm_forward(L.ptr(v_c), L.ptr(g_c), L.ptr(W), L.ptr(norms), rows, cols,
                                                      L.stream_ptr(v.device)), "weight_norm")
        ctx.save_for_backward(v_c, g_c, norms)
        return presplit(W)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gW):
        v, g, norms = ctx.saved_tensors
        gW = gW.contiguous()
        gv = L.scratch_like(v)
        gg = L.scratch_like(g)
        with L.device_guard(v.device):
            L.check(L.lib().recmv_weight_norm_backward(L.ptr(v), L.ptr(g), L.ptr(norms), L.ptr(gW), L.ptr(gv),
                                                       L.ptr(gg), v.shape[0], v.shape[1], L.stream_ptr(v.device)),
                    "weight_norm_backward")
        return gv, gg


class WeightNormShared(torch.autograd.Function):
    """WeightNorm whose result lives in a per-layer cache entry `[key, W, W^T, token of W, token of W^T, norms]` (weight_norm_shared):
    the first pass of an optimiser step launches the kernel, every later pass — with or without a graph — takes a fresh alias of the
    same W (and, in its backward, the same W^T).  The SDF nets go through ~8 autograd passes per iteration: 9 normalisations and 9
    transposes each became 9 + 9 per STEP."""

    @staticmethod
    def forward(ctx, v, g, entry):
        if entry[1] is None:
            v_c, g_c = v.detach().contiguous(), g.detach().contiguous()
            rows, cols = v_c.shape
            W = L.scratch_like(v_c)
            norms = L.scratch(rows, torch.float32, v.device)
            with L.device_guard(v.device):
                L.check(L.lib().recmv_weight_norm_forward(L.ptr(v_c), L.ptr(g_c), L.ptr(W), L.ptr(norms), rows, cols,
                                                          L.stream_ptr(v.device)), "weight_norm")
            entry[1], entry[5], entry[3] = presplit(W), norms, L.publish(W.device)
        else:
            L.acquire(entry[3])
        ctx.save_for_backward(v.detach(), g.detach(), entry[5])
        return entry[1].detach()                       # a fresh alias: each graph gets its own node output, all share the data

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gW):
        v, g, norms = ctx.saved_tensors
        v, g = v.contiguous(), g.contiguous()
        gW = gW.contiguous()
        gv = L.scratch_like(v)
        gg = L.scratch_like(g)
        with L.device_guard(v.device):
            L.check(L.lib().recmv_weight_norm_backward(L.ptr(v), L.ptr(g), L.ptr(norms), L.ptr(gW), L.ptr(gv),
                                                       L.ptr(gg), v.shape[0], v.shape[1], L.stream_ptr(v.device)),
                    "weight_norm_backward")
        return gv, gg, None


def weight_norm_shared(module, l, v, g):
    """The weight-normed matrix of layer `l` of `module` for the CURRENT values of (v, g): computed once per parameter version (they
    only change at optimizer.step()) and shared by every pass of the step and every stream of the loop (the entry carries the events
    behind its producers; a hit on another stream waits).  No-grad callers get the cached tensor itself, autograd callers an alias
    behind a WeightNormShared node whose backward finds the cached W^T through `transposed()`.  RECMV_SHARED_WN=0: one kernel per
    autograd pass again (A/B)."""
    if not (v.is_cuda and v.dtype == torch.float32):
        return g * (v / v.norm(dim=1, keepdim=True))
    cache = module.__dict__.setdefault('_wn_cache', {})
    key = (v._version, g._version, v.data_ptr())
    hit = cache.get(l)
    if hit is None or hit[0] != key:
        hit = cache[l] = [key, None, None, None, None, None]        # key, W, W^T, token of W, token of W^T, row norms
    if not torch.is_grad_enabled() or not (v.requires_grad or g.requires_grad):
        if hit[1] is None:
            with torch.no_grad():
                WeightNormShared.apply(v, g, hit)
        else:
            L.acquire(hit[3])
        return hit[1]
    if os.environ.get('RECMV_SHARED_WN', '1') == '0':
        return WeightNorm.apply(v, g)
    W = WeightNormShared.apply(v, g, hit)
    W._recmv_entry = hit
    return W


def weight_norm(v, g):
    """g [rows,1], v [rows,cols] -> W; fused kernel on the GPU."""
    if v.is_cuda and v.dtype == torch.float32:
        return WeightNorm.apply(v, g)
    return g * (v / v.norm(dim=1, keepdim=True))


class DefRegu(torch.autograd.Function):
    """y [P] = GM(sum_i log^2 sigma_i(J)) per 3x3 Jacobian — the deformation regulariser of OptimGarmentNetwork.py:1143-1155
    (host torch.svd + log + utils.GMRobustError there) — with dy/dJ from the same launch (recmv_def_regu, csrc/def_regu.hip):
    the backward pass is one scaling."""

    @staticmethod
    def forward(ctx, J, c):
        Jc = J.detach().contiguous()
        P = Jc.shape[0]
        y = L.scratch(P, torch.float32, J.device)
        gJ = L.scratch_like(Jc)
        with L.device_guard(J.device):
            L.check(L.lib().recmv_def_regu(L.ptr(Jc), P, float(c), L.ptr(y), L.ptr(gJ), L.stream_ptr(J.device)), "def_regu")
        ctx.save_for_backward(gJ)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        gJ, = ctx.saved_tensors
        return gJ * gy.view(-1, 1, 1), None


def def_regu(J, c):
    """Per-matrix regulariser value (see DefRegu); f32 CUDA [P,3,3] only."""
    L.require_cuda(J, "J")
    if J.dtype != torch.float32 or J.dim() != 3 or J.shape[1:] != (3, 3):
        raise RuntimeError("recmv.ops.def_regu: expected a float32 [P,3,3] tensor")
    return DefRegu.apply(J, c)


class LinearAct(torch.autograd.Function):
    """y = act(x @ W^T + b), one fused kernel forward; backward of any order on the same kernels."""

    @staticmethod
    def forward(ctx, x, W, b, act, act_param):
        y = gemm_nt(x.detach(), W.detach(), None if b is None else b.detach(), act, act_param)
        ctx.act, ctx.act_param = act, act_param
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, W, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W, y = ctx.saved_tensors
        if not torch.is_grad_enabled():
            gx, gW, gb = linear_backward(gy, y, x, W, ctx.act, ctx.act_param, ctx.needs_input_grad[0],
                                         ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
            return gx, gW, gb, None, None
        gz = gy if ctx.act == ACT_NONE else ActGrad.apply(gy, y, ctx.act, ctx.act_param)
        gx = gW = gb = None
        if ctx.needs_input_grad[0]:
            gx = MatmulNT.apply(gz, W.t().contiguous())
        if ctx.needs_input_grad[1]:
            gW = MatmulTN.apply(gz, x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gz.sum(0)
        return gx, gW, gb, None, None


def linear_act(x, W, b=None, act=ACT_NONE, act_param=0.0):
    """Fused layer; uses the plain kernel when nothing needs a gradient."""
    if not torch.is_grad_enabled() or not (x.requires_grad or W.requires_grad or
                                           (b is not None and b.requires_grad)):
        return gemm_nt(x, W, b, act, act_param)
    return LinearAct.apply(x, W, b, act, act_param)
