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

import torch

from . import _lib as L

ACT_NONE, ACT_RELU, ACT_SOFTPLUS, ACT_TANH = L.ACT_NONE, L.ACT_RELU, L.ACT_SOFTPLUS, L.ACT_TANH

_tn_ws = {}


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
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    else:
        assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    if bias is not None:
        bias = bias.contiguous()
        assert bias.numel() == N and bias.dtype == torch.float32
    lda = A.stride(0) if M > 1 else max(K, 1)
    ldb = B.stride(0) if N > 1 else max(K, 1)
    ldc = out.stride(0) if M > 1 else max(N, 1)
    with torch.cuda.device(A.device):
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
    out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    lib = L.lib()
    with torch.cuda.device(A.device):
        need = int(lib.recmv_gemm_tn_workspace_bytes(M, N, K))
        key = A.device.index
        ws = _tn_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 256), dtype=torch.uint8, device=A.device)
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
        out = torch.empty((P, width), dtype=torch.float32, device=x.device)
    assert out.stride(1) == 1 and out.shape[0] == P and out.shape[1] >= width
    ldo = out.stride(0) if P > 1 else out.shape[1]
    fill = width if ld_fill is None else ld_fill
    wbuf = None
    if weights is not None:
        assert len(weights) == 2 * multires
        wbuf = (C.c_float * (2 * multires))(*[float(w) for w in weights])
    with torch.cuda.device(x.device):
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
        if ctx.needs_input_grad[0]:
            gA = MatmulNT.apply(B, gC)                         # B [K,N] @ gC^T [N,M]
        if ctx.needs_input_grad[1]:
            gB = MatmulNT.apply(A, gC.t().contiguous())        # A [K,M] @ gC [M,N]
        return gA, gB


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
        d = _dact_from_output(y, ctx.act, ctx.act_param)
        gz = gy if d is None else gy * d
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
