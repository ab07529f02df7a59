"""Drop-in for the reference's `FastMinv` extension (FastMinv/M3x3Inv.cpp:12-63)."""
import torch

from . import _lib as L


def Fast3x3Minv(ms):
    """ms [N,3,3] CUDA contiguous f32|f64 -> [invs [N,3,3], checks [N] bool]   (M3x3Inv.cpp:12-36)."""
    L.require_cuda(ms, "ms")
    L.require_contiguous(ms, "ms")
    if ms.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("rs must be a float/double tensor")
    N = ms.size(0)
    invs = torch.empty((N, 3, 3), dtype=ms.dtype, device=ms.device)
    checks = torch.empty((N,), dtype=torch.bool, device=ms.device)
    with L.device_guard(ms.device):
        L.check(L.lib().recmv_inv3x3_forward(L.ptr(ms), L.ptr(invs), L.ptr(checks), N, L.dtype_code(ms),
                                             L.stream_ptr(ms.device)), "Fast3x3Minv")
    return [invs, checks]


def Fast3x3Minv_backward(grads, invs):
    """grads, invs [N,3,3] contiguous same dtype -> outs = -(inv^T grads inv^T)   (M3x3Inv.cpp:38-59)."""
    L.require_cuda(grads, "grads")
    L.require_contiguous(grads, "grads")
    L.require_cuda(invs, "invs")
    L.require_contiguous(invs, "invs")
    if grads.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("grads must be a float/double tensor")
    if invs.dtype != grads.dtype:
        raise RuntimeError("invs must have same type with grads")
    N = invs.size(0)
    outs = torch.empty((N, 3, 3), dtype=invs.dtype, device=invs.device)
    with L.device_guard(invs.device):
        L.check(L.lib().recmv_inv3x3_backward(L.ptr(grads), L.ptr(invs), L.ptr(outs), N, L.dtype_code(invs),
                                              L.stream_ptr(invs.device)), "Fast3x3Minv_backward")
    return outs
