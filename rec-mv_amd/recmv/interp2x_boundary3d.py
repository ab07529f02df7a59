"""Drop-in for the reference's `interp2x_boundary3d` extension (MCAcc/cuda/interp2x_boundary3d.cpp:17-37)."""
import torch

from . import _lib as L


def forward(input, balance_value):
    L.require_cuda(input, "input")
    L.require_contiguous(input, "input")
    B, C, d, h, w = input.shape
    out = torch.empty((B, C, 2 * d - 1, 2 * h - 1, 2 * w - 1), dtype=input.dtype, device=input.device)
    bnd = torch.empty(out.shape, dtype=torch.bool, device=input.device)
    with L.device_guard(input.device):
        L.check(L.lib().recmv_interp2x_boundary3d_forward(L.ptr(input), L.ptr(out), L.ptr(bnd), B * C, d, h, w,
                                                          float(balance_value), L.dtype_code(input),
                                                          L.stream_ptr(input.device)), "interp2x_boundary3d.forward")
    return [out, bnd]


def backward(grad_output):
    L.require_cuda(grad_output, "grad_output")
    L.require_contiguous(grad_output, "grad_output")
    B, C, D, H, W = grad_output.shape
    gi = torch.empty((B, C, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2), dtype=grad_output.dtype,
                     device=grad_output.device)
    with L.device_guard(grad_output.device):
        L.check(L.lib().recmv_interp2x_boundary3d_backward(L.ptr(grad_output), L.ptr(gi), B * C, D, H, W,
                                                           L.dtype_code(grad_output),
                                                           L.stream_ptr(grad_output.device)),
                "interp2x_boundary3d.backward")
    return gi
