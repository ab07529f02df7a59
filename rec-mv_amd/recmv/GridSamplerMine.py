"""Drop-in for the reference's `GridSamplerMine` extension (MCAcc/cuda/GridSamplerMine.cpp:73-103).

Extensions over the reference, all optional: `need_grad_input=False` skips the scatter into the volume
(the hot path's volume is a frozen buffer), and `grad_output_input=None` in dbackward means zeros.
"""
import torch

from . import _lib as L


def _check(input, grid, interp, pad):
    # GridSamplerMine.cpp:24-71
    if not (isinstance(input, torch.Tensor) and isinstance(grid, torch.Tensor)):
        raise RuntimeError("grid_sampler(): expected input and grid to not be undefined")
    if input.device != grid.device:
        raise RuntimeError("grid_sampler(): expected input and grid to be on same device, but input is on "
                           f"{input.device} and grid is on {grid.device}")
    if input.dtype != grid.dtype:
        raise RuntimeError("grid_sampler(): expected input and grid to have same dtype, but input has "
                           f"{input.dtype} and grid has {grid.dtype}")
    if input.dim() != 5 or grid.dim() != 5:
        raise RuntimeError("grid_sampler(): expected 5D input and grid with same number of dimensions, but "
                           f"got input with sizes {tuple(input.shape)} and grid with sizes {tuple(grid.shape)}")
    if input.size(0) != grid.size(0):
        raise RuntimeError("grid_sampler(): expected grid and input to have same batch size")
    if grid.size(-1) != 3:
        raise RuntimeError("grid_sampler(): expected grid to have size 3 in last dimension")
    if interp != 0:
        raise RuntimeError("grid_sampler(): only support Bilinear now")
    if pad != 1:
        raise RuntimeError("grid_sampler(): only support Border Padding now")
    for i in range(2, 5):
        if input.size(i) <= 0:
            raise RuntimeError("grid_sampler(): expected input to have non-empty spatial dimensions")
    L.require_cuda(input, "input")


# The reference also dispatches its kernels for half (AT_DISPATCH_FLOATING_TYPES_AND_HALF, GridSamplerMineKernel.cu:931,963,1001); nothing
# in the loop samples in f16.  Here an f16 call runs the f32 kernels on up-cast operands and rounds the results to f16 ONCE: not the
# reference's per-operation half arithmetic (its intermediate roundings are not reproduced — a result can differ from it by a few f16
# ulp, on the accurate side), but the same API surface instead of a rejected dtype.
def _half(*ts):
    return any(t is not None and t.dtype == torch.float16 for t in ts)


def _up(t):
    return t.float() if t is not None and t.dtype == torch.float16 else t


def _down(t):
    return t.half() if t is not None else None


def forward(input, grid, interpolation_mode, padding_mode):
    _check(input, grid, interpolation_mode, padding_mode)
    if _half(input):
        return _down(forward(_up(input), _up(grid), interpolation_mode, padding_mode))
    N, C = input.size(0), input.size(1)
    out = torch.empty((N, C, grid.size(1), grid.size(2), grid.size(3)), dtype=input.dtype, device=input.device)
    di, dg, do = L.desc5(input), L.desc5(grid), L.desc5(out)
    with L.device_guard(input.device):
        L.check(L.lib().recmv_grid_sample3d_forward(L.ptr(input), di, L.ptr(grid), dg, L.ptr(out), do,
                                                    interpolation_mode, padding_mode, L.dtype_code(input),
                                                    L.stream_ptr(input.device)), "GridSamplerMine.forward")
    return out


def backward(input, grid, grad_output, interpolation_mode, padding_mode, need_grad_input=True):
    _check(input, grid, interpolation_mode, padding_mode)
    if _half(input):
        gi, gg = backward(_up(input), _up(grid), _up(grad_output), interpolation_mode, padding_mode, need_grad_input)
        return _down(gi), _down(gg)
    grad_input = torch.zeros_like(input) if need_grad_input else None
    grad_grid = torch.empty(grid.shape, dtype=grid.dtype, device=grid.device)  # contiguous
    di, dg, dgo = L.desc5(input), L.desc5(grid), L.desc5(grad_output)
    dgi = L.desc5(grad_input) if grad_input is not None else di
    with L.device_guard(input.device):
        L.check(L.lib().recmv_grid_sample3d_backward(L.ptr(input), di, L.ptr(grid), dg, L.ptr(grad_output), dgo,
                                                     L.ptr(grad_input), dgi, L.ptr(grad_grid),
                                                     interpolation_mode, padding_mode, L.dtype_code(input),
                                                     L.stream_ptr(input.device)), "GridSamplerMine.backward")
    return grad_input, grad_grid


def dbackward(grad_output_input, grad_output_grid, input, grid, grad_output, interpolation_mode, padding_mode,
              need_grad_input=True):
    _check(input, grid, interpolation_mode, padding_mode)
    if _half(input):
        gi, gg, ggo = dbackward(_up(grad_output_input), _up(grad_output_grid), _up(input), _up(grid), _up(grad_output),
                                interpolation_mode, padding_mode, need_grad_input)
        return _down(gi), _down(gg), _down(ggo)
    grad_input = torch.zeros_like(input) if need_grad_input else None
    grad_grid = torch.empty(grid.shape, dtype=grid.dtype, device=grid.device)
    ggo = torch.empty(grad_output.shape, dtype=grad_output.dtype, device=grad_output.device)
    di, dg, dgo = L.desc5(input), L.desc5(grid), L.desc5(grad_output)
    dgI = L.desc5(grad_output_input) if grad_output_input is not None else di
    dgG = L.desc5(grad_output_grid)
    dgi = L.desc5(grad_input) if grad_input is not None else di
    with L.device_guard(input.device):
        L.check(L.lib().recmv_grid_sample3d_dbackward(
            L.ptr(grad_output_input), dgI, L.ptr(grad_output_grid), dgG, L.ptr(input), di, L.ptr(grid), dg,
            L.ptr(grad_output), dgo, L.ptr(grad_input), dgi, L.ptr(grad_grid), L.ptr(ggo), L.desc5(ggo),
            interpolation_mode, padding_mode, L.dtype_code(input), L.stream_ptr(input.device)),
            "GridSamplerMine.dbackward")
    return grad_input, grad_grid, ggo


def current_mode():
    """The sampler mode in force (recmv_set_sampler_mode): 0 record-coalesced lanes, 1 the reference's summation order."""
    return int(L.lib().recmv_get_sampler_mode())


class exact_order:
    """`with GridSamplerMine.exact_order():` — backward / double backward sum their channels in the reference's order (one lane per
    point, GridSamplerMineKernel.cu:333-914; bit-equal to the oracle) instead of the default record-coalesced lanes (same terms,
    lane butterfly; within a few ulp of sum |terms|).  `recmv_set_sampler_mode` of the C ABI; RECMV_SAMPLER_EXACT=1 sets it for a
    whole process.  The autograd Functions of MCAcc.grid_sampler_mine record the mode at FORWARD time and apply it to their backward /
    double backward, so wrapping the forward call is enough.  (The switch itself is process-global: not for concurrent use from several
    threads with different modes.)"""

    def __init__(self, exact=True):
        self.mode = 1 if exact else 0

    def __enter__(self):
        self.prev = L.lib().recmv_set_sampler_mode(self.mode)
        return self

    def __exit__(self, *exc):
        L.lib().recmv_set_sampler_mode(self.prev)
        return False
