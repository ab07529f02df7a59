"""Autograd wrappers of the 3-D sampler (MCAcc/grid_sampler_mine.py:8-65 of the reference).

Same two Functions: the forward Function whose backward is itself a Function, so the deformer Jacobian
(built with create_graph=True, utils/utils.py:146-150) can be differentiated again by the loss.  Two
differences, both invisible to callers: the scatter into grad_input is skipped when the volume does not
require a gradient (it is a frozen buffer in the hot path), and unused second-order inputs are not
materialised as 181 MB zero volumes.
"""
import torch
from torch.autograd import Function

from .. import GridSamplerMine


class GridSamplerMine3dFunction(Function):
    @staticmethod
    def forward(ctx, input, grid, mode='bilinear', padding_mode='border', align_corners=False):
        ctx.save_for_backward(input, grid)
        if align_corners == True:
            raise NotImplementedError
        # the lane order of backward / double backward is chosen where the FORWARD ran (`with GridSamplerMine.exact_order():`
        # around the forward call is enough: autograd launches the other two later, outside the block, on its own thread)
        ctx.sampler_mode = GridSamplerMine.current_mode()
        return GridSamplerMine.forward(input, grid, 0, 1)

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        o0, o1 = GridSamplerMine3dBackwardFunction.apply(input, grid, grad_output, ctx.sampler_mode)
        return o0, o1, None, None, None


class GridSamplerMine3dBackwardFunction(Function):
    @staticmethod
    def forward(ctx, input, grid, grad_output, sampler_mode=None):
        ctx.save_for_backward(input, grid, grad_output)
        ctx.set_materialize_grads(False)
        ctx.need_gi = input.requires_grad
        ctx.sampler_mode = GridSamplerMine.current_mode() if sampler_mode is None else sampler_mode
        with GridSamplerMine.exact_order(ctx.sampler_mode == 1):
            gi, gg = GridSamplerMine.backward(input, grid, grad_output, 0, 1, need_grad_input=ctx.need_gi)
        return gi, gg

    @staticmethod
    def backward(ctx, grad_output_input, grad_output_grid):
        input, grid, grad_output = ctx.saved_tensors
        if grad_output_grid is None:
            grad_output_grid = torch.zeros_like(grid)
        with GridSamplerMine.exact_order(ctx.sampler_mode == 1):
            o0, o1, o2 = GridSamplerMine.dbackward(grad_output_input, grad_output_grid.contiguous(), input, grid,
                                                   grad_output, 0, 1, need_grad_input=ctx.needs_input_grad[0])
        return o0, o1, o2, None
