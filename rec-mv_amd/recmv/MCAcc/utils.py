"""Grid helpers of MCAcc/utils.py (reference lines 88-101, 133-146)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def create_grid3D(min, max, steps, device="cuda:0"):
    """Integer lattice coordinates [N,3] as (x,y,z), x fastest (MCAcc/utils.py:88-101)."""
    if type(min) is int:
        min = (min, min, min)
    if type(max) is int:
        max = (max, max, max)
    if type(steps) is int:
        steps = (steps, steps, steps)
    arrangeX = torch.linspace(min[0], max[0], steps[0]).long().to(device)
    arrangeY = torch.linspace(min[1], max[1], steps[1]).long().to(device)
    arrangeZ = torch.linspace(min[2], max[2], steps[2]).long().to(device)
    gridD, gridH, gridW = torch.meshgrid([arrangeZ, arrangeY, arrangeX], indexing="ij")
    coords = torch.stack([gridW, gridH, gridD])
    return coords.view(3, -1).t()


class SmoothConv3D(nn.Module):
    """k^3 box filter (MCAcc/utils.py:133-146); only `> 0` of its output is ever used (dilation)."""

    def __init__(self, in_channels, out_channels, kernel_size=3):
        super().__init__()
        assert kernel_size % 2 == 1, "kernel_size for smooth_conv must be odd: {3, 5, ...}"
        self.padding = (kernel_size - 1) // 2
        self.kernel_size = kernel_size
        weight = torch.ones((in_channels, out_channels, kernel_size, kernel_size, kernel_size),
                            dtype=torch.float32) / (kernel_size ** 3)
        self.register_buffer('weight', weight)

    def forward(self, input):
        return F.conv3d(input, self.weight, padding=self.padding)
