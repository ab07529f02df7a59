"""Coarse-to-fine sparse SDF-grid evaluation (MCAcc/seg3d_lossless.py:13-428 of the reference).

Same class name, constructor signature and mutable public attributes (`query_func`, `balance_value`,
`b_min`, `b_max`, `resolutions`, `spacing_*`, `b{x,y,z}` — mutated from outside by discretizeSDF,
OptimGarmentNetwork.py:585-586, and set_hierarchical_config, utils/utils.py:335-347) and the same
algorithm as `_forward` (:233-428): dense query at the coarsest level; per finer level 2x-1 trilinear
upsample, boundary = sign disagreement dilated by a 3^3 box, query only unevaluated boundary voxels,
scatter, then re-query 27-neighbourhoods of sign conflicts until none remain.

Differences in HOW: the upsample + boundary mask is the HIP kernel (interp2x_boundary3d) whenever the
volume lives on the GPU (`use_cuda_impl=True`, the default the recmv pipeline passes; the reference's
F.interpolate route is kept behind `use_cuda_impl=False`) and the box-filter dilation is a max-pool (exact
for 0/1 input).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .utils import SmoothConv3D, create_grid3D


class Seg3dLossless(nn.Module):
    def __init__(self, query_func, b_min, b_max, resolutions, channels=1, balance_value=0.5, align_corners=False,
                 visualize=False, debug=False, use_cuda_impl=False, faster=False, use_shadow=False, **kwargs):
        super().__init__()
        self.query_func = query_func
        b_min = b_min if torch.is_tensor(b_min) else torch.tensor(b_min)
        b_max = b_max if torch.is_tensor(b_max) else torch.tensor(b_max)
        self.register_buffer('b_min', b_min.float().view(1, 1, 3))
        self.register_buffer('b_max', b_max.float().view(1, 1, 3))
        if type(resolutions[0]) is int:
            resolutions = torch.tensor([(res, res, res) for res in resolutions])
        else:
            resolutions = torch.tensor(resolutions)
        self.register_buffer('resolutions', resolutions)
        tmp = ((self.b_max.view(3) - self.b_min.view(3)) / self.resolutions[-1].to(self.b_max.device).view(3).float())
        self.spacing_x = tmp[0].item()
        self.spacing_y = tmp[1].item()
        self.spacing_z = tmp[2].item()
        self.bx = self.b_min.view(-1)[0].item() + self.spacing_x / 2.
        self.by = self.b_min.view(-1)[1].item() + self.spacing_y / 2.
        self.bz = self.b_min.view(-1)[2].item() + self.spacing_z / 2.
        self.batchsize = self.b_min.size(0)
        assert self.batchsize == 1
        self.balance_value = balance_value
        self.channels = channels
        assert self.channels == 1
        self.align_corners = align_corners
        assert (align_corners == False)
        self.visualize = visualize
        assert visualize == False
        self.debug = debug
        self.use_cuda_impl = use_cuda_impl
        self.faster = faster
        assert faster == False, "the reference always runs faster=False (model/network.py:304)"
        self.use_shadow = use_shadow
        assert use_shadow == False
        for resolution in resolutions:
            assert resolution[0] % 2 == 1 and resolution[1] % 2 == 1, \
                f"resolution {resolution} need to be odd becuase of align_corner."
        init_coords = create_grid3D(0, resolutions[-1] - 1, steps=resolutions[0], device="cpu")
        self.register_buffer('init_coords', init_coords.unsqueeze(0).repeat(self.batchsize, 1, 1))
        calculated = torch.zeros((self.resolutions[-1][2], self.resolutions[-1][1], self.resolutions[-1][0]),
                                 dtype=torch.bool)
        self.register_buffer('calculated', calculated)
        gird8_offsets = torch.stack(torch.meshgrid([torch.tensor([-1, 0, 1]), torch.tensor([-1, 0, 1]),
                                                    torch.tensor([-1, 0, 1])], indexing="ij")).int().view(3, -1).t()
        self.register_buffer('gird8_offsets', gird8_offsets)
        self.smooth_conv3x3 = SmoothConv3D(in_channels=1, out_channels=1, kernel_size=3)
        if self.use_cuda_impl:
            from .interp2x_boundary3d import Interp2xBoundary3d
            self.upsampler = Interp2xBoundary3d(self.balance_value)

    # ------------------------------------------------------------------------------------------
    def batch_eval(self, coords, **kwargs):
        """coords: integer voxel coordinates of the FINAL resolution -> world points -> query_func
        (seg3d_lossless.py:89-108)."""
        coords = coords.detach()
        step = 1.0 / self.resolutions[-1].float()
        coords2D = coords.float() / self.resolutions[-1] + step / 2
        coords2D = coords2D * (self.b_max - self.b_min) + self.b_min
        occupancys = self.query_func(**kwargs, points=coords2D)
        if type(occupancys) is list:
            occupancys = torch.stack(occupancys)
        assert len(occupancys.size()) == 3, "query_func should return a occupancy with shape of [bz, C, N]"
        return occupancys

    def forward(self, **kwargs):
        return self._forward(**kwargs)

    def _upsample(self, occupancys, D, H, W):
        if self.use_cuda_impl and occupancys.is_cuda:
            self.upsampler.balance_value = self.balance_value
            return self.upsampler(occupancys.contiguous())
        with torch.no_grad():
            valid = F.interpolate((occupancys > self.balance_value).float(), size=(D, H, W), mode="trilinear",
                                  align_corners=True)
        occupancys = F.interpolate(occupancys.float(), size=(D, H, W), mode="trilinear", align_corners=True)
        return occupancys, (valid > 0.0) & (valid < 1.0)

    def _forward(self, **kwargs):
        calculated = self.calculated.clone()
        occupancys = None
        # The reference carries the already-evaluated voxels as a coordinate list (`coords_accum`) that it doubles per
        # level and de-duplicates with `unique(dim=1)` — a lexicographic sort of up to 10^5-10^6 rows, twice per level.
        # The same SET is kept here as a boolean volume of the current level (`done`): doubling = writing it to the
        # even lattice of the next level, union = a scatter of True.  Same voxels are queried, no sort.
        done = None
        for resolution in self.resolutions:
            W, H, D = [int(v) for v in resolution]
            stride = (self.resolutions[-1] - 1) // (resolution - 1)
            if torch.equal(resolution, self.resolutions[0]):
                coords = self.init_coords.clone()
                occupancys = self.batch_eval(coords, **kwargs).view(self.batchsize, self.channels, D, H, W)
                with torch.no_grad():
                    done = torch.ones((D, H, W), dtype=torch.bool, device=occupancys.device)   # every level-0 voxel
                    calculated[coords[0, :, 2], coords[0, :, 1], coords[0, :, 0]] = True
                continue

            with torch.no_grad():
                done_prev = done
                done = torch.zeros((D, H, W), dtype=torch.bool, device=occupancys.device)
                done[::2, ::2, ::2] = done_prev                                        # coords_accum * 2 (:271)
            occupancys, is_boundary = self._upsample(occupancys, D, H, W)
            with torch.no_grad():
                # 3^3 box filter > 0  ==  3^3 max-pool of the 0/1 mask (seg3d_lossless.py:296)
                is_boundary = (F.max_pool3d(is_boundary.float(), 3, 1, 1) > 0)[0, 0]
                is_boundary &= ~done                                                    # minus already computed (:299-301)
                point_coords = is_boundary.permute(2, 1, 0).nonzero(as_tuple=False).unsqueeze(0)
                point_indices = (point_coords[:, :, 2] * H * W + point_coords[:, :, 1] * W + point_coords[:, :, 0])
                R, C, D, H, W = occupancys.shape
                occupancys_interp = torch.gather(occupancys.reshape(R, C, D * H * W), 2, point_indices.unsqueeze(1))
                coords = point_coords * stride
            if coords.size(1) == 0:
                continue
            occupancys_topk = self.batch_eval(coords, **kwargs)
            R, C, D, H, W = occupancys.shape
            occupancys = (occupancys.reshape(R, C, D * H * W)
                          .scatter_(2, point_indices.unsqueeze(1).expand(-1, C, -1), occupancys_topk)
                          .view(R, C, D, H, W))
            with torch.no_grad():
                conflicts = ((occupancys_interp - self.balance_value) * (occupancys_topk - self.balance_value) < 0)[0, 0]
                done.view(-1)[point_indices[0]] = True                                  # union with the new voxels
                calculated[coords[0, :, 2], coords[0, :, 1], coords[0, :, 0]] = True

            while conflicts.sum() > 0:
                with torch.no_grad():
                    conflicts_coords = coords[0, conflicts, :]
                    conflicts_boundary = (conflicts_coords.int() + self.gird8_offsets.unsqueeze(1) * stride.int()
                                          ).reshape(-1, 3).long().unique(dim=0)
                    conflicts_boundary[:, 0] = conflicts_boundary[:, 0].clamp(0, calculated.size(2) - 1)
                    conflicts_boundary[:, 1] = conflicts_boundary[:, 1].clamp(0, calculated.size(1) - 1)
                    conflicts_boundary[:, 2] = conflicts_boundary[:, 2].clamp(0, calculated.size(0) - 1)
                    coords = conflicts_boundary[calculated[conflicts_boundary[:, 2], conflicts_boundary[:, 1],
                                                           conflicts_boundary[:, 0]] == False]
                    coords = coords.unsqueeze(0)
                    point_coords = coords // stride
                    point_indices = (point_coords[:, :, 2] * H * W + point_coords[:, :, 1] * W + point_coords[:, :, 0])
                    R, C, D, H, W = occupancys.shape
                    occupancys_interp = torch.gather(occupancys.reshape(R, C, D * H * W), 2,
                                                     point_indices.unsqueeze(1))
                    coords = point_coords * stride
                if coords.size(1) == 0:
                    break
                occupancys_topk = self.batch_eval(coords, **kwargs)
                with torch.no_grad():
                    conflicts = ((occupancys_interp - self.balance_value) *
                                 (occupancys_topk - self.balance_value) < 0)[0, 0]
                occupancys = (occupancys.reshape(R, C, D * H * W)
                              .scatter_(2, point_indices.unsqueeze(1).expand(-1, C, -1), occupancys_topk)
                              .view(R, C, D, H, W))
                with torch.no_grad():
                    done.view(-1)[point_indices[0]] = True
                    calculated[coords[0, :, 2], coords[0, :, 1], coords[0, :, 0]] = True
        return occupancys
