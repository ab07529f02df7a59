"""Coarse-to-fine sparse SDF-grid evaluation — `Seg3dLossless` of MCAcc/seg3d_lossless.py:13-428 of the reference.

Same class name, constructor signature and mutable public attributes (`query_func`, `balance_value`, `b_min`, `b_max`,
`resolutions`, `spacing_*`, `b{x,y,z}` — mutated from outside by discretizeSDF, OptimGarmentNetwork.py:585-586, and
set_hierarchical_config, utils/utils.py:335-347) and the same result as `_forward` (:233-428): dense query at the
coarsest level; per finer level a 2x-1 trilinear upsample, the voxels within one step of a sign disagreement that have
not been evaluated yet are queried and written back, and wherever a queried value contradicts the sign of the
interpolated one the not-yet-evaluated voxels of its 3^3 neighbourhood are queried too, until no contradiction is left.

How it is organised here (not the reference's lists): the SET of evaluated voxels is a volume of the current level.
  * device route (`use_cuda_impl=True` and a CUDA volume — what the loop uses): one bit per voxel and four kernels per
    level (csrc/seg3d.hip: select / points / apply / expand) around the HIP upsampler; the host reads one counter per
    query, nothing else crosses the bus;
  * volume route (any device, `use_cuda_impl=False`; the CPU port and the tests' cross-check): the same steps as whole-
    volume torch operations on a boolean volume (dilation = 3^3 max-pool, neighbourhood growth = max-pool of the
    conflict volume).
Both query exactly the voxels the reference queries, so the grids are identical up to the network's own arithmetic.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from .utils import SmoothConv3D, create_grid3D


class Seg3dLossless(nn.Module):
    def __init__(self, query_func, b_min, b_max, resolutions, channels=1, balance_value=0.5, align_corners=False,
                 visualize=False, debug=False, use_cuda_impl=False, faster=False, use_shadow=False, **kwargs):
        super().__init__()
        self.query_func = query_func
        as_row = lambda v: (v if torch.is_tensor(v) else torch.tensor(v)).float().view(1, 1, 3)
        self.register_buffer('b_min', as_row(b_min))
        self.register_buffer('b_max', as_row(b_max))
        res = torch.tensor([(r, r, r) for r in resolutions] if type(resolutions[0]) is int else resolutions)
        self.register_buffer('resolutions', res)                                   # rows (W, H, D), coarse to fine
        spacing = (self.b_max.view(3) - self.b_min.view(3)) / self.resolutions[-1].view(3).float()
        self.spacing_x, self.spacing_y, self.spacing_z = (spacing[i].item() for i in range(3))
        # marching-cubes origin: centre of the first voxel (seg3d_lossless.py:38-44)
        self.bx = self.b_min.view(-1)[0].item() + self.spacing_x / 2.
        self.by = self.b_min.view(-1)[1].item() + self.spacing_y / 2.
        self.bz = self.b_min.view(-1)[2].item() + self.spacing_z / 2.
        self.batchsize, self.channels = 1, channels
        self.balance_value = balance_value
        self.align_corners, self.visualize, self.debug = align_corners, visualize, debug
        self.use_cuda_impl, self.faster, self.use_shadow = use_cuda_impl, faster, use_shadow
        # the configurations the reference itself never leaves (model/network.py:293-305)
        assert channels == 1 and not align_corners and not visualize and not faster and not use_shadow, \
            "Seg3dLossless: only channels=1, align_corners=False, visualize=faster=use_shadow=False are supported"
        for r in self.resolutions:
            assert r[0] % 2 == 1 and r[1] % 2 == 1 and r[2] % 2 == 1, f"resolution {r} must be odd (2n-1 nesting)"
        for a, b in zip(self.resolutions[:-1], self.resolutions[1:]):
            assert torch.equal(2 * a - 1, b), "every level must be 2n-1 of the previous one"
        self.register_buffer('init_coords', create_grid3D(0, self.resolutions[-1] - 1, steps=self.resolutions[0],
                                                          device="cpu").unsqueeze(0))
        # attributes of the reference's module that outside code may look at (never read on this path)
        self.register_buffer('calculated', torch.zeros(tuple(int(v) for v in self.resolutions[-1].flip(0)),
                                                       dtype=torch.bool))
        self.smooth_conv3x3 = SmoothConv3D(in_channels=1, out_channels=1, kernel_size=3)
        # checkpoint key of the reference's module (`engine.gird8_offsets`, sic): the 27 offsets of a 3^3 block
        self.register_buffer('gird8_offsets', torch.stack(torch.meshgrid([torch.arange(-1, 2)] * 3, indexing="ij"))
                             .int().view(3, -1).t())

    # ------------------------------------------------------------------------------------------ queries
    def batch_eval(self, coords, **kwargs):
        """Integer voxel coordinates of the FINAL resolution [1,N,3] -> world points -> query_func -> [1,C,N]
        (seg3d_lossless.py:89-108)."""
        step = 1.0 / self.resolutions[-1].float()
        pts = coords.detach().float() / self.resolutions[-1] + step / 2
        pts = pts * (self.b_max - self.b_min) + self.b_min
        return self._query(pts, **kwargs)

    def _query(self, pts, **kwargs):
        out = self.query_func(**kwargs, points=pts)
        if type(out) is list:
            out = torch.stack(out)
        assert out.dim() == 3, "query_func should return a occupancy with shape of [bz, C, N]"
        return out

    def forward(self, **kwargs):
        return self._forward(**kwargs)

    def _forward(self, **kwargs):
        W0, H0, D0 = (int(v) for v in self.resolutions[0])
        occ = self.batch_eval(self.init_coords.clone(), **kwargs).view(1, 1, D0, H0, W0)
        if self.use_cuda_impl and occ.is_cuda:
            return self._forward_device(occ, **kwargs)
        return self._forward_volume(occ, **kwargs)

    # ------------------------------------------------------------------------------------------ volume route
    def _forward_volume(self, occ, **kwargs):
        bv = self.balance_value
        final = self.resolutions[-1]
        done = torch.ones(occ.shape[2:], dtype=torch.bool, device=occ.device)          # level 0: every voxel evaluated
        for res in self.resolutions[1:]:
            W, H, D = (int(v) for v in res)
            stride = ((final - 1) // (res - 1)).to(occ.device)
            with torch.no_grad():
                sign = F.interpolate((occ > bv).float(), size=(D, H, W), mode="trilinear", align_corners=True)
                parent, done = done, torch.zeros((D, H, W), dtype=torch.bool, device=occ.device)
                done[::2, ::2, ::2] = parent
            occ = F.interpolate(occ.float(), size=(D, H, W), mode="trilinear", align_corners=True)
            with torch.no_grad():
                disagree = ((sign > 0.0) & (sign < 1.0)).float()
                todo = (F.max_pool3d(disagree, 3, 1, 1)[0, 0] > 0) & ~done               # box filter > 0 == dilation
            flat = occ.view(-1)
            while True:
                with torch.no_grad():
                    idx = todo.view(-1).nonzero(as_tuple=True)[0]
                if idx.numel() == 0:
                    break
                with torch.no_grad():
                    xyz = torch.stack([idx % W, (idx // W) % H, idx // (W * H)], dim=-1) * stride
                    guess = flat[idx]
                values = self.batch_eval(xyz.unsqueeze(0), **kwargs).view(-1)
                flat[idx] = values
                with torch.no_grad():
                    done.view(-1)[idx] = True
                    wrong = torch.zeros_like(done)
                    wrong.view(-1)[idx] = (guess - bv) * (values - bv) < 0
                    # the 3^3 block around every contradiction (clamping at the faces == zero-padded max-pool)
                    todo = (F.max_pool3d(wrong[None, None].float(), 3, 1, 1)[0, 0] > 0) & ~done
        return occ

    # ------------------------------------------------------------------------------------------ device route
    def forward_multi(self, query_funcs, **kwargs):
        """The pyramids of SEVERAL fields over the same box and resolutions, level by level in lockstep (the re-mesh
        extracts the body and every garment, OptimGarmentNetwork.py:593-617): one counter read-back per step serves all
        fields, and the small launches of the coarse levels of one field overlap the others'.  Returns one volume per
        query function; each equals what `forward()` gives with that function."""
        W0, H0, D0 = (int(v) for v in self.resolutions[0])
        keep = self.query_func
        try:
            occs = []
            for q in query_funcs:
                self.query_func = q
                occs.append(self.batch_eval(self.init_coords.clone(), **kwargs).view(1, 1, D0, H0, W0))
            if self.use_cuda_impl and occs[0].is_cuda:
                return self._forward_device(occs, query_funcs, **kwargs)
            out = []
            for q, occ in zip(query_funcs, occs):
                self.query_func = q
                out.append(self._forward_volume(occ, **kwargs))
            return out
        finally:
            self.query_func = keep

    def _forward_device(self, occs, query_funcs=None, **kwargs):
        from .. import _lib as L
        from .. import interp2x_boundary3d
        single = not isinstance(occs, (list, tuple))
        if single:
            occs, query_funcs = [occs], [self.query_func]
        K = len(occs)
        lib, dev = L.lib(), occs[0].device
        bv = float(self.balance_value)
        final = [int(v) for v in self.resolutions[-1]]
        res_f = (C.c_float * 3)(*[float(v) for v in final])
        extent = (C.c_float * 3)(*(self.b_max - self.b_min).view(-1).tolist())
        origin = (C.c_float * 3)(*self.b_min.view(-1).tolist())
        counters = torch.zeros((K, 2), dtype=torch.int32, device=dev)          # per field: [list length, conflicts]
        host = torch.zeros((K, 2), dtype=torch.int32).pin_memory()
        words = lambda n: (n + 31) // 32 + 1

        def list_lengths():
            host.copy_(counters, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            return [int(host[k, 0]) for k in range(K)]

        D, H, W = occs[0].shape[2:]
        done = [torch.full((words(D * H * W),), -1, dtype=torch.int32, device=dev) for _ in range(K)]   # level 0: all
        keep = self.query_func
        try:
            with L.device_guard(dev):
                st = lambda: L.stream_ptr(dev)
                for res in self.resolutions[1:]:
                    W, H, D = (int(v) for v in res)
                    n_vox = D * H * W
                    stride = (C.c_int32 * 3)(*[(f - 1) // (r - 1) for f, r in zip(final, (W, H, D))])
                    todo = [None] * K
                    for k in range(K):
                        occs[k], boundary = interp2x_boundary3d.forward(occs[k].contiguous(), bv)
                        parent, done[k] = done[k], torch.empty(words(n_vox), dtype=torch.int32, device=dev)
                        todo[k] = torch.empty(n_vox, dtype=torch.int32, device=dev)
                        L.check(lib.recmv_seg3d_select(L.ptr(boundary), L.ptr(parent), D, H, W, L.ptr(done[k]),
                                                       L.ptr(todo[k]), n_vox, L.ptr(counters[k]), st()), "seg3d_select")
                    n = list_lengths()
                    while any(n):
                        for k in range(K):
                            if n[k] == 0:
                                continue
                            pts = torch.empty((1, n[k], 3), dtype=torch.float32, device=dev)
                            L.check(lib.recmv_seg3d_points(L.ptr(todo[k]), n[k], H, W, stride, res_f, extent, origin,
                                                           L.ptr(pts), st()), "seg3d_points")
                            self.query_func = query_funcs[k]
                            values = self._query(pts, **kwargs).reshape(-1).contiguous().float()
                            flags = torch.empty(n[k], dtype=torch.uint8, device=dev)
                            grown = torch.empty(min(27 * n[k], n_vox), dtype=torch.int32, device=dev)
                            L.check(lib.recmv_seg3d_apply(L.ptr(todo[k]), L.ptr(values), n[k], bv, L.ptr(occs[k]),
                                                          L.ptr(flags), L.ptr(counters[k, 1:]), st()), "seg3d_apply")
                            # grown unconditionally (no contradiction -> empty list): one read-back answers both questions
                            L.check(lib.recmv_seg3d_expand(L.ptr(todo[k]), L.ptr(flags), n[k], D, H, W, L.ptr(done[k]),
                                                           L.ptr(grown), grown.numel(), L.ptr(counters[k]), st()),
                                    "seg3d_expand")
                            todo[k] = grown
                        n = list_lengths()              # a field that had nothing to query left a 0 in its counter
        finally:
            self.query_func = keep
        return occs[0] if single else occs
