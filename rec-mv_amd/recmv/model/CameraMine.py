"""Camera of the hot path: rays, camera centre, projection, half-pixel angular threshold.

Stand-alone restatement of the methods of `RectifiedPerspectiveCameras` that the optimisation loop calls
(model/CameraMine.py:146-208 of the reference).  The reference class derives from pytorch3d's CamerasBase
(absent here and only needed by the rasterisers, which are out of this tier's scope — SURVEY.md §8f).
"""
import os

import numpy as np
import torch

from .. import _lib as L


class _CamProject(torch.autograd.Function):
    """World points [P,3] -> NDC (mode 0) / screen (1) / pixel (2) coordinates: one kernel forward, one + a fixed-order reduction
    backward (csrc/camera.hip) instead of ~25 element-wise launches each way.  Gradients reach the points, the translation, the
    focal length and the principal point; first order only (nothing in the loop differentiates a projection twice)."""

    @staticmethod
    def forward(ctx, pts, T, f, pp, cam16, W, H, mode):
        p = pts.detach().contiguous()
        P = p.shape[0]
        out = L.scratch((P, 2 if mode == 2 else 3), torch.float32, p.device)
        with L.device_guard(p.device):
            L.check(L.lib().recmv_cam_project(L.ptr(p), P, L.ptr(cam16), float(W), float(H), int(mode), L.ptr(out),
                                              L.stream_ptr(p.device)), "cam_project")
        ctx.save_for_backward(p, cam16)
        ctx.args = (float(W), float(H), int(mode), T.shape, f.shape, pp.shape)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        p, cam16 = ctx.saved_tensors
        W, H, mode, sT, sf, spp = ctx.args
        P = p.shape[0]
        g = g.contiguous()
        g_pts = L.scratch((P, 3), torch.float32, p.device) if ctx.needs_input_grad[0] else None
        g7 = L.scratch(7, torch.float32, p.device)
        nf = int(L.lib().recmv_cam_partial_floats(P))
        part = L.scratch(max(nf, 8), torch.float32, p.device)
        with L.device_guard(p.device):
            L.check(L.lib().recmv_cam_project_backward(L.ptr(p), L.ptr(g), P, L.ptr(cam16), W, H, mode, L.ptr(g_pts), L.ptr(g7),
                                                       L.ptr(part), part.numel(), L.stream_ptr(p.device)), "cam_project_backward")
        return (g_pts, g7[0:3].view(sT) if ctx.needs_input_grad[1] else None, g7[3:5].view(sf) if ctx.needs_input_grad[2] else None,
                g7[5:7].view(spp) if ctx.needs_input_grad[3] else None, None, None, None, None)


class _CamRays(torch.autograd.Function):
    """Pixels -> unit world rays (view_rays): `pix` [P,3] float or (col, row) int64; gradients reach focal length / principal point."""

    @staticmethod
    def forward(ctx, f, pp, cam16, pix, col, row):
        if pix is not None:
            pix = pix.detach().contiguous()
            P, dev = pix.shape[0], pix.device
        else:
            col, row = col.contiguous(), row.contiguous()
            P, dev = col.shape[0], col.device
        out = L.scratch((P, 3), torch.float32, dev)
        with L.device_guard(dev):
            L.check(L.lib().recmv_cam_rays(L.ptr(pix), L.ptr(col), L.ptr(row), P, L.ptr(cam16), L.ptr(out), L.stream_ptr(dev)),
                    "cam_rays")
        ctx.save_for_backward(pix, col, row, cam16)
        ctx.shapes = (f.shape, pp.shape)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        pix, col, row, cam16 = ctx.saved_tensors
        dev = cam16.device
        P = g.shape[0]
        g = g.contiguous()
        g4 = L.scratch(4, torch.float32, dev)
        part = L.scratch(max(int(L.lib().recmv_cam_partial_floats(P)), 8), torch.float32, dev)
        with L.device_guard(dev):
            L.check(L.lib().recmv_cam_rays_backward(L.ptr(pix), L.ptr(col), L.ptr(row), L.ptr(g), P, L.ptr(cam16), L.ptr(g4),
                                                    L.ptr(part), part.numel(), L.stream_ptr(dev)), "cam_rays_backward")
        sf, spp = ctx.shapes
        return (g4[0:2].view(sf) if ctx.needs_input_grad[0] else None, g4[2:4].view(spp) if ctx.needs_input_grad[1] else None,
                None, None, None, None)


class RectifiedPerspectiveCameras:
    def __init__(self, focal_length, principal_point, R, T, image_size, device=None):
        self.focal_length = focal_length          # [N,2]
        self.principal_point = principal_point    # [N,2]
        self.R = R                                # [N,3,3]
        self.T = T                                # [N,3]
        if not torch.is_tensor(image_size):
            image_size = torch.tensor(image_size, dtype=torch.float32)
        self.image_size = image_size              # [N,2] (W,H)
        self.device = device if device is not None else R.device
        self._pack()

    def _pack(self):
        """The camera as 16 device floats for the fused kernels (R, T, focal length, principal point), packed WHEN THE CAMERA IS BUILT,
        on the stream that builds it: a stream that may read the camera's own tensors may read the packed copy (the loop builds its
        cameras on the main stream before the side streams fork).  (None, None) where the kernels do not apply: host tensors, other
        dtypes, several cameras, a rotation that wants a gradient, RECMV_FUSED_CAMERA=0."""
        self._cam16 = (None, None)
        R, T, f, pp = self.R, self.T, self.focal_length, self.principal_point
        if (all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 for t in (R, T, f, pp)) and R.shape[0] == 1
                and T.shape[0] == 1 and f.shape[0] == 1 and pp.shape[0] == 1 and not R.requires_grad
                and os.environ.get('RECMV_FUSED_CAMERA', '1') != '0'):
            with torch.no_grad():
                c = torch.cat([R.reshape(-1), T.reshape(-1), f.reshape(-1), pp.reshape(-1)]).contiguous()
            self._cam16 = (c, L.raw_stream(c.device))

    def _packed(self, x, cam_id=0):
        if not (x.is_cuda and x.dtype == torch.float32 and cam_id == 0):
            return None
        t, made_on = self._cam16
        if t is not None and L.raw_stream(t.device) != made_on:
            t.record_stream(torch.cuda.current_stream(t.device))      # read on a side stream: the allocator must know
        return t

    def to(self, device):
        self.focal_length = self.focal_length.to(device)
        self.principal_point = self.principal_point.to(device)
        self.R = self.R.to(device)
        self.T = self.T.to(device)
        self.device = device
        self._pack()
        return self

    def view_rays(self, ps, cam_id=0):
        """Pixel (x, y, 1) -> unit world-space ray (CameraMine.py:146-169)."""
        f, pp = self.focal_length, self.principal_point
        c16 = self._packed(ps, cam_id)
        if c16 is not None:
            return _CamRays.apply(f, pp, c16, ps, None, None)
        r0 = -ps[:, 0] / f[cam_id, 0] + ps[:, 2] * pp[cam_id, 0] / f[cam_id, 0]
        r1 = -ps[:, 1] / f[cam_id, 1] + ps[:, 2] * pp[cam_id, 1] / f[cam_id, 1]
        rays = torch.stack([r0, r1, ps[:, 2]], dim=1)
        rays = rays / torch.norm(rays, p=2, dim=1, keepdim=True)
        Rt = self.R[cam_id].transpose(0, 1)
        return (rays.unsqueeze(-1) * Rt.unsqueeze(0)).sum(-2)            # rays @ R^T without BLAS

    def view_rays_pix(self, col_inds, row_inds, cam_id=0):
        """view_rays of the pixel centres (col, row, 1) given as integer index tensors — what the loop asks for
        (OptimGarmentNetwork.py:1032-1033, :2168-2170 build the float [P,3] tensor by hand)."""
        if col_inds.is_cuda and col_inds.dtype == torch.int64 and row_inds.dtype == torch.int64 and cam_id == 0:
            c16 = self._packed(self.focal_length, cam_id)
            if c16 is not None:
                return _CamRays.apply(self.focal_length, self.principal_point, c16, None, col_inds.reshape(-1), row_inds.reshape(-1))
        return self.view_rays(torch.cat([col_inds.view(-1, 1), row_inds.view(-1, 1), torch.ones_like(col_inds.view(-1, 1))],
                                        dim=-1).float(), cam_id)

    def _fused_project(self, ps, mode, cam_id):
        c16 = self._packed(ps, cam_id) if ps.dim() == 2 and ps.shape[-1] == 3 else None
        if c16 is None:
            return None
        return _CamProject.apply(ps, self.T, self.focal_length, self.principal_point, c16, float(self.image_size[cam_id, 0]),
                                 float(self.image_size[cam_id, 1]), mode)

    def project(self, ps, cam_id=0):
        out = self._fused_project(ps, 2, cam_id)
        if out is not None:
            return out
        ps = (ps.unsqueeze(-1) * self.R[cam_id].unsqueeze(0)).sum(-2) + self.T[cam_id].view(1, 3)
        x = self.principal_point[cam_id, 0] - ps[:, 0] * self.focal_length[cam_id, 0] / ps[:, 2]
        y = self.principal_point[cam_id, 1] - ps[:, 1] * self.focal_length[cam_id, 1] / ps[:, 2]
        return torch.cat([x.view(-1, 1), y.view(-1, 1)], dim=1)

    def transform_points_ndc(self, ps, cam_id=0):
        """World points [P,3] -> (x_ndc, y_ndc, z_view): the full projection of the reference camera with its own
        calibration matrix (model/CameraMine.py:62-88, 281-300: fx' = fx/(W/2), px' = 1 - 1/W - px/(W/2), rows
        [fx',0,px',0],[0,fy',py',0]), so that pixel column c is hit by the ray of `view_rays((c, r, 1))`."""
        out = self._fused_project(ps, 0, cam_id)
        if out is not None:
            return out
        v = (ps.unsqueeze(-1) * self.R[cam_id].unsqueeze(0)).sum(-2) + self.T[cam_id].view(1, 3)
        W = float(self.image_size[cam_id, 0])          # host tensor: no device round trip
        H = float(self.image_size[cam_id, 1])
        fx = self.focal_length[cam_id, 0] / (W / 2.0)
        fy = self.focal_length[cam_id, 1] / (H / 2.0)
        px = 1. - 1. / W - self.principal_point[cam_id, 0] / (W / 2.0)
        py = 1. - 1. / H - self.principal_point[cam_id, 1] / (H / 2.0)
        z = v[:, 2]
        return torch.stack([(fx * v[:, 0] + px * z) / z, (fy * v[:, 1] + py * z) / z, z], dim=1)

    def transform_points_screen(self, points, image_size=None, cam_id=0):
        """World points [N,P,3] (or [P,3]) -> (screen_x, screen_y, ndc_z) with the reference's override of the
        pytorch3d method (model/CameraMine.py:104-142): screen = (S-1)/2 - S * ndc / 2, so that screen_x / screen_y are
        the pixel coordinates `view_rays` shoots through; ndc_z is the projective depth 1 / z_view."""
        W = float(self.image_size[cam_id, 0])
        H = float(self.image_size[cam_id, 1])
        flat = points.reshape(-1, 3)
        out = self._fused_project(flat, 1, cam_id)
        if out is not None:
            return out.view(points.shape)
        ndc = self.transform_points_ndc(flat, cam_id)
        sx = (W - 1.) / 2. - W * ndc[:, 0] / 2.
        sy = (H - 1.) / 2. - H * ndc[:, 1] / 2.
        return torch.stack((sx, sy, 1.0 / ndc[:, 2]), dim=1).view(points.shape)

    def angThreshold(self, pixoffset=0.4, cam_id=0):
        """Smallest angle (degrees) subtended by `pixoffset` pixels at the image border
        (CameraMine.py:176-205)."""
        H = float(self.image_size[cam_id, 1])
        W = float(self.image_size[cam_id, 0])
        cx = self.principal_point[cam_id, 0].item()
        cy = self.principal_point[cam_id, 1].item()
        fx = self.focal_length[cam_id, 0].item()
        fy = self.focal_length[cam_id, 1].item()

        def ang(r1, r2):
            r1, r2 = torch.tensor(r1), torch.tensor(r2)
            return torch.arcsin(torch.linalg.cross(r1, r2).norm() / (r1.norm() * r2.norm())) / np.pi * 180.

        thred = ang([(W - cx) / fx, 0., 1.], [(W + pixoffset - cx) / fx, 0., 1.])
        thred = torch.min(thred, ang([-cx / fx, 0., 1.], [(pixoffset - cx) / fx, 0., 1.]))
        thred = torch.min(thred, ang([0., (H - cy) / fy, 1.], [0., (H + pixoffset - cy) / fy, 1.]))
        thred = torch.min(thred, ang([0., -cy / fy, 1.], [0., (pixoffset - cy) / fy, 1.]))
        return thred.item()

    def cam_pos(self, cam_id=0):
        return -(self.R[cam_id] * self.T[cam_id].view(1, 3)).sum(-1)
