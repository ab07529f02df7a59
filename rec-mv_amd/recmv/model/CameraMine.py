"""Camera of the hot path: rays, camera centre, projection, half-pixel angular threshold.

Stand-alone restatement of the methods of `RectifiedPerspectiveCameras` that the optimisation loop calls
(model/CameraMine.py:146-208 of the reference).  The reference class derives from pytorch3d's CamerasBase
(absent here and only needed by the rasterisers, which are out of this tier's scope — SURVEY.md §8f).
"""
import numpy as np
import torch


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

    def to(self, device):
        self.focal_length = self.focal_length.to(device)
        self.principal_point = self.principal_point.to(device)
        self.R = self.R.to(device)
        self.T = self.T.to(device)
        self.device = device
        return self

    def view_rays(self, ps, cam_id=0):
        """Pixel (x, y, 1) -> unit world-space ray (CameraMine.py:146-169)."""
        f, pp = self.focal_length, self.principal_point
        r0 = -ps[:, 0] / f[cam_id, 0] + ps[:, 2] * pp[cam_id, 0] / f[cam_id, 0]
        r1 = -ps[:, 1] / f[cam_id, 1] + ps[:, 2] * pp[cam_id, 1] / f[cam_id, 1]
        rays = torch.stack([r0, r1, ps[:, 2]], dim=1)
        rays = rays / torch.norm(rays, p=2, dim=1, keepdim=True)
        Rt = self.R[cam_id].transpose(0, 1)
        return (rays.unsqueeze(-1) * Rt.unsqueeze(0)).sum(-2)            # rays @ R^T without BLAS

    def project(self, ps, cam_id=0):
        ps = (ps.unsqueeze(-1) * self.R[cam_id].unsqueeze(0)).sum(-2) + self.T[cam_id].view(1, 3)
        x = self.principal_point[cam_id, 0] - ps[:, 0] * self.focal_length[cam_id, 0] / ps[:, 2]
        y = self.principal_point[cam_id, 1] - ps[:, 1] * self.focal_length[cam_id, 1] / ps[:, 2]
        return torch.cat([x.view(-1, 1), y.view(-1, 1)], dim=1)

    def transform_points_ndc(self, ps, cam_id=0):
        """World points [P,3] -> (x_ndc, y_ndc, z_view): the full projection of the reference camera with its own
        calibration matrix (model/CameraMine.py:62-88, 281-300: fx' = fx/(W/2), px' = 1 - 1/W - px/(W/2), rows
        [fx',0,px',0],[0,fy',py',0]), so that pixel column c is hit by the ray of `view_rays((c, r, 1))`."""
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
