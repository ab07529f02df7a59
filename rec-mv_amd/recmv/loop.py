"""One optimiser iteration of the per-frame implicit-surface loop on synthetic frames.

This is the hot path of `train.py:317-351` -> `OptimGarmentNetwork.forward` (engineer/networks/
OptimGarmentNetwork.py:1885-1969) -> `loss.backward()` -> `propagateTmpPsGrad` (:2159-2313) ->
`optimizer.step()`, restated on the recmv kernels, with the SAME call structure per garment:

  marching_cube_update  :678-740   every `remesh_intersect` iterations: Seg3dLossless + MC for body + garments
  mask_loss             :841-981   deformer on all garment MC vertices of all N frames (fwd+bwd), LBS-only
                                   consistency term, explicit-vertex SGD step, |SDF(verts)| loss
  sample_train_ray      :983-1055  ~sample_pix/garments*N rays per garment
  opt_garment_surface_ps:1057-1081 root finder, <= 20 steps
  surface_render_loss   :1083-1219 eikonal, deformation regulariser, SDF normal, cardinal rays, colour MLP,
                                   colour L1, weighted normal loss
  dct_poses_loss        :1221-1250 (pose smoothness on 30-frame windows)
  propagateTmpPsGrad    :2159-2313 implicit differentiation of the surface point

Everything of `OptimGarmentNetwork.forward` is here.  The two pytorch3d renderers on the path are restated on HIP
kernels (recmv/raster.py):
  * `pcRender` (point splat, 50 points per pixel, alpha compositor, :937) -> csrc/rasterize_points.hip, forward and
    backward; the IoU mask loss of compute_garment_pc_loss (:621-667) runs on its silhouettes;
  * `maskRender` (first-hit mesh rasteriser, :767) -> csrc/rasterize_meshes.hip; the visible canonical surface points
    come from its fragments + `utils.FindSurfacePs`, and `sample_train_ray` draws its Bernoulli subset of them with the
    host RNG exactly like the reference (:1020).
The feature-curve branch (`project_2d_loss` :1772-1883, recmv/curves.py) is optional (`curves=True`): SURVEY.md §8f
"next" row 3.  Frames are synthetic (SyntheticFrames): images, normals, garment segmentations and 2-D feature lines are
generated, not loaded.
The CPU SVD of the deformer Jacobians (:1148, a host round trip per garment per iteration) is replaced by
closed-form singular values on the device (`singular_values_3x3`).
"""
from __future__ import annotations

import contextlib
import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import MCGpu
from . import curves as fl
from . import raster
from . import utils
from .FastMinv import Fast3x3Minv
from .MCAcc import Seg3dLossless
from .model import CompositeDeformer, LBSkinner, RectifiedPerspectiveCameras, getRenderNet, getTmpSdf, getTranslatorNet
from .utils.constant import CURVE_AWARE, FL_INFOS, MASK_KEYS, TEMPLATE_GARMENT

SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
                        dtype=np.int64)

# coarse-to-fine grid pyramids of train.py:42-79 (W, H, D)
RESOLUTIONS = {
    'coarse': [(15, 21, 9), (29, 41, 17), (57, 81, 33), (113, 161, 65), (225, 321, 129)],
    'medium': [(19, 25, 13), (37, 49, 25), (73, 97, 49), (145, 193, 97), (289, 385, 193)],
    'fine': [(21, 27, 15), (41, 53, 29), (81, 105, 57), (161, 209, 113), (321, 417, 225)],
    'higher256': [(33, 33, 33), (65, 65, 65), (129, 129, 129), (257, 257, 257)],
}


_SV_CONST = {}


def singular_values_3x3(J):
    """Singular values of [P,3,3] matrices, descending, closed form on the device: square roots of the
    eigenvalues of J^T J (trigonometric solution of the symmetric 3x3 characteristic polynomial).
    Replaces `torch.svd(Jacobs.cpu())` (OptimGarmentNetwork.py:1148).  Differentiable.  Written on whole [P,3,3] / [P,3]
    tensors (about twenty launches forward; the element-by-element form of round 1 took sixty)."""
    key = (J.device, J.dtype)
    const = _SV_CONST.get(key)
    if const is None:
        const = _SV_CONST[key] = (torch.eye(3, device=J.device, dtype=J.dtype),
                                  torch.arange(2, device=J.device, dtype=J.dtype) * (2.0 * math.pi / 3.0))
        if J.is_cuda:
            torch.cuda.current_stream(J.device).synchronize()     # built once, read from every stream afterwards
    eye, shifts = const
    A = (J.transpose(-1, -2).unsqueeze(-1) * J.unsqueeze(-3)).sum(-2)        # J^T J without BLAS
    q = torch.diagonal(A, dim1=-2, dim2=-1).sum(-1) / 3.0                    # trace / 3
    B0 = A - q.view(-1, 1, 1) * eye
    p = torch.sqrt(torch.clamp((B0 * B0).sum((-2, -1)) / 6.0, min=1e-30))    # sum (a_ii - q)^2 + 2 sum_{i<j} a_ij^2
    B = B0 / p.view(-1, 1, 1)
    detB = (B[:, 0] * torch.linalg.cross(B[:, 1], B[:, 2], dim=-1)).sum(-1)
    r = torch.clamp(detB / 2.0, -1.0 + 1e-7, 1.0 - 1e-7)
    phi = torch.acos(r) / 3.0
    e02 = q.unsqueeze(-1) + 2.0 * p.unsqueeze(-1) * torch.cos(phi.unsqueeze(-1) + shifts)   # largest, smallest
    e1 = 3.0 * q - e02.sum(-1)
    ev = torch.stack([e02[:, 0], e1, e02[:, 1]], dim=1)
    return torch.sqrt(torch.clamp(ev, min=1e-20))


def dct_nullspace(nlen=30, keep=10, device="cpu"):
    """Rows of the orthonormal DCT-II basis above the `keep` lowest frequencies: projecting a length-`nlen`
    joint trajectory on them measures its high-frequency content (dct_poses_loss, :1221-1250)."""
    n = torch.arange(nlen, dtype=torch.float64)
    k = torch.arange(nlen, dtype=torch.float64).view(-1, 1)
    basis = torch.cos(math.pi / nlen * (n + 0.5) * k) * math.sqrt(2.0 / nlen)
    basis[0] *= 1.0 / math.sqrt(2.0)
    return basis[keep:].float().to(device)


class SyntheticFrames:
    """Stands in for dataset/dataset.py: per-frame learnable tensors + camera + images
    (`get_grad_parameters`, `get_camera_parameters`, `learnable_weights`: dataset.py:253-258, 425-437)."""

    def __init__(self, n_frames, n_garments, H, W, device, seed=0, condlen=128, rendlen=256, image_frames=None):
        g = torch.Generator().manual_seed(seed)
        self.F, self.H, self.W = n_frames, H, W
        self.device = device
        self.poses = (0.15 * torch.randn(n_frames, 24, 3, generator=g)).to(device).requires_grad_(True)
        self.trans = (0.01 * torch.randn(n_frames, 3, generator=g)).to(device).requires_grad_(True)
        self.d_cond = (0.1 * torch.randn(n_frames, condlen * (1 + n_garments), generator=g)).to(device).requires_grad_(True)
        self.rendcond = (0.1 * torch.randn(n_frames, rendlen, generator=g)).to(device).requires_grad_(True)
        self.focal = torch.tensor([[1000.0 * W / 512, 1000.0 * H / 512]], device=device, requires_grad=True)
        self.pp = torch.tensor([[W / 2.0, H / 2.0]], device=device, requires_grad=True)
        # camera-to-world convention of the reference: x and y flipped (OptimGarmentNetwork.py:1041-1045)
        self.R = torch.diag(torch.tensor([-1.0, -1.0, 1.0])).view(1, 3, 3).to(device)
        self.T = torch.tensor([[0.0, 0.0, 3.0]], device=device, requires_grad=True)
        # the reference stores the camera rotation as a quaternion (dataset/dataset.py:232-235); (0,0,0,1) is R above
        self.quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=device)
        self.shape = torch.zeros(1, 10, device=device)                      # SMPL betas: not used by the loop
        self.frame_num = n_frames
        # images: smooth random colour / normal fields in [-1,1], shared by a few image slots to bound memory
        k = image_frames or min(n_frames, 8)
        low = torch.randn(k, 6, 16, 16, generator=g)
        img = torch.tanh(F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)).permute(0, 2, 3, 1)
        self.img = img[..., :3].contiguous().to(device)
        self.normal = F.normalize(img[..., 3:], dim=-1).contiguous().to(device)
        self.n_img = k

    # -- the attribute names utils.save_model / load_model of the reference read and write (utils/utils.py:350-420)
    @property
    def camera_params(self):
        return {'focal_length': self.focal, 'princeple_points': self.pp, 'cam2world_coord_quat': self.quat,
                'world2cam_coord_trans': self.T}

    @camera_params.setter
    def camera_params(self, d):
        self.focal, self.pp = d['focal_length'], d['princeple_points']
        self.quat, self.T = d['cam2world_coord_quat'], d['world2cam_coord_trans']
        from .utils import quat2mat
        self.R = quat2mat(self.quat.detach().view(1, 4)).view(1, 3, 3)

    @property
    def conds(self):
        return _CondPair(self)

    def __len__(self):
        return self.F

    def get_grad_parameters(self, frame_ids, device):
        return self.poses[frame_ids], self.trans[frame_ids], self.d_cond[frame_ids], self.rendcond[frame_ids]

    def get_camera_parameters(self, N, device):
        return self.focal, self.pp, self.R, self.T, self.H, self.W

    def get_batchframe_data(self, name, frame_ids, nlen):
        """Windows of `nlen` consecutive frames around each frame id (dataset.py get_batchframe_data)."""
        data = getattr(self, name)
        start = torch.clamp(frame_ids - nlen // 2, 0, max(self.F - nlen, 0))
        idx = start.view(-1, 1) + torch.arange(min(nlen, self.F), device=frame_ids.device).view(1, -1)
        return data[idx], idx

    def learnable_weights(self):
        return [self.poses, self.trans, self.d_cond, self.rendcond, self.focal, self.pp, self.T]

    def images(self, frame_ids):
        sl = frame_ids % self.n_img
        return self.img[sl], self.normal[sl]

    def get_batch(self, frame_ids, mask_keys=('upper', 'bottom')):
        """The `datas` dict of one mini-batch as the reference's DataLoader collates it (dataset/dataset.py:617-680;
        read by OptimGarmentNetwork.forward :1888-1904): img / normal [N,H,W,3], mask and one garment region per entry of
        `mask_keys` [N,H,W] ('upper' / 'bottom', or 'upper_bottom' for a one-piece garment), fl_pts [N, n_curves*M, 2] and
        fl_masks [N, n_curves] when feature lines exist."""
        img, normal = self.images(frame_ids)
        out = {'img': img, 'normal': normal, 'frame_ids': frame_ids}
        masks = [self.garment_masks(g, frame_ids) for g in range(len(self._masks))]
        for name, m in zip(mask_keys, masks):
            out[name] = m > 0
        out['mask'] = torch.stack(masks, 0).amax(0)
        if hasattr(self, 'gt_fl_pts'):
            slot = frame_ids % self.n_img
            out['fl_pts'], out['fl_masks'] = self.gt_fl_pts[slot], self.fl_masks[slot]
        return out

    def set_garment_silhouettes(self, radii, seed=0):
        """Synthetic ground-truth segmentation: garment g of image slot k is a slightly elliptical disc — the outline
        of a sphere of radius radii[g] seen by this camera, jittered by a few pixels per slot — so that the IoU term of
        the mask loss starts near, but not at, its optimum."""
        g = torch.Generator().manual_seed(seed)
        fx, fy = float(self.focal.detach()[0, 0]), float(self.focal.detach()[0, 1])
        cx, cy, Z = float(self.pp.detach()[0, 0]), float(self.pp.detach()[0, 1]), float(self.T.detach()[0, 2])
        ys, xs = torch.meshgrid(torch.arange(self.H, dtype=torch.float32), torch.arange(self.W, dtype=torch.float32),
                                indexing='ij')
        self._masks = []
        for r in radii:
            rx, ry = fx * r / math.sqrt(Z * Z - r * r), 1.06 * fy * r / math.sqrt(Z * Z - r * r)
            jit = 3.0 * torch.randn(self.n_img, 2, generator=g)
            m = (((xs[None] - cx - jit[:, 0, None, None]) / rx) ** 2
                 + ((ys[None] - cy - jit[:, 1, None, None]) / ry) ** 2 <= 1.0).float()
            self._masks.append(m.to(self.device))

    def garment_masks(self, g_i, frame_ids):
        """Ground-truth garment segmentation of the batch's frames [N,H,W] (datas['upper'] / ['bottom'],
        OptimGarmentNetwork.py:1896-1902)."""
        return self._masks[g_i][frame_ids % self.n_img]


class _CondPair:
    """dataset.conds[0] / [1] = the per-frame deformer / colour codes, assignable (utils/utils.py:386-387)."""

    def __init__(self, ds):
        self.ds = ds

    def __getitem__(self, i):
        return (self.ds.d_cond, self.ds.rendcond)[i]

    def __setitem__(self, i, v):
        setattr(self.ds, ('d_cond', 'rendcond')[i], v)


class HotLoop:
    """The per-frame optimisation inner loop (see module docstring)."""

    def __init__(self, conf, device, n_frames=64, H=512, W=512, stage='coarse', seed=0, resolutions=None,
                 skin_grid=(65, 225, 129), bbox=None, world_size=1, rank=0, curves=False, large_pose=False,
                 dataset=None, skinner_state=None):
        self.conf_all = conf
        self.conf = conf.get_config('loss_' + stage)
        self.device = device
        self.stage = stage
        # The garment set comes from the capture's name like the reference's (model/network.py:187-192,
        # OptimGarmentNetwork.py:141-164): `train.garment_type` -> TEMPLATE_GARMENT -> one SDF net, explicit mesh and deformer
        # code per garment template, in that order; the config wins, a caller's dataset names the capture otherwise.
        self.garment_type = conf.get_string('train.garment_type') if 'train.garment_type' in conf else None
        if self.garment_type is None:
            self.garment_type = getattr(dataset, 'garment_type', None)
        if self.garment_type not in TEMPLATE_GARMENT:
            raise KeyError("train.garment_type = %r is not a capture of utils/constant.py TEMPLATE_GARMENT (%s)"
                           % (self.garment_type, ', '.join(sorted(TEMPLATE_GARMENT))))
        self.garment_names = list(TEMPLATE_GARMENT[self.garment_type])                     # :160
        self.garment_size = len(self.garment_names)                                      # :164
        # one-piece garments are supervised with the union region of the parsing (:152-156, :1894-1905)
        self.is_upper_bottom = bool(conf.get_bool('train.is_upper_bottom')) if 'train.is_upper_bottom' in conf else False
        self.mask_keys = list(MASK_KEYS[self.is_upper_bottom])
        if self.garment_size > 2:
            raise NotImplementedError('only support less or equal than 2 garment_type')    # :934
        if self.garment_size > len(self.mask_keys):
            raise ValueError("train.is_upper_bottom supervises ONE garment with the union region; %s has %d garments"
                             % (self.garment_type, self.garment_size))
        self.isfine = False                                   # train.py:239,312 set it with the fine stage
        torch.manual_seed(seed)
        mult = conf.get_int('sdf_net.multires')
        # body + one SDF net per garment (model/network.py:188-199); different radii so the meshes differ
        self.sdf = getTmpSdf(device, mult, bias=0.5)
        self.garment_nets = torch.nn.ModuleList([getTmpSdf(device, conf.get_int('garment_sdf_net.multires'), bias=b)
                                                 for b in (0.55, 0.45)[:self.garment_size]])
        skinner = None
        if skinner_state is not None:
            # the reference's `initial_skinner_<pose type>.pth` (model/network.py:225-236): the baked skinning volume (or the
            # FITE-diffused one beside the capture), its box, the rest skeleton, the A-pose inverse transforms, the SMPL
            # template in canonical space; the canonical box is the volume's box plus the reference's margins (:291)
            st = skinner_state
            skinner = LBSkinner(st['ws'], st['bmins'], st['bmaxs'], st['Js'], st['parents'], init_pose=st['init_pose'],
                                align_corners=False, extra_trans=st.get('extra_trans'), bbox_center=st['bbox_center'],
                                bbox_extend=st['bbox_extend'])
            self.tmpBodyVs = torch.as_tensor(st['tmpBodyVs']).float().to(device)
            self.tmpBodyFs = torch.as_tensor(st['tmpBodyFs']).long().to(device)
            if bbox is None:
                lo, hi = skinner.bbox_size()
                bbox = (tuple(float(v) for v in lo.view(-1)), tuple(float(v) for v in hi.view(-1)))
        if bbox is None:
            # The geometric initialisation gives only approximately the nominal sphere radius.  Size the canonical
            # box from the measured radius so that the coarse pyramid yields about the vertex counts the reference
            # reports for its garment meshes ("8w, 7w": OptimGarmentNetwork.py:691).
            r = max(_zero_level_radius(n, device) for n in self.garment_nets)
            h = 1.45 * r
            bbox = ((-h, -1.44 * h, -h), (h, 1.44 * h, h))
        bmin, bmax = bbox
        if skinner is None:
            skinner = self._synthetic_skinner(bmin, bmax, skin_grid)
        self.deformer = CompositeDeformer([getTranslatorNet(device, conf.get_config('mlp_deformer')),
                                           skinner.to(device)])
        self.netRender = getRenderNet(device, conf.get_config('render_net'))
        if dataset is None:
            dataset = SyntheticFrames(n_frames, self.garment_size, H, W, device, seed=seed + 2,
                                      condlen=conf.get_int('mlp_deformer.condlen'),
                                      rendlen=conf.get_int('render_net.condlen'))
            dataset.set_garment_silhouettes([_zero_level_radius(n, device) for n in self.garment_nets], seed=seed + 3)
        self.dataset = dataset                     # getOptNet hands over the caller's dataset (model/network.py:352)
        self._datas = None                         # the mini-batch dict of forward(datas, ...), when a caller passes one
        # large-pose fitting (OptimGarmentNetwork_Large_Pose.py:122-137): the surfaces are frozen, only the deformation,
        # the per-frame tensors, the colour net and the camera move
        self.large_pose = bool(large_pose)
        if self.large_pose:
            self.freeze_sdf()
        res = resolutions if resolutions is not None else RESOLUTIONS[stage]
        self.engine = Seg3dLossless(query_func=None, b_min=list(bmin), b_max=list(bmax), resolutions=res,
                                    align_corners=False, balance_value=0.0, use_cuda_impl=True, faster=False).to(device)
        self.remesh_intersect = conf.get_int(f'train.{stage}.point_render.remesh_intersect')
        self.pc_radius = conf.get_float(f'train.{stage}.point_render.radius')        # OptimNetwork.py:87-93
        self.batch_size = conf.get_int(f'train.{stage}.point_render.batch_size')
        self.sample_pix = conf.get_int('train.sample_pix_num')
        self.sdfShrinkRadius = 0.0
        self.forward_time = 0
        self.opt_times = 0.0
        self.next_conf = self.next_train_conf = None
        self.body_vs = self.body_fs = None
        self.garment_vs, self.garment_fs = [], []
        # 30-frame windows (OptimGarmentNetwork.py:1222); a caller's dataset insists on windows shorter than the video
        # (dataset/dataset.py:442), the synthetic frames allow a window as long as it
        nlen = min(30, n_frames if hasattr(dataset, 'F') else n_frames - 1)
        self.dctnull = dct_nullspace(nlen, min(10, max(nlen // 3, 1)), device)
        self.info = {}
        self.world_size, self.rank = world_size, rank
        # feature-curve branch (project_2d_loss, SURVEY.md §8f "next" row 3): off unless asked for
        self.curves = bool(curves)
        if self.curves:
            self._init_curves(seed + 4)
        self.optimizer = self.rebuild_optimizer()
        cams = self._cameras()
        self.angThred = cams.angThreshold(0.5)                                   # OptimNetwork.py:65

    @staticmethod
    def _synthetic_skinner(bmin, bmax, skin_grid):
        """Synthetic rig: a 24-joint skeleton that fits the canonical box and SMOOTH blend weights (softmax of the squared
        distance to the joints), like the diffused SMPL weights the reference bakes into its volume (model/Deformer.py:
        289-330).  Smoothness matters for the workload: neighbouring surface points must stay neighbours after skinning,
        or the rasterised first hits are no starting points for the root finder."""
        D, Hh, Ww = skin_grid
        Js = _skeleton(0.5 * (bmax[1] - bmin[1]) / 1.44 / 1.45)
        axes = [torch.linspace(bmin[i], bmax[i], n + 1)[:-1] + 0.5 * (bmax[i] - bmin[i]) / n
                for i, n in ((2, D), (1, Hh), (0, Ww))]                       # voxel centres, align_corners=False
        zz, yy, xx = torch.meshgrid(*axes, indexing='ij')
        vox = torch.stack([xx, yy, zz], dim=-1)                                # [D,H,W,3] (x,y,z)
        d2 = (torch.cdist(vox.view(-1, 3), Js) ** 2).view(D, Hh, Ww, 24)
        sigma = 0.12 * (bmax[1] - bmin[1]) / 1.44 / 1.45
        ws = torch.softmax(-d2 / (2.0 * sigma * sigma), dim=-1).permute(3, 0, 1, 2).unsqueeze(0).contiguous()
        return LBSkinner(ws, list(bmin), list(bmax), Js, SMPL_PARENTS, init_pose=_apose(), align_corners=False,
                         bbox_extend=torch.tensor([bmax[i] - bmin[i] for i in range(3)]),
                         bbox_center=torch.tensor([(bmax[i] + bmin[i]) / 2 for i in range(3)]))

    # ------------------------------------------------------------------------------------------ stages / state
    def set_stage(self, stage, resolutions=None):
        """utils.set_hierarchical_config (utils/utils.py:330-348): the batch size and the Seg3dLossless pyramid (same
        box) change NOW; the stage's loss weights, point radius and re-mesh period are only parked in `next_conf` /
        `next_train_conf` and take effect at the next scheduled re-mesh (update_hierarchical_config)."""
        conf = self.conf_all
        self.stage = stage
        self.batch_size = conf.get_int(f'train.{stage}.point_render.batch_size')
        self.next_conf = conf.get_config('loss_' + stage)
        self.next_train_conf = conf.get_config('train.' + stage)
        old = self.engine
        self.engine = Seg3dLossless(query_func=None, b_min=old.b_min.view(-1).tolist(), b_max=old.b_max.view(-1).tolist(),
                                    resolutions=resolutions if resolutions is not None else RESOLUTIONS[stage],
                                    align_corners=False, balance_value=0.0, use_cuda_impl=True,
                                    faster=False).to(self.device)

    def update_hierarchical_config(self):
        """OptimNetwork.update_hierarchical_config (engineer/networks/OptimNetwork.py:79-117), called from
        marching_cube_update (:697): apply the parked stage configuration and restart the re-mesh counter."""
        if getattr(self, 'next_conf', None) is not None:
            self.conf = self.next_conf
            self.forward_time = 0
            self.pc_radius = self.next_train_conf.get_float('point_render.radius')
            self.remesh_intersect = self.next_train_conf.get_int('point_render.remesh_intersect')
            self.sdfShrinkRadius = 0.0
            self.next_conf = self.next_train_conf = None

    def freeze_sdf(self):
        """OptimGarmentNetwork_LargePose.freeze_sdf (:130-137): when fitting large poses the surfaces are not optimised."""
        for net in list(self.garment_nets) + [self.sdf]:
            for para in net.parameters():
                para.requires_grad = False

    # -- ground truth of the mini-batch: the `datas` dict of forward(datas, ...) when given, else the synthetic frames
    def _gt_images(self, frame_ids):
        d = self._datas
        if d is not None:
            return d['img'].to(self.device), (d['normal'].to(self.device) if 'normal' in d else None)
        return self.dataset.images(frame_ids)

    def _gt_garment_mask(self, g_i, frame_ids):
        d = self._datas
        if d is not None:
            return d[self.mask_keys[g_i]].to(self.device).float()         # datas['upper'] / ['bottom'] / ['upper_bottom'] :1894-1905
        return self.dataset.garment_masks(g_i, frame_ids)

    def _gt_feature_lines(self, frame_ids):
        d = self._datas
        if d is not None:
            return d['fl_pts'].to(self.device), d['fl_masks'].to(self.device)            # :1891-1892
        slot = frame_ids % self.dataset.n_img
        return self.dataset.gt_fl_pts[slot], self.dataset.fl_masks[slot]

    def _modules(self):
        mods = {'sdf': self.sdf, 'garment_nets': self.garment_nets, 'deformer': self.deformer,
                'netRender': self.netRender, 'engine': self.engine}
        if getattr(self, 'curves', False):
            mods['inter_free_curve'] = self.inter_free_curve          # optNet.inter_free_curve of the reference
        return mods

    def state_dict(self):
        """Keys as `optNet.state_dict()` of the reference names them (getOptNet, model/network.py:182-361): `sdf.*`,
        `garment_nets.{i}.*`, `deformer.defs.0.*` (offset MLP), `deformer.defs.1.*` (skinner buffers), `netRender.*`,
        `engine.*`, and `inter_free_curve.*` when the feature-curve branch is on."""
        out = {}
        for prefix, mod in self._modules().items():
            for k, v in mod.state_dict().items():
                out[prefix + '.' + k] = v
        for name in self._BUFFERS:
            if getattr(self, name, None) is not None:
                out[name] = getattr(self, name)
        return out

    # buffers the reference registers on the optimisation object itself: the SMPL template (:128-129) and, once the pre-fit has run,
    # the body mesh extracted from the pre-fitted SDF (`load_init_sdf_vertices`, :176-178) — both travel in `model_state_dict`
    _BUFFERS = ('tmpBodyVs', 'tmpBodyFs', 'tmp_sdf_body_vs', 'tmp_sdf_face_vs')

    def load_state_dict(self, sd, strict=True):
        missing, unexpected = [], set(sd.keys())
        for name in self._BUFFERS:
            if name in sd:
                setattr(self, name, sd[name].to(self.device))
                unexpected.discard(name)
        for prefix, mod in self._modules().items():
            sub = {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + '.')}
            unexpected -= {prefix + '.' + k for k in sub}
            res = mod.load_state_dict(sub, strict=False)
            missing += [prefix + '.' + k for k in res.missing_keys]
            unexpected |= {prefix + '.' + k for k in res.unexpected_keys}
        if strict and (missing or unexpected):
            raise RuntimeError(f'load_state_dict: missing {missing}, unexpected {sorted(unexpected)}')
        return missing, sorted(unexpected)

    def to(self, device):
        return self

    def parameters(self):
        for mod in (self.garment_nets, self.deformer, self.netRender):
            yield from mod.parameters()

    # ------------------------------------------------------------------------------------------ helpers
    @contextlib.contextmanager
    def _phase(self, name):
        """Wall time per phase of an iteration into self.phase_ms when RECMV_TIMING=1 (adds device syncs)."""
        if os.environ.get('RECMV_HOST_TRACE') == '1' and torch.cuda.is_available():
            # no synchronisation: host interval of the phase + HIP events on the stream the phase runs on (tools/phase_overlap.py)
            rec = self.__dict__.setdefault('phase_trace', [])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            yield
            e1.record()
            rec.append((name, t0, time.perf_counter(), e0, e1))
            return
        if os.environ.get('RECMV_TIMING', '0') != '1':
            yield
            return
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.nvtx.range_push('recmv:' + name)      # roctx range: rocprofv3 --marker-trace
            if os.environ.get('RECMV_PHASE_LOG'):            # range order, for tools/prof_phases.py
                with open(os.environ['RECMV_PHASE_LOG'], 'a') as fh:
                    fh.write(name + '\n')
        t0 = time.perf_counter()
        yield
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.nvtx.range_pop()
        acc = self.__dict__.setdefault('phase_ms', {})
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

    def _cameras(self):
        focals, pps, Rs, Ts, H, W = self.dataset.get_camera_parameters(1, self.device)
        return RectifiedPerspectiveCameras(focals, pps, Rs, Ts, image_size=[(W, H)])

    def get_grad_parameters(self, frame_ids, device):
        poses, trans, d_cond, rendcond = self.dataset.get_grad_parameters(frame_ids, device)
        split = [d_cond.shape[-1] // (self.garment_size + 1)] * (self.garment_size + 1)
        return torch.split(d_cond, split, dim=-1), poses, trans, rendcond

    def shared_parameters(self):
        """Tensors whose gradients are all-reduced across frame-sharded ranks (list 1)."""
        return [p for group in self.optimizer.param_groups for p in group['params']]

    def early_shared_parameters(self):
        """The shared tensors propagateTmpPsGrad does not touch: the colour net and the per-frame colour codes."""
        early = [p for p in self.netRender.parameters() if p.requires_grad]
        rend = getattr(self.dataset, 'rendcond', None)
        if rend is None and hasattr(self.dataset, 'conds'):
            rend = self.dataset.conds[1]
        if rend is not None and rend.requires_grad:
            early.append(rend)
        ids = {id(p) for p in self.shared_parameters()}
        return [p for p in early if id(p) in ids]

    # ------------------------------------------------------------------------------------------ MC path
    def discretizeSDF(self, ratio, engine=None, balance_value=0.):
        """OptimGarmentNetwork.py:581-618."""
        engine = engine or self.engine
        trace = getattr(self, 'remesh_trace', None)          # a list: bench.py's configs[2] block asks for the split of a re-mesh

        def span(name):
            """(event, event) bracket on the current stream + the host's interval, appended to the trace; no synchronisation."""
            if trace is None or not torch.cuda.is_available():
                return contextlib.nullcontext()

            @contextlib.contextmanager
            def cm():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                yield
                e1.record()
                trace.append((name, e0, e1, t0, time.perf_counter()))
            return cm()

        def query_of(net):
            def query(points):
                with torch.no_grad(), span('query'):
                    # only the SDF value is read here: skip the 256 render-feature rows of the last layer
                    return net.forward(points.reshape(-1, 3), ratio, features=False).reshape(1, 1, -1)
            return query

        nets = [self.sdf] + list(self.garment_nets)
        engine.balance_value = balance_value
        # A net whose parameters have not changed since its last extraction on this grid gives the same volume and the same mesh,
        # bit for bit (the pyramid and marching cubes are deterministic): it is not extracted again.  In the optimisation stage that
        # is the BODY net (no loss term reaches it: SURVEY.md §8e — a quarter of a re-mesh's queries), in the large-pose stage all
        # three (freeze_sdf).  The reference re-extracts them every time (:593-617).  RECMV_REMESH_CACHE=0: always extract (A/B).
        ws = None if ratio is None else (ratio if isinstance(ratio, (int, float)) else ratio.get('sdfRatio'))
        grid = (tuple(tuple(int(v) for v in r) for r in engine.resolutions), tuple(engine.b_min.view(-1).tolist()),
                tuple(engine.b_max.view(-1).tolist()), float(balance_value), ws)
        cache = self.__dict__.setdefault('_remesh_cache', {})
        keys = [(grid,) + tuple((q.data_ptr(), q._version) for q in net.parameters()) for net in nets]
        use_cache = os.environ.get('RECMV_REMESH_CACHE', '1') != '0' and torch.device(self.device).type == 'cuda'
        todo = [i for i, k in enumerate(keys) if not (use_cache and i in cache and cache[i][0] == k)]
        results = {i: cache[i][1] for i in range(len(nets)) if i not in todo}
        if todo:
            # the pyramids of the nets that moved run level by level in lockstep (Seg3dLossless.forward_multi)
            with span('pyramid'):
                volumes = engine.forward_multi([query_of(nets[i]) for i in todo])
            with span('mc'):
                vs, fs = self._extract(volumes, engine, balance_value)
            for i, v, f in zip(todo, vs, fs):
                results[i] = (v, f)
                if use_cache:
                    cache[i] = (keys[i], (v.detach().clone(), f))
        # (garment vertices become leaves of the explicit-mesh SGD: a cached extraction hands out its own copy)
        out = [results[i] if i in todo else (results[i][0].clone(), results[i][1]) for i in range(len(nets))]
        return [v for v, _ in out], [f for _, f in out]

    def _extract(self, volumes, engine, balance_value):
        vols = [sdfs[0, 0].permute(2, 1, 0).contiguous() for sdfs in volumes]
        # all nets' extractions in one set of launches and one counter read-back (MCGpu.mc_gpu_multi; on the CPU port and for a
        # grid's first extraction: one mc_gpu per net, the reference's form)
        multi = getattr(MCGpu, 'mc_gpu_multi', None) if vols[0].is_cuda else None
        if multi is not None:
            res = multi(vols, engine.spacing_x, engine.spacing_y, engine.spacing_z, engine.bx, engine.by, engine.bz, balance_value)
        else:
            res = [MCGpu.mc_gpu(v, engine.spacing_x, engine.spacing_y, engine.spacing_z, engine.bx, engine.by, engine.bz,
                                balance_value) for v in vols]
        return [r[0] for r in res], [r[1] for r in res]

    def reserve_memory(self, megabytes=None):
        """Park one large free block per stream of the loop in torch's caching allocator (allocate, release: the block stays cached and
        later requests are split off it).  A re-mesh re-sizes every vertex-sized buffer; whatever does not fit a cached block goes to
        hipMalloc in the middle of that step — 3-4 calls, and on a host that is busy (a box in its first minutes) they were +30 ms on
        the re-mesh step (profiles/r06_remesh_first_process.txt).  `megabytes`: per stream in the order main, ray, curve, second garment
        (default RECMV_RESERVE_MB or 6144,3072,1024,3072 — a few percent of the 288 GB); 0 entries are skipped.  Call once, after the
        first mesh exists; returns the bytes parked."""
        if torch.device(self.device).type != 'cuda':
            return 0
        if megabytes is None:
            megabytes = [int(v) for v in os.environ.get('RECMV_RESERVE_MB', '6144,3072,1024,3072').split(',') if v.strip()]
        if getattr(self, '_surface_stream', None) is None:
            self._surface_stream = _make_stream(self.device)
        if getattr(self, '_curve_stream', None) is None:
            self._curve_stream = _make_stream(self.device)
        from .utils.FindSurfacePs import _streams
        streams = [torch.cuda.current_stream(self.device), self._surface_stream, self._curve_stream]
        streams += _streams(torch.device(self.device), max(self.garment_size - 1, 0))
        total = 0
        for st, mb in zip(streams, megabytes):
            if mb > 0:
                with torch.cuda.stream(st):
                    block = torch.empty(mb << 20, dtype=torch.uint8, device=self.device)
                total += block.numel()
                del block
        return total

    def marching_cube_update(self, ratio):
        """OptimGarmentNetwork.py:678-740 (openmesh vertex->face tables are never read by the loop: dropped)."""
        trace = getattr(self, 'remesh_trace', None)
        t_all = None
        if trace is not None and torch.cuda.is_available():
            t_all = (torch.cuda.Event(enable_timing=True), time.perf_counter())
            t_all[0].record()
        vs_list, fs_list = self.discretizeSDF(ratio, None, -self.sdfShrinkRadius)
        try:
            self._marching_cube_handover(vs_list, fs_list)
        finally:
            if t_all is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                trace.append(('remesh', t_all[0], e1, t_all[1], time.perf_counter()))

    def _marching_cube_handover(self, vs_list, fs_list):
        self.body_vs, self.body_fs = vs_list[0], fs_list[0]
        self.garment_vs, self.garment_fs = vs_list[1:], fs_list[1:]
        self.update_hierarchical_config()                                                  # :697
        if self.body_vs.shape[0] == 0:
            raise AssertionError('tmp sdf vanished...')
        if any(v.shape[0] == 0 for v in self.garment_vs):
            raise AssertionError('a garment sdf has no zero level inside the box (diverged optimisation?)')
        for v in self.garment_vs:
            v.requires_grad = True
        self.garment_optimizer = torch.optim.SGD(self.garment_vs, lr=0.05, momentum=0.9)
        if getattr(self, 'curves', False):
            self.fl_optimizer = torch.optim.AdamW(self.inter_free_curve.parameters(), lr=1e-4)              # :712

    # ------------------------------------------------------------------------------------------ start-up stage
    def initializeSDF(self, network, optimizer, sche, batch_size, nepochs, device, vs, ns, with_normals, save_name, log=print):
        """OptimGarmentNetwork.py:387-443 — fit an SDF net to an oriented point cloud before the loop (IGR): |f| on the
        surface points, the eikonal term on points scattered around them and in the box, and (with normals) ∇f against the
        point normals; `nepochs` passes over shuffled mini-batches of `batch_size` points, the scheduler stepped per epoch,
        the state dict written to `save_name`.  Positional-encoding weights are off (ratio -1), as in the reference.
        On the GPU the value and ∇ₓf come from one jet pass per batch (csrc/mlp_jet.hip) instead of a double backward."""
        for epoch in range(1, nepochs + 1):
            permute = torch.randperm(vs.shape[0])
            evs, ens = torch.split(vs[permute], batch_size), torch.split(ns[permute], batch_size)
            for data_index, (mnfld_pnts, normals) in enumerate(zip(evs, ens)):
                mnfld_pnts = mnfld_pnts.to(device)
                nonmnfld_pnts = utils.sample_points(mnfld_pnts, 1.8, 0.01)      # local (sigma 0.01) + uniform in the box
                mnfld_pnts.requires_grad_()
                nonmnfld_pnts.requires_grad_()
                mnfld_pred = network(mnfld_pnts, -1, jet=True, features=False)
                mnfld_grad = network.gradient(mnfld_pnts, mnfld_pred)
                nonmnfld_pred = network(nonmnfld_pnts, -1, jet=True, features=False)
                nonmnfld_grad = network.gradient(nonmnfld_pnts, nonmnfld_pred)
                mnfld_loss = mnfld_pred.abs().mean()
                grad_loss = ((nonmnfld_grad.norm(2, dim=-1) - 1) ** 2).mean()
                loss = mnfld_loss + 0.1 * grad_loss
                if with_normals:
                    normals_loss = (mnfld_grad - normals.to(device).view(-1, 3)).abs().norm(2, dim=1).mean()
                    loss = loss + 1.0 * normals_loss
                else:
                    normals_loss = torch.zeros(1)
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                if log is not None and data_index == len(evs) - 1:
                    log('Train Epoch: {}\tTrain Loss: {:.6f}\tManifold loss: {:.6f}\tGrad loss: {:.6f}\tNormals Loss: {:.6f}'
                        .format(epoch, loss.item(), mnfld_loss.item(), grad_loss.item(), normals_loss.item()))
            sche.step()
        torch.save(network.state_dict(), save_name)

    def initializeFL(self, dataloader, n_epochs, device, save_mesh_name):
        """OptimGarmentNetwork.py:470-486 — register the template feature lines (`self.garment_fl_templates`: {line name:
        mesh}) to the annotated frames; writes `<folder of save_mesh_name>/fl_init/init_trans_matrix.pth`."""
        from .engineer.core.fl_optimizer import scale_rigid_optimizer
        from .utils.constant import FL_INFOS
        save_fl_path = os.path.join(os.path.dirname(save_mesh_name), 'fl_init')
        os.makedirs(save_fl_path, exist_ok=True)
        self._ensure_body_template()
        return scale_rigid_optimizer(self.deformer.defs[1], self.garment_fl_templates, (self.tmpBodyVs, self.tmpBodyFs), None,
                                     self.dataset, dataloader, save_fl_path, FL_INFOS[self.garment_type], device=device)

    def initializeTmpSDF(self, nepochs, save_name, with_normals=False, dataloader=None, body_points=None,
                         garment_points=None, fl_templates=None, log=print):
        """OptimGarmentNetwork.py:490-578 — the pre-fit train.py runs when there is no `initial_sdf_idr_*.pth` yet: register
        the template feature lines (when `fl_templates` and a loader are given), then fit the body net to the SMPL
        template and every garment net to its closed garment template, each for `nepochs` epochs of 5000-point batches
        (Adam 5e-3, halved every 500 epochs), saved under the reference's file names (`..sdf..` / `..sdf_<garment>..`).

        The reference cuts the garment templates and their feature lines out of its SMPL garment assets
        (`../smpl_clothes_template`, laplacian registration to the registered lines, hole closing — mesh tools outside this
        package); here the caller passes the result: `body_points` = (vertices, normals or None) of the canonical body
        (default `self.tmpBodyVs` / `self.tmpBodyNs`), `garment_points` = one (vertices, normals) per garment."""
        if fl_templates is not None:
            self.garment_fl_templates = fl_templates
            if dataloader is not None:
                self.initializeFL(dataloader, nepochs, self.device, save_name)
        if body_points is None:
            body_points = (getattr(self, 'tmpBodyVs', None), getattr(self, 'tmpBodyNs', None))
        if body_points[0] is None or garment_points is None or len(garment_points) != self.garment_size:
            raise ValueError("initializeTmpSDF needs the canonical body points and one (vertices, normals) pair per garment "
                             "(%s): the reference builds them from its SMPL garment template assets" % ', '.join(self.garment_names))

        def fit(network, points, name):
            vs, ns = points[0].to(self.device), points[1]
            use_normals = bool(with_normals) and ns is not None
            ns = ns.to(self.device) if ns is not None else torch.ones_like(vs) / math.sqrt(3)
            for p in network.parameters():
                p.requires_grad_(True)
            optimizer = torch.optim.Adam([{"params": network.parameters(), "lr": 0.005, "weight_decay": 0}])
            sche = torch.optim.lr_scheduler.StepLR(optimizer, 500, 0.5)
            self.initializeSDF(network, optimizer, sche, 5000, nepochs, self.device, vs, ns, use_normals, name, log=log)

        log and log('Fitting_body_net!')
        fit(self.sdf, body_points, save_name)
        for g_name, points, net in zip(self.garment_names, garment_points, self.garment_nets):
            log and log('Fitting_garment_net {}!'.format(g_name))
            fit(net, points, save_name.replace("sdf", 'sdf_{}'.format(g_name)))                  # :563-576
        if self.large_pose:
            self.freeze_sdf()

    def load_init_sdf_vertices(self, verts, faces=None):
        """OptimGarmentNetwork.py:176-178 — keep the body mesh extracted from the pre-fitted SDF (train.py:198); accepts the
        (vertices, faces) pair or a mesh object with `.vertices` / `.faces`."""
        if faces is None:
            verts, faces = verts.vertices, verts.faces
        self.tmp_sdf_body_vs = torch.as_tensor(verts).float().to(self.device)
        self.tmp_sdf_face_vs = torch.as_tensor(faces).float().to(self.device)

    def _deform_garments(self, N, frame_ids, ratio):
        """The deformed garment vertices of this iteration, with their autograd graph (mask_loss :910).  The reference
        evaluates the same expression a second time per garment inside fl_visible_by_body_zbuff (:1396) — same
        vertices, same deformer parameters, only depths are read from it — so the curve branch shares this one."""
        hit = getattr(self, '_def_cache', None)         # cleared at the start of every forward()
        if hit is not None:
            return hit
        d_cond_list, poses, trans, _ = self.get_grad_parameters(frame_ids, self.device)
        def_vs = [self.deformer(gv[None, :, :].expand(N, -1, 3), [d_cond_list[g_i + 1], [poses, trans]], ratio=ratio,
                                offset_type=name)
                  for g_i, (gv, name) in enumerate(zip(self.garment_vs, self.garment_names))]
        self._def_cache = def_vs
        self._def_params = (d_cond_list, poses, trans)       # the mask loss differentiates the same graph: the same gathers
        self._frag_cache = {}
        return def_vs

    def _garment_fragments(self, g_i, def_v, cameras):
        """First-hit fragments of garment g_i's deformed meshes, rasterised once per iteration (the z-buffer of the
        curve branch :1399 and the surface points of find_surface_ps :767 are the same image).  The two users run on
        different streams: the cache keeps the event recorded behind the rasterisation, a reader on another stream waits
        for it."""
        cache = getattr(self, '_frag_cache', None)
        if cache is None:
            cache = self._frag_cache = {}
        cuda = torch.device(self.device).type == 'cuda'
        if g_i not in cache:
            rast = raster.MeshRasterizer(cameras, (self.dataset.H, self.dataset.W), blur_radius=0.,
                                         perspective_correct=True, cull_backfaces=False)           # :2336-2347
            with torch.no_grad():
                frags = rast(def_v.detach(), self.garment_fs[g_i])
            ev = None
            if cuda:
                ev = torch.cuda.Event()
                ev.record()
            cache[g_i] = (frags, ev, torch.cuda.current_stream(self.device) if cuda else None)
        frags, ev, producer = cache[g_i]
        if ev is not None and torch.cuda.current_stream(self.device) != producer:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return frags

    # ------------------------------------------------------------------------------------------ feature curves
    def _ensure_body_template(self):
        """`tmpBodyVs` / `tmpBodyFs`: the SMPL template in canonical space the body z-buffer tests rasterise (6890 vertices in
        the reference).  A stored skinner file brings it along; otherwise a coarse extraction of the body SDF stands in."""
        if getattr(self, 'tmpBodyVs', None) is not None:
            return
        dev, res = self.device, 41
        lo, hi = self.engine.b_min.view(-1), self.engine.b_max.view(-1)
        ax = [torch.linspace(float(lo[i]), float(hi[i]), res, device=dev) for i in range(3)]
        X, Y, Z = torch.meshgrid(*ax, indexing='ij')
        with torch.no_grad():
            vol = self.sdf(torch.stack([X, Y, Z], -1).view(-1, 3), 1.0, features=False).view(res, res, res).contiguous()
        step = [float(a[1] - a[0]) for a in ax]
        self.tmpBodyVs, self.tmpBodyFs = MCGpu.mc_gpu(vol, step[0], step[1], step[2], float(ax[0][0]), float(ax[1][0]),
                                                      float(ax[2][0]), 0.0)

    def _feature_line_tables(self, available=None):
        """({garment: its feature lines}, [all lines in the order of the dataset's `fl_pts` / `fl_masks` columns]):
        FL_EXTRACT per garment template (OptimGarmentNetwork.py:1568, :1620) and FL_INFOS of the capture (:161; the order
        `deform_feature_line` splits the 2-D ground truth in, :1539-1561).  `available`: restrict to the lines a registration
        actually produced."""
        extract = {g: [n for n in fl.FL_EXTRACT[g] if available is None or n in available] for g in self.garment_names}
        used = [n for g in self.garment_names for n in extract[g]]
        infos = FL_INFOS.get(self.garment_type, [])
        names = list(infos) if set(used) <= set(infos) else used        # ('dance' & co. list a garment name there: unusable)
        if available is not None:
            names = [n for n in names if n in available]
        return extract, names

    def _init_curves(self, seed, samples=200, gt_samples=100):
        """Synthetic stand-in for `align_fl` + the dataset's 2-D feature lines (OptimGarmentNetwork.py:3380-3546,
        dataset/dataset.py:113-155): closed rings on the initial garment spheres as canonical curves, the same rings
        shrunk onto the body as their canonical-SMPL counterparts, a coarse body mesh as the SMPL template, and the
        projected rings (+ pixel noise) as per-frame 2-D ground truth."""
        dev = self.device
        g = torch.Generator().manual_seed(seed)
        radii = [_zero_level_radius(n, dev) for n in self.garment_nets]
        self.fl_extract, self.fl_names = self._feature_line_tables()
        t = torch.linspace(0, 2 * math.pi, samples + 1)[:-1]
        ring = {}
        for name, r in zip(self.garment_names, radii):
            for n in self.fl_extract[name]:
                if n == 'neck':
                    y = 0.75 * r
                    rho = math.sqrt(r * r - y * y)
                    p = torch.stack([rho * torch.cos(t), torch.full_like(t, y), rho * torch.sin(t)], -1)
                elif n in ('upper_bottom', 'bottom_curve'):
                    y = -0.6 * r
                    rho = math.sqrt(r * r - y * y)
                    p = torch.stack([rho * torch.cos(t), torch.full_like(t, y), rho * torch.sin(t)], -1)
                elif n in ('left_cuff', 'right_cuff'):
                    x = (0.8 if n == 'left_cuff' else -0.8) * r
                    rho = math.sqrt(r * r - x * x)
                    p = torch.stack([torch.full_like(t, x), rho * torch.cos(t), rho * torch.sin(t)], -1)
                else:                                                   # left_pant / right_pant
                    cx = (0.35 if n == 'left_pant' else -0.35) * r
                    rho = 0.3 * r
                    y = -math.sqrt(max(r * r - (abs(cx) + rho) ** 2, 0.0)) * 0.9
                    p = torch.stack([cx + rho * torch.cos(t), torch.full_like(t, y), rho * torch.sin(t)], -1)
                ring[n] = p.float()
        curves_list = [ring[n] for n in self.fl_names]
        # canonical-SMPL counterparts: the same rings pulled radially onto the body's zero level, so that the body
        # z-buffer test sees them ON the surface where it faces the camera and a body-thickness behind it elsewhere
        r_body = _zero_level_radius(self.sdf, dev)
        owner = {n: r for name, r in zip(self.garment_names, radii) for n in self.fl_extract[name]}
        for n in self.fl_names:                    # (a line of the capture no garment of the set carries: on the first one)
            if n not in ring:
                ring[n], owner[n] = ring[self.fl_extract[self.garment_names[0]][0]].clone(), radii[0]
        smpl_list = [ring[n] * (r_body / owner[n]) for n in self.fl_names]
        self.inter_free_curve = fl.Intersect_Free_Curve(curves_list, smpl_list, self.fl_names).to(dev)
        self._ensure_body_template()
        # 2-D ground truth per image slot: the rings seen without articulation, jittered by a pixel
        cams = self._cameras()
        idx = torch.linspace(0, samples - 1, gt_samples).long()
        gt = torch.stack([cams.project(c[idx].to(dev)) for c in curves_list], 0)            # [L,M,2]
        if hasattr(self.dataset, 'n_img'):
            k = self.dataset.n_img
            noise = torch.randn(k, gt.shape[0] * gt_samples, 2, generator=g).to(dev)
            self.dataset.gt_fl_pts = (gt.reshape(1, -1, 2) + noise).detach()                     # [k, L*M, 2]
            self.dataset.fl_masks = torch.ones(k, len(self.fl_names), device=dev)
            self.dataset.fl_weights = {n: 1.0 for n in self.fl_names}
        elif not hasattr(self.dataset, 'fl_weights'):
            # a caller's dataset (recmv.dataset.SceneDataset, or the reference's): its 2-D feature lines arrive with every
            # mini-batch (`datas['fl_pts']`), its per-line weights come from `area_size_statistic`
            self.dataset.fl_weights = {n: 1.0 for n in self.fl_names}
        self.fl_optimizer = torch.optim.AdamW(self.inter_free_curve.parameters(), lr=1e-4)

    def fl_visible_by_body_zbuff(self, cameras, d_cond, smpl_conds, ratio, def_fl_vs, cano_smpl_verts_list, g_i,
                                 garment_name, N):
        """OptimGarmentNetwork.py:1374-1448: [N,P,2] = how far each deformed curve sample lies behind the rasterised
        garment surface, and how far its canonical-SMPL counterpart lies behind the rasterised body."""
        H, W = self.dataset.H, self.dataset.W
        rast = raster.MeshRasterizer(cameras, (H, W))
        def_garment_vs = self._shared_def_vs[g_i].detach()   # only depths / comparisons are read (:1396-1403 detach)
        gfrags = self._garment_fragments(g_i, def_garment_vs, cameras)
        with torch.no_grad():
            fl_cat = torch.cat(def_fl_vs, dim=1).detach()
            garment_check = fl.surface_depth_check(cameras, (W, H), gfrags.zbuf, def_garment_vs, fl_cat)
            # (rows of the skinner are independent: the lines' canonical-SMPL points as one batch, :1424-1433)
            def_smpl_fl = self.deformer.defs[1](torch.cat([v.reshape(-1, 3) for v in cano_smpl_verts_list], dim=0)
                                                .view(1, -1, 3).expand(N, -1, 3), smpl_conds)
            # the posed body and its z-buffer are the same for every garment of the iteration (:1436-1446 redoes them)
            hit = self._frag_cache.get('body')
            if hit is None:
                body = self.deformer.defs[1](self.tmpBodyVs.view(1, -1, 3).expand(N, -1, 3), smpl_conds)
                bfrags = rast(body, self.tmpBodyFs)
                ev = torch.cuda.Event() if body.is_cuda else None
                if ev is not None:
                    ev.record()
                self._frag_cache['body'] = (body, bfrags, ev, L_raw_stream(body))
            else:
                body, bfrags, ev, sid = hit
                if ev is not None and L_raw_stream(body) != sid:
                    torch.cuda.current_stream(body.device).wait_event(ev)
            smpl_check = fl.surface_depth_check(cameras, (W, H), bfrags.zbuf, body, def_smpl_fl)
        return torch.stack([garment_check, smpl_check], dim=-1)

    def compute_fl_proj_loss(self, def_fl_meshes, check_values, fl_masks, gt_fl_pts, garment_name, fl_vs_split, cameras):
        """OptimGarmentNetwork.py:1605-1711 (fl_visible_method = zbuff): project the deformed curve samples, keep those
        whose canonical-SMPL counterpart is at most ZBUF_THRESHOLD behind the body surface and whose feature line is
        labelled in the frame, chamfer them against the frame's 2-D curve, add the curve regulariser."""
        conf = self.conf
        names = self.fl_extract[garment_name]
        n_fl = len(names)
        H, W = self.dataset.H, self.dataset.W
        def_fl_verts = torch.cat(def_fl_meshes, dim=1)
        screen_pts = cameras.transform_points_screen(def_fl_verts, (W, H))
        thr = torch.cat([torch.full((n_s,), fl.ZBUF_THRESHOLD[n], device=def_fl_verts.device)
                         for n, n_s in zip(names, fl_vs_split)]).view(1, -1, 1)
        body_visible = (check_values < thr)[..., 1]                                       # :1644-1648
        gt_list = list(torch.split(gt_fl_pts, [gt_fl_pts.shape[1] // n_fl for _ in range(n_fl)], dim=1))
        screen_list = list(torch.split(screen_pts, fl_vs_split, dim=1))
        vis_list = list(torch.split(body_visible, fl_vs_split, dim=1))
        visible_masks = []
        for i, (pts, vis) in enumerate(zip(screen_list, vis_list)):
            fl_mask = fl_masks[:, None, i:i + 1].expand_as(pts)
            visible_masks.append(torch.logical_and(fl_mask, vis[..., None].expand_as(pts)))
        weights = [self.dataset.fl_weights[n] for n in names]
        fl_loss = fl.fl_proj_loss(screen_list, gt_list, visible_masks, weights) * (
            conf.get_float('fl_weight.weight') if 'fl_weight.weight' in conf else 1.)
        # The regulariser does not depend on the garment: its curvature term is a function of the curve parameters alone and its
        # centre term is multiplied by zero whatever `fl_masks` says (garment_structure.py:141) — the reference evaluates it once per
        # garment (and the curves once more inside each time); here once per iteration on the curves project_2d_loss already has.
        # The same graph node enters every garment's loss: same values, same gradients (g + g is 2 g exactly).
        cur = getattr(self, '_curves_now', None)
        if cur is None:                                 # (called on its own: the reference's form)
            reg = self.inter_free_curve.regularization(fl_masks)
        else:
            hit = getattr(self, '_curve_reg', None)
            if hit is None or hit[0] is not cur:
                hit = self._curve_reg = (cur, self.inter_free_curve.regularization(fl_masks, cano_verts=cur))
            reg = hit[1]
        center = reg['center_offset'] * (conf.get_float('alpha_weight.center_weight') if 'alpha_weight' in conf else 1.)
        diff = reg['diff_a_loss'] * (conf.get_float('alpha_weight.diff_weight') if 'alpha_weight' in conf else 1.)
        self.info['fl_loss']['{}_project loss'.format(garment_name)] = fl_loss.detach()
        self.info['fl_loss']['{}_visible'.format(garment_name)] = body_visible.float().mean().detach()
        return fl_loss + center + diff

    def project_2d_loss(self, N, frame_ids, ratio, cameras):
        """OptimGarmentNetwork.py:1772-1883 (deform_feature_line :1507-1603, compute_fl_proj_loss :1605-1711): deform
        the explicit curves, keep the samples the body does not hide, chamfer them against the frame's 2-D feature
        lines, tie the canonical curves to their garment's zero level, one AdamW step on the curve parameters.  The
        gradients this leaves on the shared networks are cleared by the optimiser's zero_grad that follows."""
        conf = self.conf
        # fl_visible_method (:1573-1583): every shipped config says `zbuff`.  `surface` (fl_visible_by_surface_normal, :1312-1372)
        # cannot run in the reference either since the feature lines became curves: deform_feature_line hands it
        # `fl_meshes_dict[name] = None` (:1562) and it dereferences that.  Anything else raises there too (`raise NotImplemented`).
        method = conf.get_string('fl_visible_method') if 'fl_visible_method' in conf else 'zbuff'
        if method != 'zbuff':
            raise NotImplementedError("fl_visible_method = %r: only 'zbuff' runs (in the reference as well: its 'surface' branch "
                                      "reads curve MESHES that deform_feature_line no longer builds)" % method)
        early = getattr(self, '_early', None)
        if early is not None and early.get('frame_ids') is frame_ids and torch.device(self.device).type == 'cuda':
            # only the curve parameters are differentiated here (see below): the per-frame tensors enter as constants — the detached
            # gathers the ray pipeline prepared on the main stream (no second gather, no graph through codes / poses)
            torch.cuda.current_stream(self.device).wait_event(early['ready'])
            d_cond_list, poses, trans = [None] + list(early['d_cond']), early['poses'], early['trans']
        else:
            d_cond_list, poses, trans, _ = self.get_grad_parameters(frame_ids, self.device)
        smpl_conds = [poses, trans]
        self._shared_def_vs = self._deform_garments(N, frame_ids, ratio)
        curves_now = self._curves_now = self.inter_free_curve()                          # [L,S,3]
        self._curve_reg = None
        fl_vs_dict = {n: curves_now[i] for i, n in enumerate(self.fl_names)}
        gt_all, fl_mask_all = self._gt_feature_lines(frame_ids)                          # [N, L*M, 2], [N, L]
        M = gt_all.shape[1] // len(self.fl_names)
        gt_dict = {n: gt_all[:, i * M:(i + 1) * M] for i, n in enumerate(self.fl_names)}
        mask_dict = {n: fl_mask_all[:, i:i + 1] for i, n in enumerate(self.fl_names)}
        project_loss, sdf_loss = 0., 0.
        self.info['fl_loss'] = {}
        for g_i, name in enumerate(self.garment_names):
            names = self.fl_extract[name]
            d_cond = d_cond_list[g_i + 1]
            # the reference deforms the lines one by one (:1568); rows of the deformer are independent, so the garment's
            # lines go through it as one batch (same values, a third of the launches)
            split = [fl_vs_dict[n].view(-1, 3).shape[0] for n in names]
            def_all = self.deformer(torch.cat([fl_vs_dict[n].view(-1, 3) for n in names], dim=0).expand(N, -1, 3),
                                    [d_cond, smpl_conds], ratio=ratio, offset_type=names[0])
            def_fl_vs = list(torch.split(def_all, split, dim=1))
            cano_smpl = self.inter_free_curve.query_canosmpl_verts(names)
            checks = self.fl_visible_by_body_zbuff(cameras, d_cond, smpl_conds, ratio, def_fl_vs, cano_smpl, g_i,
                                                   name, N)                               # [N,P,2]
            fl_masks = torch.cat([mask_dict[n] for n in names], dim=-1)                   # [N, lines]
            garment_proj_loss = self.compute_fl_proj_loss(def_fl_vs, checks, fl_masks,
                                                          torch.cat([gt_dict[n] for n in names], dim=1), name, split,
                                                          cameras)
            project_loss = project_loss + garment_proj_loss
            # ---- canonical curves on their garment's zero level (:1855-1858)
            cano = torch.cat([fl_vs_dict[n].view(-1, 3) for n in names], dim=0)
            cano_sdf = self.garment_nets[g_i](cano, ratio, features=False).view(-1)
            s_loss = (cano_sdf + self.sdfShrinkRadius).abs().mean()
            self.info['fl_loss']['pc_{}_loss_sdf'.format(name)] = s_loss.detach()
            sdf_loss = sdf_loss + s_loss * (conf.get_float('fl_weight.sdf_weight') if 'fl_weight' in conf else 60.)
        self.fl_optimizer.zero_grad()
        if self.large_pose:
            loss = 0. * sdf_loss + 0. * project_loss           # OptimGarmentNetwork_Large_Pose.py:219: zero-weighted
        else:
            loss = 10. * sdf_loss + 1. * project_loss                                     # :1865
        # The reference calls loss.backward() here (:1866) and throws the gradients it leaves on the shared networks away
        # with the optimiser's zero_grad that follows (:1934): only the curve parameters' gradients are used.  They are
        # computed directly, so the branch touches no shared `.grad` and can run beside the mask loss on its own stream.
        curve_params = list(self.inter_free_curve.parameters())
        for q, gq in zip(curve_params, torch.autograd.grad(loss, curve_params, allow_unused=True)):
            q.grad = gq if gq is not None else torch.zeros_like(q)
        if getattr(self, '_allreduce', None) is not None:
            self._allreduce(list(self.inter_free_curve.parameters()))     # curve gradients are shared across ranks (§8e)
        self.fl_optimizer.step()
        self._curves_now = self._curve_reg = None          # (the curves have moved: this iteration's graph is spent)
        self.info['fl_loss']['total'] = loss.detach()

    # ------------------------------------------------------------------------------------------ mask loss
    def compute_garment_pc_loss(self, def_verts, defconds, imgs, gtMs, garment_type, garment_vs):
        """OptimGarmentNetwork.py:621-667: 1 - IoU of the splatted silhouette against the (dilated) ground-truth mask,
        plus the robust distance between the full deformation and skinning alone.  (The Laplacian / edge / normal
        terms have negative weights in every config of the reference — disabled, :633-650.)"""
        conf = self.conf
        N = gtMs.shape[0]
        masks = imgs[..., -1]                                                             # :624-629
        mask_loss = (1. - (masks * gtMs).view(N, -1).sum(1)
                     / (masks + gtMs - masks * gtMs).abs().view(N, -1).sum(1)).mean()
        self.info['pc_{}_mask_loss'.format(garment_type)] = mask_loss.detach()
        loss = mask_loss * (conf.get_float('pc_weight.mask_weight') if 'pc_weight.mask_weight' in conf else 1.)
        cw = conf.get_float('pc_weight.def_consistent.weight') if 'pc_weight.def_consistent' in conf else -1.
        if cw > 0.:                                                                       # :651-662
            offset2 = def_verts - self.deformer.defs[1](garment_vs.view(1, -1, 3).expand(N, -1, 3), defconds[1])
            offset2 = (offset2 * offset2).sum(-1)
            cc = conf.get_float('pc_weight.def_consistent.c')
            closs = utils.GMRobustError(offset2, cc, True).mean() if cc > 0. else torch.sqrt(offset2).mean()
            loss = loss + closs * cw
        return loss

    def mask_loss(self, N, frame_ids, ratio, cameras, defer_sdf_terms=False):
        """OptimGarmentNetwork.py:841-981 (`defer_sdf_terms`: stop after the SGD step; the caller adds pc_sdf_terms()): deform the explicit garment meshes, splat the merged point cloud into one
        alpha-composited silhouette per garment (pcRender, :937), IoU loss against the dilated ground-truth masks +
        LBS-consistency term (compute_garment_pc_loss, :621-667), SGD step on the explicit vertices, |SDF| loss."""
        conf = self.conf
        H, W = self.dataset.H, self.dataset.W
        def_vs = self._deform_garments(N, frame_ids, ratio)                                # :910
        d_cond_list, poses, trans = self._def_params
        whole = torch.cat(def_vs, dim=1) if len(def_vs) > 1 else def_vs[0]                 # :925-935
        pc_render = raster.PointsRendererWithFrags_Split(cameras, (H, W), radius=self.pc_radius, points_per_pixel=50)
        garment_masks_list, _frags = pc_render(whole, split_size=self.garment_vs[0].shape[0])   # :937
        rpx = int(np.round(self.pc_radius / 2. * float(min(H, W)) / 1.2))                  # :940-941
        garment_loss = 0.
        for g_i, name in enumerate(self.garment_names):
            gt = self._gt_garment_mask(g_i, frame_ids)
            if rpx > 0:                                                                   # :947
                gt = F.max_pool2d(gt, kernel_size=2 * rpx + 1, stride=1, padding=rpx)
            garment_loss = garment_loss + self.compute_garment_pc_loss(
                def_vs[g_i], [d_cond_list[g_i + 1], [poses, trans]], garment_masks_list[g_i], gt, name,
                self.garment_vs[g_i])
        # (the snapshot find_surface_ps reads — deformed meshes and PRE-step vertices — was taken in forward(), right after
        # the deformation: the ray pipeline runs on side streams while the backward below keeps the device busy)
        self.garment_optimizer.zero_grad()
        garment_loss.backward()                    # grads also reach deformer / codes / poses and stay for Adam (:959)
        if getattr(self, '_allreduce', None) is not None:
            self._allreduce(self.garment_vs)       # explicit MC vertices: identical numbering on every rank (§8e-2)
        self.garment_optimizer.step()
        if torch.device(self.device).type == 'cuda':
            self._sgd_done = torch.cuda.Event()      # the render loss samples around the POST-step vertices (:1108)
            self._sgd_done.record()
        if defer_sdf_terms:
            return [d.detach() for d in def_vs], None
        return [d.detach() for d in def_vs], self.pc_sdf_terms(ratio)

    def pc_sdf_terms(self, ratio, before_curve_term=None):
        """The |SDF| terms at the end of mask_loss (:963-972): the moved explicit vertices and the curve-aware disc pull
        their garment's zero level towards them.  Both parts are differentiated where they are formed (_backward_early) and
        return detached; `before_curve_term()` runs between them (the three-stream order waits for the curve branch THERE: the
        vertices' part does not read the curves, so the main stream does not sit idle until the curve stream has caught up)."""
        conf = self.conf
        pc_sdf_loss = 0.
        # The reference feeds the vertices WITH their autograd flag (:966), so its final backward also leaves d|f|/dv on
        # `garment_vs.grad` — which nothing ever reads: the only consumer of that field is the SGD step of the NEXT mask loss, and
        # `garment_optimizer.zero_grad()` (:959) clears it first (a re-mesh replaces the vertices altogether).  Detached here: same
        # parameters after every step, one input-gradient product per layer of the SDF nets on ~160 k vertices less (0.6 of the
        # iteration's 8 TFLOP; RECMV_PC_SDF_VERTEX_GRAD=1 computes the dead gradient again, for the A/B).
        dead = os.environ.get('RECMV_PC_SDF_VERTEX_GRAD') == '1'
        for g_i, name in enumerate(self.garment_names):                                   # :966-970
            verts = self.garment_vs[g_i] if dead else self.garment_vs[g_i].detach()
            mnfld_pred = self.garment_nets[g_i](verts, ratio, features=False).view(-1)
            sdf_loss = (mnfld_pred + self.sdfShrinkRadius).abs().mean()
            self.info['pc_{}_loss_sdf'.format(name)] = sdf_loss.detach()
            pc_sdf_loss = pc_sdf_loss + sdf_loss * conf.get_float('pc_weight.weight')
        pc_sdf_loss = self._backward_early(pc_sdf_loss)
        if before_curve_term is not None:
            before_curve_term()
        return pc_sdf_loss + self._backward_early(self.curve_aware_loss(ratio))            # :972

    CURVE_AWARE = CURVE_AWARE                                                              # utils/constant.py:228-232
    CURVE_AWARE_SAMPLES = 50000                                                            # :808, :833

    def curve_aware_loss(self, ratio, sampler=None):
        """OptimGarmentNetwork.py:787-839: the disc spanned by the `upper_bottom` curve (the waist opening of the upper
        garment) must lie ON the zero level of the LAST garment net (the bottom garment closes there): a triangle fan
        from the curve to its centroid, 50 000 area-weighted uniform samples on it, |SDF| of garment_nets[-1].  The
        samples are constants (the reference takes them through numpy), so the term only reaches the SDF parameters.
        Fires whenever 'upper_bottom' is a feature line of the garment set (female-3-casual: yes) and the weight is
        non-zero; datasets listed in CURVE_AWARE add the same term for their `bottom_curve` in the fine stage.

        The reference samples with trimesh 3.10.5 `Trimesh.sample` on the host from numpy's global RNG and uploads the
        points every iteration (:807-808); here `sample_fan_mesh` draws them on the device from torch's generator —
        same distribution, no host round trip.  `sampler(verts, faces, n)` overrides the draw (parity tests)."""
        conf = self.conf
        weight = conf.get_float('pc_weight.curve_aware_weight') if 'pc_weight.curve_aware_weight' in conf else 60.
        if weight == 0. or not getattr(self, 'curves', False):
            return 0.
        targets = []
        if 'upper_bottom' in self.fl_names:
            targets.append('upper_bottom')
        extra = self.CURVE_AWARE.get(getattr(self, 'garment_type', None))
        if extra is not None and getattr(self, 'isfine', False) and extra in self.fl_names:
            targets.append(extra)
        ca_loss = 0.
        for name in targets:
            curve_pts = self.inter_free_curve()[self.fl_names.index(name)].detach()         # [S,3]
            verts, faces = fan_mesh(curve_pts)
            pts = (sampler or sample_fan_mesh)(verts, faces, self.CURVE_AWARE_SAMPLES)
            pred = self.garment_nets[-1](pts, ratio, features=False).view(-1)
            circle = (pred + self.sdfShrinkRadius).abs().mean()
            self.info['pc_{}_circle_loss_sdf'.format(name)] = circle.detach()
            ca_loss = ca_loss + circle * weight
        return ca_loss

    # ------------------------------------------------------------------------------------------ rays
    def find_surface_ps(self, def_vs, tmp_vs, cameras):
        """OptimGarmentNetwork.py:742-767: per garment, rasterise the N deformed meshes and turn the first-hit
        fragments into (batch, row, col, canonical point, face) of every covered pixel."""
        out = []
        with torch.no_grad():
            for g_i, (def_v, gv, gf) in enumerate(zip(def_vs, tmp_vs, self.garment_fs)):
                out.append(utils.FindSurfacePs(gv, gf, self._garment_fragments(g_i, def_v, cameras)))
        return out

    def sample_train_ray(self, N, frame_ids, cameras):
        """find_surface_ps + sample_train_ray (:742-767, :983-1055).  Runs on a side stream: `nonzero` inside
        FindSurfacePs waits for the rasteriser only, not for the mask-loss backward queued on the main stream."""
        sample_pix = self.conf.get_int('sample_pix_num') if 'sample_pix_num' in self.conf else self.sample_pix
        sample_pix = sample_pix // self.garment_size
        def_vs, tmp_vs = self._surface_inputs
        cuda = torch.device(self.device).type == 'cuda'
        if cuda:
            main = torch.cuda.current_stream(self.device)
            if getattr(self, '_surface_stream', None) is None:
                self._surface_stream = _make_stream(self.device)
            side = self._surface_stream
            side.wait_event(self._surface_ready)
        ctx = torch.cuda.stream(side) if cuda else contextlib.nullcontext()
        out = []
        with ctx, torch.no_grad():
            found = self.find_surface_ps(def_vs, tmp_vs, cameras)
            for g_i, (batch_inds, row_inds, col_inds, init_pts, _faces) in enumerate(found):
                gt = self._gt_garment_mask(g_i, frame_ids)                              # :1013-1018
                keep = (gt[batch_inds, row_inds, col_inds] > 0.).nonzero(as_tuple=True)[0]
                batch_inds, row_inds, col_inds, init_pts = (t[keep] for t in (batch_inds, row_inds, col_inds,
                                                                              init_pts))
                pnum = batch_inds.shape[0]
                if pnum > sample_pix * N:                                               # :1019-1027, host RNG
                    # same draw as the reference (torch's host generator); the compare + nonzero run in numpy: torch
                    # would fork its whole intra-op thread pool for ~1e5 elements (tens of ms on a 256-core host)
                    sel = torch.rand(pnum).numpy() < float(sample_pix * N) / float(pnum)
                    idx = torch.from_numpy(np.flatnonzero(sel)).to(batch_inds.device, non_blocking=False)
                    batch_inds, row_inds, col_inds, init_pts = (t[idx] for t in (batch_inds, row_inds, col_inds,
                                                                                 init_pts))
                rays = cameras.view_rays_pix(col_inds, row_inds)
                out.append((batch_inds, row_inds, col_inds, init_pts.contiguous(), rays))
        if cuda:
            main.wait_stream(side)
            for sample in out:
                for t in sample:
                    t.record_stream(main)
        self._surface_inputs = None
        self.info['surface_pixels'] = [int(f[0].shape[0]) for f in found]
        return out

    def _prepare_rays_early(self, frame_ids, cameras, ratio):
        """What the ray pipeline (find_surface_ps -> sample_train_ray -> root finder) needs from the main stream, produced
        NOW, before the curve branch and the mask loss are queued: the per-frame tensors of the batch, the camera centre,
        the root finder's shared state (weight-normed weights + transposes, posed skeleton, chain descriptors).  With
        these and the deformed vertices the pipeline runs on side streams underneath the mask loss's large GEMMs instead
        of behind them; the root finder's result is the same (it reads the nets, it does not change them)."""
        if torch.device(self.device).type != 'cuda' or os.environ.get('RECMV_SERIAL') == '1':
            self._early = None              # RECMV_SERIAL=1: the reference's order on one stream (A/B timing, determinism)
            return
        with torch.no_grad():
            d_cond_list, poses, trans, _ = self.get_grad_parameters(frame_ids, self.device)
            early = dict(d_cond=[c.detach() for c in d_cond_list[1:]], poses=poses.detach(), trans=trans.detach(),
                         cam_pos=cameras.cam_pos().detach().clone())
            utils.prepare_root_finder(list(self.garment_nets), self.deformer, [early['poses'], early['trans']], ratio)
        early['frame_ids'] = frame_ids
        early['ready'] = torch.cuda.Event()
        early['ready'].record()
        self._early = early

    def opt_garment_surface_ps(self, frame_ids, cameras, ratio, samples):
        early = getattr(self, '_early', None)
        if early is not None:
            defconds_list = [early['d_cond'], [early['poses'], early['trans']]]
            cam_pos, after = early['cam_pos'], [early['ready'], self._surface_stream]
        else:
            d_cond_list, poses, trans, _ = self.get_grad_parameters(frame_ids, self.device)
            defconds_list = [d_cond_list[1:], [poses, trans]]
            cam_pos, after = cameras.cam_pos().detach(), None
        pts, checks = utils.OptimizeGarmentSurfacePs(
            cam_pos, [s[4].detach() for s in samples], [s[3] for s in samples],
            [s[0] for s in samples], self.garment_nets, ratio, self.deformer, defconds_list,
            garment_names=self.garment_names, dthreshold=5.e-5, athreshold=self.angThred, w1=3.05, w2=1., times=20,
            after=after)
        self.info['rays_total'] = sum(c.numel() for c in checks)
        self._ray_valid = [c.sum() for c in checks]
        self._ray_valid_host = None
        if checks and checks[0].is_cuda:
            # the converged-ray counts travel to pinned memory behind the root finder; the render loss waits for THIS copy (an event),
            # once for all garments, instead of one blocking read per garment
            host = torch.empty(len(checks), dtype=torch.int64).pin_memory()
            host.copy_(torch.stack(self._ray_valid), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._ray_valid_host = (host, ev)
        return pts, checks

    # ------------------------------------------------------------------------------------------ render loss
    def surface_render_loss(self, N, cameras, frame_ids, ratio, checks, init_ps_list, samples):
        conf = self.conf
        gtCs, gtNs = self._gt_images(frame_ids)
        self.TmpPs = [None] * self.garment_size
        self.rays = [None] * self.garment_size
        self.batch_inds = [None] * self.garment_size
        self.row_inds = [None] * self.garment_size
        self.col_inds = [None] * self.garment_size
        surface_sample_points = 4096 // self.garment_size
        d_cond_list, poses, trans, rendcond = self.get_grad_parameters(frame_ids, self.device)
        dev = self.device
        # The garments' terms are independent chains of small launches (jets on a few thousand points, the colour net, a tail of
        # element-wise work) — forward AND backward, since autograd runs every node on the stream of its forward — and the backward
        # of this loss is the tail of the iteration, one stream of dependent launches after both chains have joined.  The second
        # garment's chain goes to the root finder's side stream (no new hardware queue), forked behind everything queued so far and
        # joined before the sum (the host draws its random numbers in the same order either way, every garment's terms are summed
        # on their own first, and the shared parameters' gradients are accumulated in the engine's order: 40 iterations bit-identical
        # to the one-stream order and from run to run, tools/determinism_probe.py).  Round 3 measured this slower (the phase was
        # paced by host read-backs then); with those gone and one jet pass per net it is worth 3-4 % of the iteration
        # (tools/ab_interleaved.py render_streams, profiles/r04_ab_render_streams.txt).  RECMV_RENDER_STREAMS=0: one stream (A/B).
        side = None
        # (Round 4 took the side stream out of the bf16x6 mode's default on the strength of 12-run samples; round 5's in-process counts
        # — tools/erratum/loop_repro_inproc.py, 40-100 repetitions per cell — show that mode parting with AND without it, and why: a kernel of
        # the iteration computes wrong values beside that mode's product kernels, whatever the schedule, DESIGN.md §9.  The f32 mode is
        # identical in every repetition of every cell, so the side stream is simply the default.)
        rs = os.environ.get('RECMV_RENDER_STREAMS', '1')
        if (torch.device(dev).type == 'cuda' and self.garment_size > 1 and os.environ.get('RECMV_SERIAL') != '1'
                and rs != '0'):
            from .utils.FindSurfacePs import _streams
            base = torch.cuda.current_stream(dev)
            side = _streams(torch.device(dev), self.garment_size - 1)
            fork = torch.cuda.Event()
            fork.record()
        losses = []
        for g_i, (init_ps, sample) in enumerate(zip(init_ps_list, samples)):
            ctx = contextlib.nullcontext()
            if side is not None and g_i > 0:
                side[g_i - 1].wait_event(fork)
                ctx = torch.cuda.stream(side[g_i - 1])
            with ctx:
                losses.append(self._garment_render_terms(g_i, init_ps, sample, N, cameras, ratio, checks[g_i], gtCs, gtNs,
                                                         surface_sample_points, d_cond_list, poses, trans, rendcond))
        if side is not None:
            for st in side:
                base.wait_stream(st)
            # what the side chains leave for the base stream (the sum below, the backward pass, propagateTmpPsGrad): their blocks
            # belong to the side stream's pool, so the allocator has to know about the second reader
            self._record_handover(base, losses[1:])
        total_loss = 0.
        for loss_g in losses:
            total_loss = total_loss + loss_g
        return total_loss

    def _garment_render_terms(self, g_i, init_ps, sample, N, cameras, ratio, check, gtCs, gtNs, surface_sample_points,
                              d_cond_list, poses, trans, rendcond):
        """One garment's share of surface_render_loss (:1100-1217): eikonal term, deformation regulariser, and on the converged
        rays the colour and the weighted normal term; leaves what propagateTmpPsGrad reads (TmpPs, rays, pixel indices)."""
        conf, dev = self.conf, self.device
        total_loss = 0.
        batch_inds, row_inds, col_inds, _, rays = sample
        name = self.garment_names[g_i]
        net = self.garment_nets[g_i]
        TmpVs = self.garment_vs[g_i]
        nonmnfld = utils.sample_points(torch.cat([init_ps, _host_subset(TmpVs, surface_sample_points)], dim=0), 1.8, 0.01)
        nonmnfld.requires_grad_()
        # The eikonal points and (below) the converged rays go through THIS net with the same parameters: one jet pass over both
        # row blocks instead of two (rows are independent; the reference evaluates the net twice, :1113 and :1166) — the ray
        # block's nine layers of small launches, forward and backward, ride along in the launches of the eikonal block.  The pass
        # is issued once the converged rays are known; RECMV_MERGE_JETS=0 keeps the two passes (A/B).
        merge = os.environ.get('RECMV_MERGE_JETS', '1') != '0'
        grad_loss = None
        if not merge:
            pred = net(nonmnfld, ratio, jet=True, features=False)
            grad = net.gradient(nonmnfld, pred)
            grad_loss = ((grad.norm(2, dim=-1) - 1) ** 2).mean()                       # eikonal :1118
        d_cond = d_cond_list[g_i + 1]
        def_term = None
        def_pts = None
        if 'def_regu' in conf and conf.get_float('def_regu.weight') > 0.:               # :1135-1155
            pts = torch.cat([init_ps, _host_subset(TmpVs, surface_sample_points)], dim=0)
            pts = torch.cat([pts, utils.sample_points(pts, 1.8, 0.01, 0)], dim=0).view(1, -1, 3).expand(N, -1, 3)
            def_pts = pts.contiguous().requires_grad_()
            if not (merge and torch.device(dev).type == 'cuda'):
                def_term = HotLoop._def_regu_term(self, name, def_pts, self.deformer.defs[0](def_pts, d_cond, ratio=ratio, offset_type=name,
                                                                                    jet=True))
        # the reference gates on rayInfo[1] > 0 via .item(); the gate is kept but read once per garment
        host = getattr(self, '_ray_valid_host', None)
        if host is not None:
            host[1].synchronize()
            n_valid = int(host[0][g_i])
        else:
            n_valid = int(self._ray_valid[g_i])
        self.info.setdefault('rays_converged', []).append(n_valid)
        sdfs = nx = rend_feat = None
        if n_valid > 0:
            # rows of the converged rays: their count is on the host already, so ONE index list of known size serves the five
            # gathers (`x[check]` would read its own count back five times: a host round trip each, in the middle of a phase the
            # device is waiting on — tools/phase_overlap.py)
            idx = _nonzero_known(check, n_valid)
            self.TmpPs[g_i] = init_ps.index_select(0, idx)
            self.TmpPs[g_i].requires_grad = True
            self.rays[g_i] = rays.index_select(0, idx)
            self.batch_inds[g_i] = batch_inds.index_select(0, idx)
            self.col_inds[g_i] = col_inds.index_select(0, idx)
            self.row_inds[g_i] = row_inds.index_select(0, idx)
        if def_pts is not None and def_term is None:
            # the regulariser's points and the converged rays through the offset MLP as ONE jet pass too (same net, same code table;
            # the reference calls it at :1143 and again inside compute_cardinal_rays :1176): the rays' block is parked on the net and
            # served to the composite deformer's call below
            tr = self.deformer.defs[0]
            jet_ok = (getattr(tr, 'embed_fn', None) is not None and def_pts.is_cuda and def_pts.dtype == torch.float32
                      and torch.is_grad_enabled())             # MLPTranslator.forward's own gate for its jet path
            if n_valid > 0 and jet_ok and hasattr(tr, 'jet_two_blocks'):
                defVs_r = tr.jet_two_blocks(def_pts, d_cond, self.TmpPs[g_i], self.batch_inds[g_i], ratio['deformerRatio'], name)
            else:
                defVs_r = tr(def_pts, d_cond, ratio=ratio, offset_type=name, jet=True)
            def_term = HotLoop._def_regu_term(self, name, def_pts, defVs_r)
        if grad_loss is None:
            if n_valid > 0:
                n0 = nonmnfld.shape[0]
                both = torch.cat([nonmnfld, self.TmpPs[g_i]], dim=0)
                pred_all = net(both, ratio, jet=True)
                feat_all = net.rendcond
                grad_all = net.gradient(both, pred_all)
                grad, sdfs, nx, rend_feat = grad_all[:n0], pred_all[n0:], grad_all[n0:], feat_all[n0:]
            else:
                pred = net(nonmnfld, ratio, jet=True, features=False)
                grad = net.gradient(nonmnfld, pred)
            grad_loss = ((grad.norm(2, dim=-1) - 1) ** 2).mean()                       # eikonal :1118
        self.info['{}_grad_loss'.format(name)] = grad_loss.detach()
        total_loss = total_loss + grad_loss * conf.get_float('grad_weight')
        if def_term is not None:
            total_loss = total_loss + def_term
        if n_valid > 0:
            p, b = self.TmpPs[g_i], self.batch_inds[g_i]
            if sdfs is None:
                sdfs = net(p, ratio, jet=True)
                rend_feat = net.rendcond
                nx = net.gradient(p, sdfs)       # = autograd.grad(sdfs, p, ones, create_graph=True) (:1169-1172)
            onx = nx.detach()
            nx = nx / nx.norm(dim=1, keepdim=True)
            defconds = [d_cond, [poses, trans]]
            crays, defVs = utils.compute_cardinal_rays(self.deformer, p, self.rays[g_i], defconds, b, ratio,
                                                       'train', offset_type=name)
            grad_d_p = None                      # d(defVs)/dp, formed once for the two terms below that use it
            if conf.get_float('color_weight') > 0.:
                colors = utils.compute_netRender_color(self.netRender, p, defVs, nx, crays, rend_feat,
                                                       rendcond[b], ratio)
                color_loss = (gtCs[b, self.row_inds[g_i], self.col_inds[g_i], :] - colors).abs().sum(1)
                color_loss = utils.scatter_mean(color_loss, b, N).mean()
                self.info['{}_color_loss'.format(name)] = color_loss.detach()
                total_loss = total_loss + conf.get_float('color_weight') * color_loss
            if 'normal_weight' in conf and conf.get_float('normal_weight') > 0. and gtNs is not None:  # :1191-1217
                if 'weighted_normal' in conf and conf.get_bool('weighted_normal'):
                    # compute_deformed_normals(..., 'test') of the reference (:1196) evaluates the SDF gradient and
                    # the deformer Jacobian at p once more without a graph; both are at hand already (the SDF jet
                    # above, the deformer jet of compute_cardinal_rays): J^-T grad f, J grad f where J is singular
                    if grad_d_p is None:
                        grad_d_p = utils.compute_Jacobian(p, defVs, True, True)
                    with torch.no_grad():
                        Jd = grad_d_p.detach()
                        Jd_inv, inv_ok = utils.FastDiff3x3MinvFunction.apply(Jd)
                        cnx = torch.where(inv_ok.view(-1, 1), (Jd_inv.transpose(-2, -1) * onx.unsqueeze(-2)).sum(-1),
                                          (Jd * onx.unsqueeze(-2)).sum(-1))
                        cnx = cnx / cnx.norm(dim=1, keepdim=True)
                    weights = torch.clamp((-self.rays[g_i] * cnx).sum(1).detach(), max=1., min=0.) ** 2
                else:
                    weights = torch.ones(nx.shape[0], device=dev)
                gtn = gtNs[b, self.row_inds[g_i], self.col_inds[g_i], :]
                flip = torch.diag(torch.ones(3, device=dev) * torch.arange(-1., 2., device=dev).abs().mul(-2.).add(1.))
                # = diag(-1, 1, -1), formed on the device (a host list would be a blocking H2D copy on this stream)
                M = (cameras.R[0].unsqueeze(-1) * flip.unsqueeze(0)).sum(1)             # R @ flip
                gtn = (M.unsqueeze(0) * gtn.unsqueeze(-2)).sum(-1)
                gtnorms = gtn.norm(dim=1, keepdim=True)
                valid_mask = (gtnorms > 0.0001)[..., 0]
                gtn = torch.where(valid_mask.unsqueeze(-1), gtn / gtnorms.clamp(min=1e-12), gtn)
                # d(deformed)/dp: the Jacobian carried by compute_cardinal_rays' jet pass over the same points with
                # the same parameters (the reference evaluates the deformer once more, :1207-1208)
                if grad_d_p is None:
                    grad_d_p = utils.compute_Jacobian(p, defVs, True, True)
                gtn = (grad_d_p.transpose(-2, -1) * gtn.unsqueeze(-2)).sum(-1)
                normal_loss = (gtn - nx).norm(2, dim=1) * weights
                w = valid_mask.to(normal_loss.dtype)
                from .ops import rows_sum_by_index                   # per-frame sums in a fixed order
                num = rows_sum_by_index((normal_loss * w).view(-1, 1), b, N).view(-1)
                den = rows_sum_by_index(w.view(-1, 1), b, N).view(-1)
                normal_loss = (num / den.clamp(min=1)).mean()
                self.info['{}_normal_loss'.format(name)] = normal_loss.detach()
                total_loss = total_loss + conf.get_float('normal_weight') * normal_loss
            # for propagateTmpPsGrad: the two jets at these points (same parameters until the optimiser steps)
            self.__dict__.setdefault('_prop_pre', {})[g_i] = (
                p, onx, grad_d_p.detach() if grad_d_p is not None else None, _param_versions(net, self.deformer))
        self.deformer.defs[0].__dict__.pop('_jet_prefetch', None)      # (a parked ray block nobody asked for dies with the phase)
        return total_loss

    def _record_handover(self, stream, extra=()):
        """record_stream(`stream`) on every tensor the render loss leaves on this object for a later phase on another stream."""
        keep = list(extra)
        for name in ('TmpPs', 'rays', 'batch_inds', 'row_inds', 'col_inds'):
            keep += [t for t in getattr(self, name, None) or [] if t is not None]
        for pre in getattr(self, '_prop_pre', {}).values():
            keep += [t for t in pre[:3] if torch.is_tensor(t)]
        for t in keep:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)

    def _def_regu_term(self, name, pts, defVs):
        """The deformation regulariser (:1135-1155) from the offset MLP's output at `pts` (its Jacobian carried by the jet pass)."""
        conf = self.conf
        Jacobs = utils.compute_Jacobian(pts, defVs, True, True)
        if os.environ.get('RECMV_REGU_HOST_SVD') == '1':
            # the reference's own route, round trip and all (:1148-1150: LAPACK on the host, autograd through the SVD) — an
            # experiment switch of tools/trajectory_seeds.py (which rounding seeds the trajectories' divergence?), never the loop's path
            _, s, _ = torch.svd(Jacobs.cpu())
            s = torch.log(s.to(Jacobs.device))
            def_loss = utils.GMRobustError((s * s).sum(1), conf.get_float('def_regu.c'), True).mean()
        elif Jacobs.is_cuda and Jacobs.dtype == torch.float32 and os.environ.get('RECMV_FUSED_REGU', '1') != '0':
            # value and dy/dJ per matrix from one launch (csrc/def_regu.hip) instead of ~90 torch launches forward + backward
            from .ops import def_regu
            def_loss = def_regu(Jacobs, conf.get_float('def_regu.c')).mean()
        else:
            s = torch.log(singular_values_3x3(Jacobs))
            def_loss = utils.GMRobustError((s * s).sum(1), conf.get_float('def_regu.c'), True).mean()
        self.info['def_{}_loss'.format(name)] = def_loss.detach()
        return def_loss * conf.get_float('def_regu.weight')

    def dct_poses_loss(self, poses, trans, frame_ids, N):
        if not (poses.requires_grad or trans.requires_grad) or self.conf.get_float('dct_weight') <= 0.:
            return 0.
        klen, Nlen = self.dctnull.shape
        bp, _ = self.dataset.get_batchframe_data('poses', frame_ids, Nlen)
        bt, _ = self.dataset.get_batchframe_data('trans', frame_ids, Nlen)
        posedJs = self.deformer.defs[1].posedSkeleton([bp.reshape(N * Nlen, 24, 3), bt.reshape(N * Nlen, 3)])
        x = posedJs.reshape(N, Nlen, 72)
        dct = (self.dctnull[None, :, :, None] * x[:, None, :, :]).sum(2)                    # dctnull @ x
        dct_loss = dct.abs().mean()
        self.info['dct_loss'] = dct_loss.detach()
        return dct_loss * self.conf.get_float('dct_weight')

    @staticmethod
    def _backward_early(loss):
        """The |SDF| terms of the mask loss (:963-972) depend on nothing the ray pipeline produces: their backward runs as soon as
        they exist (like the mask loss's own, :959) instead of inside the caller's `loss.backward()`, and the value travels on in
        the returned loss detached.  The gradients are the same sums, accumulated as (mask loss) + (|SDF| terms) + (render loss +
        pose prior) — the order is the same in the serial and in the three-stream schedule."""
        if torch.is_tensor(loss) and loss.requires_grad and os.environ.get('RECMV_EARLY_BWD', '1') != '0':
            loss.backward()
            return loss.detach()
        return loss

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, frame_ids, ratio, global_optimizer=None):
        """OptimGarmentNetwork.forward (:1885-1969) on the frames `frame_ids`; `global_optimizer` is the caller's Adam
        whose gradients are cleared after the curve branch (:1934; train.py passes it as a keyword, :324)."""
        N = frame_ids.numel()
        self.info = {}
        self._tail = None
        self._def_cache, self._frag_cache, self._def_params = None, {}, None      # per-iteration caches (shared deformation / fragments)
        cameras = self._cameras()
        # a second camera object for the ray phases (its own autograd graph: the mask loss's backward frees the first
        # one's, :1036), built NOW so that the side streams of the ray pipeline never wait for the main stream's queue
        cameras_rays = self._cameras()
        if self.body_vs is None or self.forward_time % self.remesh_intersect == 0:
            with self._phase('remesh'):
                self.marching_cube_update(ratio)
        total_loss = 0.
        with self._phase('deform'):
            def_vs = self._deform_garments(N, frame_ids, ratio)
            # find_surface_ps reads the deformed meshes and the PRE-step vertices (:918 runs before the SGD step)
            self._surface_inputs = ([d.detach() for d in def_vs], [v.detach().clone() for v in self.garment_vs])
            self._surface_ready = torch.cuda.Event() if torch.device(self.device).type == 'cuda' else None
            if self._surface_ready is not None:
                self._surface_ready.record()
            self._prepare_rays_early(frame_ids, cameras_rays, ratio)
        opt = global_optimizer if global_optimizer is not None else self.optimizer
        cuda = torch.device(self.device).type == 'cuda'
        # The dependency-graph order below is the default on the device — for frame-sharded ranks too: every collective is issued
        # from the stream whose branch needs it (vertices: main, curves: curve stream, shared gradients: main after the backward),
        # in the same host order on every rank.  RECMV_OVERLAP_ORDER=1 runs that ORDER on the host as well (no streams there:
        # tests/test_loop_cpu.py compares it with the serial order for world size 2).
        overlap = os.environ.get('RECMV_SERIAL') != '1' and (cuda or os.environ.get('RECMV_OVERLAP_ORDER') == '1')
        if not overlap:
            # ---- the reference's order, one phase after the other
            if self.curves:
                with self._phase('curves'):
                    self.project_2d_loss(N, frame_ids, ratio, cameras)                       # :1932
            opt.zero_grad()                                                                # :1934
            with self._phase('mask_loss'):
                def_vs, pc_sdf_loss = self.mask_loss(N, frame_ids, ratio, cameras)
            total_loss = total_loss + pc_sdf_loss                  # (detached: differentiated inside, _backward_early)
            d_cond_list, poses, trans, rendcond = self.get_grad_parameters(frame_ids, self.device)
            cameras = cameras_rays                                                         # rebuilt graph (:1036)
            with self._phase('sample_rays'):
                samples = self.sample_train_ray(N, frame_ids, cameras)
            with self._phase('root_find'):
                init_ps_list, checks = self.opt_garment_surface_ps(frame_ids, cameras, ratio, samples)
            with self._phase('render_loss_fwd'):
                total_loss = total_loss + self.surface_render_loss(N, cameras, frame_ids, ratio, checks, init_ps_list,
                                                                   samples)
            with self._phase('dct'):
                total_loss = total_loss + self.dct_poses_loss(poses, trans, frame_ids, N)
        else:
            # ---- the same terms as a dependency graph over three streams.  What depends on what:
            #   mask loss (explicit vertices: splat, IoU, backward through the deformer, SGD step)  <- deformation
            #   ray pipeline (surface points, ray sampling, root finder)                             <- deformation
            #   curve branch (its gradients reach the curve parameters only, see project_2d_loss)   <- deformation
            #   |SDF| terms: vertices after the SGD step, curve-aware disc after the curve step
            #   render loss: root finder + vertices after the SGD step
            # The mask loss's large GEMMs go to the main stream first; the two chains of small launches run beside them
            # on side streams; the final backward runs every node on the stream of its forward.  Same arithmetic, same
            # random draws in the same host order: bit-identical to the serial order (tools/determinism_probe.py).
            if cuda:
                main = torch.cuda.current_stream(self.device)
                if getattr(self, '_surface_stream', None) is None:
                    self._surface_stream = _make_stream(self.device)
                if getattr(self, '_curve_stream', None) is None:
                    self._curve_stream = _make_stream(self.device)
                s_ray, s_curve = self._surface_stream, self._curve_stream
                on = torch.cuda.stream
            else:                                            # the same order on the host: one queue, nothing to wait for
                main = s_ray = s_curve = _NoStream()
                on = lambda s_: contextlib.nullcontext()
            opt.zero_grad()              # (:1934) nothing of the curve branch lands on the shared gradients any more
            with self._phase('mask_loss'):
                self.mask_loss(N, frame_ids, ratio, cameras, defer_sdf_terms=True)
            with on(s_ray):
                with self._phase('sample_rays'):
                    samples = self.sample_train_ray(N, frame_ids, cameras_rays)            # waits for the deformation only
            with on(s_ray), self._phase('root_find'):
                init_ps_list, checks = self.opt_garment_surface_ps(frame_ids, cameras_rays, ratio, samples)
            curve_done = None
            if self.curves:
                with on(s_curve), self._phase('curves'):
                    s_curve.wait_event(self._surface_ready)
                    self.project_2d_loss(N, frame_ids, ratio, cameras)                       # :1932
                    if cuda:
                        curve_done = torch.cuda.Event()
                        curve_done.record()
            with self._phase('pc_sdf'):
                # differentiated NOW (inside, _backward_early): the host would otherwise sit in the render loss's first read-back
                # (the converged-ray count) until the root finder has finished, with this backward — the SDF nets on ~170 k
                # vertices — still unqueued.  curve_aware_loss reads the curves after their AdamW step: the wait sits in front of it
                total_loss = total_loss + self.pc_sdf_terms(
                    ratio, before_curve_term=(lambda: main.wait_event(curve_done)) if curve_done is not None else None)
            # The terms that are still to be differentiated — render loss + pose prior — are summed ON THE RAY STREAM and kept
            # (`_tail`): step() starts their backward from that stream as soon as the render loss exists.  Summed into the main
            # stream's total first (rounds 2-5), the backward's root sat in the main stream's queue behind the |SDF| terms' large
            # products and the iteration's tail (render backward + implicit differentiation, ~20 ms of small launches) started only
            # when those had drained.  Same nodes, same accumulation order; RECMV_TAIL_STREAM=0: the old form (A/B).
            tail_on_ray = cuda and os.environ.get('RECMV_TAIL_STREAM', '1') != '0'
            with on(s_ray), self._phase('render_loss_fwd'):
                s_ray.wait_event(getattr(self, '_sgd_done', None))
                render_loss = self.surface_render_loss(N, cameras_rays, frame_ids, ratio, checks, init_ps_list, samples)
                if tail_on_ray:
                    _, poses, trans = self._def_params if self._def_params is not None else self.get_grad_parameters(frame_ids, self.device)[:3]
                    tail = render_loss + self.dct_poses_loss(poses, trans, frame_ids, N)
                    self._tail = (tail, s_ray) if torch.is_tensor(tail) and tail.requires_grad else None
            main.wait_stream(s_ray)
            if tail_on_ray:
                self._record_handover(main, [tail] if torch.is_tensor(tail) else [])
                total_loss = total_loss + tail        # (a caller's plain `loss.backward()` still works — from the main stream's queue)
            else:
                if cuda:
                    self._record_handover(main, [render_loss] if torch.is_tensor(render_loss) else [])
                total_loss = total_loss + render_loss
                with self._phase('dct'):
                    # (the pose prior gathers its own 30-frame windows; of the batch's poses / translations it reads `requires_grad` only)
                    _, poses, trans = self._def_params if self._def_params is not None else self.get_grad_parameters(frame_ids, self.device)[:3]
                    total_loss = total_loss + self.dct_poses_loss(poses, trans, frame_ids, N)
        self.forward_time += 1
        return total_loss

    # ------------------------------------------------------------------------------------------ implicit diff
    def propagateTmpPsGrad(self, frame_ids, ratio):
        """Implicit differentiation of the surface point p(theta, phi, z, cam) — OptimGarmentNetwork.py:2159-2313.
        The reference `return`s (not `continue`s) at the first garment without valid rays (:2165); kept."""
        if (self.garment_size == 2 and os.environ.get('RECMV_PROP_JOINT', '1') != '0'
                and all(self.TmpPs[g] is not None and self.TmpPs[g].grad is not None and self.TmpPs[g].is_cuda for g in range(2))
                and HotLoop._propagate_joint(self, frame_ids, ratio)):
            return
        for g_i in range(self.garment_size):
            name = self.garment_names[g_i]
            if self.TmpPs[g_i] is None or self.TmpPs[g_i].grad is None:
                self.info['{}_invInfo'.format(name)] = (-1, -1)
                return
            dev = self.TmpPs[g_i].device
            d_cond_list, poses, trans, _ = self.get_grad_parameters(frame_ids, dev)
            defconds = [d_cond_list[1 + g_i], [poses, trans]]
            cameras = self._cameras()
            grad_l_p = self.TmpPs[g_i].grad
            col, row = self.col_inds[g_i], self.row_inds[g_i]
            v = cameras.view_rays_pix(col, row)
            c = cameras.cam_pos()
            p = self.TmpPs[g_i]
            net = self.garment_nets[g_i]
            opt_defconds = [t for t in (defconds[0], defconds[1][0], defconds[1][1]) if t.requires_grad]
            # grad f(p) and d(deformed)/dp at the surface points (:2176-2190): the render loss evaluated both jets at these
            # very points with these very parameters a moment ago (surface_render_loss keeps them); recomputed otherwise
            pre = getattr(self, '_prop_pre', {}).get(g_i)
            if pre is not None and pre[0] is p and pre[3] == _param_versions(net, self.deformer) and pre[2] is not None:
                grad_f_p, grad_d_p = pre[1], pre[2]
            else:
                f = net(p, ratio, jet=True)
                grad_f_p = net.gradient(p, f).detach()
                d = self.deformer(p, defconds, self.batch_inds[g_i], ratio=ratio, offset_type=name, jet=True)
                grad_d_p = utils.compute_Jacobian(p, d, False, False)
            vd = v.detach()
            zeros = torch.zeros_like(vd[:, 0])
            v_cross = torch.stack([torch.stack([zeros, -vd[:, 2], vd[:, 1]], -1),
                                   torch.stack([vd[:, 2], zeros, -vd[:, 0]], -1),
                                   torch.stack([-vd[:, 1], vd[:, 0], zeros], -1)], dim=1)   # [v]_x
            a1 = (v_cross.unsqueeze(-1) * grad_d_p.unsqueeze(-3)).sum(-2)                   # v_cross @ J
            b = torch.cat([grad_f_p.view(-1, 1, 3), a1], dim=1)                             # [P,4,3]
            btb = (b.unsqueeze(-1) * b.unsqueeze(-2)).sum(1)                                # b^T b
            btb_inv, check = Fast3x3Minv(btb.contiguous())
            self.info['{}_invInfo'.format(name)] = (check.numel(), check.sum())
            rhs_1 = (btb_inv.unsqueeze(-1) * b.permute(0, 2, 1).unsqueeze(-3)).sum(-2)       # [P,3,4]
            rhs_1 = (grad_l_p.view(-1, 3, 1) * rhs_1).sum(1, keepdim=True)                  # [P,1,4]
            # The reference injects these gradients with `loss += (param * grad).sum(); loss.backward()` (:2269-2313).
            # d/dparam of that sum IS `grad`, so the leaves are accumulated directly (one multi-tensor add) and the
            # non-leaf targets (per-frame codes gathered by frame id, rays, camera centre) get one backward call.
            targets, grads = [], []
            params = [q for q in net.parameters() if q.requires_grad]
            if params:                     # frozen in the large-pose stage (OptimGarmentNetwork_Large_Pose.py:440-452)
                pg = torch.autograd.grad(net(p, ratio), params, -rhs_1[:, :, 0])
                targets += params
                grads += list(pg)
            params = [q for q in self.deformer.parameters() if q.requires_grad]
            d = self.deformer(p, defconds, self.batch_inds[g_i], ratio=ratio, offset_type=name)
            temp = -(rhs_1[:, :, -3:].transpose(1, 2) * v_cross).sum(1)                      # rhs[1:4] @ (-[v]_x)
            # one reverse sweep for the deformer's parameters and the per-frame tensors (the reference's two `backward`
            # calls, :2286-2301, walk the same graph twice)
            pg = torch.autograd.grad(d, params + opt_defconds, temp)
            targets += params + opt_defconds
            grads += list(pg)
            if v.requires_grad:
                dc = d.detach() - c.detach().view(1, 3)
                dc_cross = torch.stack([torch.stack([zeros, -dc[:, 2], dc[:, 1]], -1),
                                        torch.stack([dc[:, 2], zeros, -dc[:, 0]], -1),
                                        torch.stack([-dc[:, 1], dc[:, 0], zeros], -1)], dim=1)
                targets.append(v)
                grads.append((rhs_1[:, :, -3:].transpose(1, 2) * dc_cross).sum(1))
            if c.requires_grad:
                targets.append(c)
                grads.append((-temp.sum(0)).view_as(c))
            _inject_gradients(targets, grads)

    def _propagate_joint(self, frame_ids, ratio):
        """propagateTmpPsGrad for BOTH garments as one block of rows wherever they share the arithmetic: the 3 x 3 algebra of the
        implicit differentiation, the camera rays, and the pass through the deformer (one offset MLP and one skinner for both; a
        row's code comes from the garments' code tables stacked) with ONE reverse sweep for the deformer's parameters and the
        per-frame tensors.  The garment nets stay one pass each.  Same terms as the per-garment loop below (the sums over the two
        garments' rows of the shared tensors' gradients are formed in one product instead of two and an add); half the launches of
        a phase that sits in the host-paced tail of the iteration.  Returns False (nothing done) unless both garments' jets are at
        hand from the render loss."""
        G = 2
        pre = getattr(self, '_prop_pre', {})
        ps = [self.TmpPs[g] for g in range(G)]
        for g in range(G):
            h = pre.get(g)
            if (h is None or h[0] is not ps[g] or h[2] is None
                    or h[3] != _param_versions(self.garment_nets[g], self.deformer)):
                return False
        dev = ps[0].device
        d_cond_list, poses, trans, _ = self.get_grad_parameters(frame_ids, dev)
        cameras = self._cameras()
        n = [int(p.shape[0]) for p in ps]
        grad_l_p = torch.cat([p.grad for p in ps], dim=0)
        col = torch.cat([self.col_inds[g] for g in range(G)])
        row = torch.cat([self.row_inds[g] for g in range(G)])
        v = cameras.view_rays_pix(col, row)
        c = cameras.cam_pos()
        grad_f_p = torch.cat([pre[g][1] for g in range(G)], dim=0)
        grad_d_p = torch.cat([pre[g][2] for g in range(G)], dim=0)
        vd = v.detach()
        zeros = torch.zeros_like(vd[:, 0])
        v_cross = torch.stack([torch.stack([zeros, -vd[:, 2], vd[:, 1]], -1),
                               torch.stack([vd[:, 2], zeros, -vd[:, 0]], -1),
                               torch.stack([-vd[:, 1], vd[:, 0], zeros], -1)], dim=1)       # [v]_x
        a1 = (v_cross.unsqueeze(-1) * grad_d_p.unsqueeze(-3)).sum(-2)                       # v_cross @ J
        b = torch.cat([grad_f_p.view(-1, 1, 3), a1], dim=1)                                 # [P,4,3]
        btb = (b.unsqueeze(-1) * b.unsqueeze(-2)).sum(1)                                    # b^T b
        btb_inv, check = Fast3x3Minv(btb.contiguous())
        for g, name in enumerate(self.garment_names[:G]):
            ck = check[:n[0]] if g == 0 else check[n[0]:]
            self.info['{}_invInfo'.format(name)] = (ck.numel(), ck.sum())
        rhs_1 = (btb_inv.unsqueeze(-1) * b.permute(0, 2, 1).unsqueeze(-3)).sum(-2)           # [P,3,4]
        rhs_1 = (grad_l_p.view(-1, 3, 1) * rhs_1).sum(1, keepdim=True)                      # [P,1,4]
        targets, grads = [], []
        off = 0
        for g in range(G):
            net = self.garment_nets[g]
            params = [q for q in net.parameters() if q.requires_grad]
            if params:                     # frozen in the large-pose stage (OptimGarmentNetwork_Large_Pose.py:440-452)
                pg = torch.autograd.grad(net(ps[g], ratio), params, -rhs_1[off:off + n[g], :, 0])
                targets += params
                grads += list(pg)
            off += n[g]
        params = [q for q in self.deformer.parameters() if q.requires_grad]
        N = int(poses.shape[0])
        b_all = torch.cat([self.batch_inds[g] for g in range(G)])
        cidx = torch.cat([self.batch_inds[g] + N * g for g in range(G)])
        tables = [d_cond_list[1 + g] for g in range(G)]
        d = self.deformer(torch.cat(ps, dim=0), [torch.cat(tables, dim=0), [poses, trans]], b_all, ratio=ratio,
                          offset_type=self.garment_names[G - 1], cond_index=cidx)
        temp = -(rhs_1[:, :, -3:].transpose(1, 2) * v_cross).sum(1)                          # rhs[1:4] @ (-[v]_x)
        opt_defconds = [t for t in tables + [poses, trans] if t.requires_grad]
        pg = torch.autograd.grad(d, params + opt_defconds, temp)
        targets += params + opt_defconds
        grads += list(pg)
        if v.requires_grad:
            dc = d.detach() - c.detach().view(1, 3)
            dc_cross = torch.stack([torch.stack([zeros, -dc[:, 2], dc[:, 1]], -1),
                                    torch.stack([dc[:, 2], zeros, -dc[:, 0]], -1),
                                    torch.stack([-dc[:, 1], dc[:, 0], zeros], -1)], dim=1)
            targets.append(v)
            grads.append((rhs_1[:, :, -3:].transpose(1, 2) * dc_cross).sum(1))
        if c.requires_grad:
            targets.append(c)
            grads.append((-temp.sum(0)).view_as(c))
        _inject_gradients(targets, grads)
        return True

    # ------------------------------------------------------------------------------------------ one step
    def iters_per_epoch(self):
        """ceil(F / (batch_size * world_size)): the reference's DataLoader keeps the short last batch (drop_last=False,
        dataset/dataset.py:1159-1183; train.py:250-260 counts ceil(len/bs) iterations per epoch)."""
        return iters_per_epoch(_n_frames(self.dataset), self.batch_size, self.world_size)

    def frame_batch_at(self, epoch, pos):
        """Frames of this rank for position `pos` of `epoch`: a seeded permutation of all frames dealt round-robin over
        ranks (the reference's RandomSampler, dataset/dataset.py:1135-1157, sharded — SURVEY.md §8e).  The last
        position of an epoch holds the remaining F mod (batch_size*world_size) frames; when that is fewer than one
        frame per rank, the permutation wraps so that every rank still has a frame (all ranks must enter the
        collectives)."""
        per_it = self.batch_size * self.world_size
        perm = torch.randperm(_n_frames(self.dataset), generator=torch.Generator().manual_seed(1234 + epoch))
        ids = perm[pos * per_it:(pos + 1) * per_it]
        if ids.numel() < self.world_size:
            ids = torch.cat([ids, perm[:self.world_size - ids.numel()]])
        ids = ids[self.rank::self.world_size][:self.batch_size]
        if torch.device(self.device).type == 'cuda':      # pinned + non_blocking: a pageable copy synchronises the stream (a drain per step)
            return ids.pin_memory().to(self.device, non_blocking=True)
        return ids.to(self.device)

    def frame_batch(self, it):
        return self.frame_batch_at(*divmod(it, self.iters_per_epoch()))

    def rebuild_optimizer(self, lr=None):
        """New Adam over the dataset's learnable tensors and the networks (train.py:213, :229-231 after a resume)."""
        params = [p for m in (self.netRender, self.deformer, self.garment_nets) for p in m.parameters()
                  if p.requires_grad]                                  # train.py:213: `if p.requires_grad`
        lr = self.conf_all.get_float('train.learning_rate') if lr is None else lr
        self.optimizer = torch.optim.Adam(self.dataset.learnable_weights() + params, lr=lr)
        return self.optimizer

    def backward(self, loss):
        """`loss.backward()` for the loss forward() has just returned (train.py:325) — started from the RAY stream when forward left
        the still-undifferentiated terms there (`_tail`: render loss + pose prior; the |SDF| terms were differentiated inside forward,
        _backward_early): the same nodes and the same accumulation order as `loss.backward()`, whose root would wait in the main
        stream's queue behind the |SDF| terms' large products.  The main stream joins behind it."""
        tail, self._tail = getattr(self, '_tail', None), None
        if tail is not None and tail[0].requires_grad:
            # The |SDF| terms' backward may still be adding to the shared `.grad`s on the main stream: the tail's gradients are taken
            # as VALUES on the ray stream (autograd.grad: nothing accumulated there) and added on the main stream behind the join —
            # one addition per leaf, in the order (mask loss) + (|SDF| terms) + (render loss + pose prior) the plain backward has.
            main = torch.cuda.current_stream(self.device)
            leaves = [q for q in self.shared_parameters() if q.requires_grad]
            leaves += [t for t in (getattr(self, 'TmpPs', None) or []) if t is not None and t.requires_grad]
            with torch.cuda.stream(tail[1]):
                grads = torch.autograd.grad(tail[0], leaves, allow_unused=True)
            main.wait_stream(tail[1])
            have_t, have_g, fresh = [], [], []
            for q, g in zip(leaves, grads):
                if g is None:
                    continue
                g.record_stream(main)
                if q.grad is None:
                    fresh.append((q, g))
                else:
                    have_t.append(q.grad)
                    have_g.append(g)
            with torch.no_grad():
                if have_t:
                    torch._foreach_add_(have_t, have_g)
                for q, g in fresh:
                    q.grad = g
        elif torch.is_tensor(loss) and loss.requires_grad:
            loss.backward()

    def step(self, it, allreduce=None, frame_ids=None):
        """train.py:317-328.  `allreduce(list_of_tensors)` is called on the gradients before each optimizer step
        when frames are sharded over ranks."""
        frame_ids = self.frame_batch(it) if frame_ids is None else frame_ids
        ratio = {'sdfRatio': 1., 'deformerRatio': self.opt_times / 2500. + 0.5, 'renderRatio': 1.}
        self._allreduce = allreduce
        loss = HotLoop.forward(self, frame_ids, ratio)      # (the facade subclass overrides forward(datas, ...))
        with self._phase('backward'):
            self.backward(loss)
        pending = []
        if allreduce is not None and hasattr(allreduce, 'start'):
            # the colour net and the per-frame colour codes are final after the backward (the implicit differentiation below adds to
            # the SDF nets, the deformer, the deformation codes, poses / translations and the camera only, :2269-2313): their
            # all-reduce travels while propagateTmpPsGrad runs
            early = self.early_shared_parameters()
            pending.append(allreduce.start(early))
        with self._phase('propagate'):
            self.propagateTmpPsGrad(frame_ids, ratio)
        with self._phase('allreduce+adam'):
            if allreduce is not None:
                if pending:
                    ids = {id(p) for p in early}
                    pending.append(allreduce.start([p for p in self.shared_parameters() if id(p) not in ids]))
                    for h in pending:
                        allreduce.finish(h)
                else:
                    allreduce([p for p in self.shared_parameters()])
            self.optimizer.step()
        self.opt_times += 1.
        return loss.detach(), self.info['rays_total']


class FrameLoader:
    """Stands in for the reference's `DataLoader(dataset, batch_size, sampler=RandomSampler(dataset, 1, shuffle))`
    (dataset/dataset.py:1135-1183) in `for data_index, (frame_ids, outs) in enumerate(dataloader)` (train.py:317): per
    epoch a seeded permutation of the frames in mini-batches of the current stage's batch size (the short last batch
    kept), dealt round-robin over the frame-sharded ranks; `outs` is the mini-batch dict (`dataset.get_batch`)."""

    def __init__(self, loop, epoch=0):
        self.loop, self.epoch = loop, epoch
        self.dataset = loop.dataset

    def set_epoch(self, epoch):
        self.epoch = epoch
        return self

    def __len__(self):
        return self.loop.iters_per_epoch()

    def __iter__(self):
        for pos in range(len(self)):
            frame_ids = self.loop.frame_batch_at(self.epoch, pos)
            yield frame_ids, self.dataset.get_batch(frame_ids, self.loop.mask_keys)


class _NoStream:
    """Stands in for a HIP stream where there is none (the dependency-graph order on the host)."""

    def wait_event(self, ev):
        pass

    def wait_stream(self, other):
        pass


def fan_mesh(curve_pts):
    """Closed curve [S,3] -> (vertices [S+1,3] = curve + centroid, faces [S,3] = (i, i+1, centre), the last one
    wrapping) — OptimGarmentNetwork.py:797-806."""
    S = curve_pts.shape[0]
    verts = torch.cat([curve_pts, curve_pts.mean(0, keepdim=True)], dim=0)
    i = torch.arange(S, device=curve_pts.device)
    faces = torch.stack([i, (i + 1) % S, torch.full_like(i, S)], dim=1)
    return verts, faces


def sample_fan_mesh(verts, faces, count, generator=None):
    """`trimesh.sample.sample_surface` (trimesh 3.10.5, third party — restated from its published algorithm, parity
    unpinned) on the device: faces picked with probability proportional to their area (inverse CDF), a uniform point
    of each picked triangle from two uniforms reflected into the lower-left half of the unit square."""
    tri = verts[faces]                                                                     # [F,3,3]
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    area = 0.5 * torch.linalg.cross(e1, e2, dim=-1).norm(dim=-1)
    cum = torch.cumsum(area, 0)
    u = torch.rand(count, device=verts.device, generator=generator) * cum[-1]
    f = torch.searchsorted(cum, u).clamp_(max=faces.shape[0] - 1)
    r = torch.rand(count, 2, device=verts.device, generator=generator)
    r = torch.where((r.sum(1, keepdim=True) > 1.0), r - 1.0, r).abs()
    return tri[f, 0] + e1[f] * r[:, 0:1] + e2[f] * r[:, 1:2]


def _host_subset(verts, expected):
    """`verts[torch.rand(V) < expected / V].detach()` as the reference draws it (OptimGarmentNetwork.py:1108, :1138): the Bernoulli
    mask comes from torch's HOST generator, so the size of the subset is known without asking the device — the selected rows travel
    as a pinned index list (the compare and the index extraction in numpy: torch would fork its intra-op pool for ~1e5 elements)."""
    V = verts.shape[0]
    sel = torch.rand(V).numpy() < float(expected) / float(V)
    idx = torch.from_numpy(np.flatnonzero(sel))
    if verts.is_cuda:
        idx = idx.pin_memory().to(verts.device, non_blocking=True)
    return verts.detach().index_select(0, idx)


def _nonzero_known(mask, count):
    """Indices of the `count` set entries of a 1-D boolean mask, in increasing order, without reading the count back."""
    if hasattr(torch, 'nonzero_static') and mask.is_cuda:
        return torch.nonzero_static(mask, size=int(count)).view(-1)
    return mask.nonzero(as_tuple=True)[0]


def _n_frames(dataset):
    """Frames of the synthetic dataset (`F`) or of a caller's dataset (`frame_num`, dataset/dataset.py:185)."""
    return int(dataset.F) if hasattr(dataset, 'F') else int(getattr(dataset, 'frame_num', len(dataset)))


def _param_versions(*modules):
    """In-place version counters of the modules' parameters (they change at optimizer.step())."""
    return tuple(q._version for m in modules for q in m.parameters())


def L_raw_stream(t):
    """Raw handle of the current stream of t's device (None for CPU tensors)."""
    if not t.is_cuda:
        return None
    from . import _lib
    return _lib.raw_stream(t.device)


def iters_per_epoch(n_frames, batch_size, world_size=1):
    """Optimiser iterations per epoch — the ONE formula behind HotLoop.iters_per_epoch and train.resumed_opt_times."""
    return max(-(-n_frames // (batch_size * world_size)), 1)


def _inject_gradients(targets, grads):
    """target.grad += grad for leaves (one multi-tensor add), one backward call for the non-leaf targets."""
    leaf_t, leaf_g, new_t, nl_t, nl_g = [], [], [], [], []
    for t, g in zip(targets, grads):
        if g is None:
            continue
        g = g.detach()
        if t.is_leaf:
            if t.grad is None:
                new_t.append((t, g))
            else:
                leaf_t.append(t.grad)
                leaf_g.append(g.reshape(t.grad.shape))
        else:
            nl_t.append(t)
            nl_g.append(g.reshape(t.shape))
    with torch.no_grad():
        if leaf_t:
            torch._foreach_add_(leaf_t, leaf_g)
        for t, g in new_t:
            t.grad = g.reshape(t.shape).clone()
    if nl_t:
        torch.autograd.backward(nl_t, nl_g)


@torch.no_grad()
def _zero_level_radius(net, device, ndir=64, iters=24):
    """Mean distance from the origin to the zero level set of `net` along random directions (bisection)."""
    g = torch.Generator().manual_seed(7)
    dirs = F.normalize(torch.randn(ndir, 3, generator=g), dim=1).to(device)
    lo = torch.full((ndir, 1), 0.02, device=device)
    hi = torch.full((ndir, 1), 2.0, device=device)
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        inside = net(dirs * mid, 1.0) < 0
        lo = torch.where(inside, mid, lo)
        hi = torch.where(inside, hi, mid)
    return float((0.5 * (lo + hi)).mean())


def _skeleton(radius):
    """24 joints in the SMPL order (pelvis, hips, spine, knees, ... hands), a generic standing figure scaled so that
    it fits inside the initial SDF sphere of the given radius."""
    j = torch.tensor([
        [0.00, 0.00, 0.00], [0.07, -0.09, 0.00], [-0.07, -0.09, 0.00], [0.00, 0.11, -0.02],
        [0.10, -0.47, 0.00], [-0.10, -0.47, 0.00], [0.00, 0.25, 0.00], [0.09, -0.87, -0.03],
        [-0.09, -0.87, -0.03], [0.00, 0.30, 0.02], [0.11, -0.93, 0.09], [-0.11, -0.93, 0.09],
        [0.00, 0.51, -0.01], [0.08, 0.42, 0.00], [-0.08, 0.42, 0.00], [0.00, 0.60, 0.03],
        [0.19, 0.44, -0.01], [-0.19, 0.44, -0.01], [0.45, 0.43, -0.03], [-0.45, 0.43, -0.03],
        [0.70, 0.43, -0.03], [-0.70, 0.43, -0.03], [0.79, 0.42, -0.04], [-0.79, 0.42, -0.04]])
    j = j - torch.tensor([0.0, -0.16, 0.0])                 # centre the figure (feet -0.93 .. head 0.60)
    return j * (radius / 0.80)


def _apose():
    pose = np.zeros((24, 3), dtype=np.float32)            # utils/utils.py:76-83, init_pose_type 0
    pose[1] = [0, 0, 10. / 180. * np.pi]
    pose[2] = [0, 0, -10. / 180. * np.pi]
    pose[16] = [0, 0, -45. / 180. * np.pi]
    pose[17] = [0, 0, 45. / 180. * np.pi]
    return pose


def _make_stream(device):
    from . import _lib
    return _lib.make_stream(device)
