"""engineer/core/fl_optimizer.py of the reference — the feature-line projection loss of the loop and the start-up
registration of the feature-line templates.

`fl_proj_loss` (:72-110) is on the hot path (project_2d_loss, every iteration): recmv/curves.py, pinned against the
reference function (tests/golden/curves.npz).

`scale_rigid_optimizer` (:111-519) and `rigid_optimizer` (:520-677) run once before the loop (`initializeFL`,
OptimGarmentNetwork.py:470-486): every template feature line (a thin ribbon mesh cut from the SMPL garment template)
gets a translation, a radial scale about its centroid and a rotation about its centroid such that — skinned to the
annotated frames and projected — it falls on the frame's 2-D annotation where the posed body does not hide it.  Three
phases of Adam on a chamfer loss: (translation + scale), scale alone, rotation (6-D parameterisation).  The result is
written to `<save_path>/init_trans_matrix.pth` (`rigid_R`, `rigid_T`, `rigid_scale`), the file `align_fl` reads.

Here the skinner is recmv's LBSkinner (one fused HIP kernel per call), the body z-buffer comes from the HIP mesh
rasteriser (recmv.raster) and the chamfer is the masked dense one of recmv.curves.  Both registrations are pinned
against the reference functions run on the same synthetic capture (tests/golden/make_golden_startup.py).
"""
import os

import torch

from ...curves import chamfer_distance_sum, fl_proj_loss, zbuff_check  # noqa: F401
from ...utils.constant import INI_FL_SCALE
from ..utils.matrix_transform import (center_transform, compute_rotation_matrix_from_ortho6d, icp_rotate_center_transform,
                                      icp_rotate_transfrom, scale_icp_rotate_center_transform, scale_icp_rotate_transfrom)

VISIBLE_BEHIND_BODY = 0.01          # a template vertex counts as seen while it is less than this behind the body (:258)


def _mesh_parts(mesh):
    if hasattr(mesh, 'verts_packed'):
        return mesh.verts_packed(), mesh.faces_packed()
    return mesh[0], mesh[1]


def _default_rasterizer(cameras, image_hw):
    from ... import raster
    return raster.MeshRasterizer(cameras, image_hw, blur_radius=0., perspective_correct=True, cull_backfaces=False)


def check_zbuf_body(smpl_mesh, N, deform_lbs, poses, trans, cameras, mask_render, img_size, screen_pts):
    """:30-59 — the posed body's depth image read at `screen_pts` ([N,P,>=2] pixel coordinates): rasterise the skinned
    body once per frame, fill the background with the farthest body vertex of the frame, bilinear read (align_corners).
    `mask_render(cameras, (H, W))` builds the rasteriser (None: recmv.raster.MeshRasterizer with the reference's
    maskRender settings); it is called as `rasteriser(verts [N,V,3], faces [F,3])` and must return `.zbuf` [N,H,W,1]."""
    verts, faces = _mesh_parts(smpl_mesh)
    posed = deform_lbs(verts[None].expand(N, -1, 3), [poses, trans])
    width, height = cameras.image_size[0, 0], cameras.image_size[0, 1]        # host tensor: no device read-back
    rast = (mask_render or _default_rasterizer)(cameras, (int(height), int(width)))
    zbuf = rast(posed.detach(), faces).zbuf
    z_max = posed[..., -1].max(-1).values[:, None, None, None].expand_as(zbuf)
    zbuf = torch.where(zbuf == -1., z_max, zbuf)
    u = 2 * screen_pts[..., 0] / img_size[:, 0].view(-1, 1) - 1
    v = 2 * screen_pts[..., 1] / img_size[:, 1].view(-1, 1) - 1
    return zbuff_check(zbuf, torch.stack([u, v], dim=-1))


def update_feature_line_mesh(fl_meshes, rigid_R, rigid_T):
    """:61-64"""
    return [m.update_padded(v[None]) for m, v in zip(fl_meshes, icp_rotate_transfrom(fl_meshes, rigid_R, rigid_T))]


def update_scale_feature_line_mesh(fl_meshes, rigid_R, rigid_T, rigid_scale):
    """:66-70"""
    moved = scale_icp_rotate_transfrom(fl_meshes, rigid_R, rigid_T, rigid_scale)
    return [m.update_padded(v[None]) for m, v in zip(fl_meshes, moved)]


class _Projection:
    """One mini-batch of the registration: skin the current line vertices to the batch's frames, project them, and score
    them against the frames' 2-D feature lines (the body of every loop of :211-470 / :571-668)."""

    def __init__(self, deform_lbs, dataset, device, smpl_mesh=None, mask_render=None, visibility_lines=None):
        self.deform_lbs, self.dataset, self.device = deform_lbs, dataset, device
        self.smpl_mesh, self.mask_render = smpl_mesh, mask_render
        self.visibility_lines = visibility_lines          # the line vertices whose visibility decides (fixed, :164)

    def _skin(self, line_verts, N, poses, trans):
        batch = torch.cat([v[None].expand(N, -1, 3) for v in line_verts], dim=1)
        return self.deform_lbs(batch, [poses, trans])

    def loss(self, frame_ids, batch, line_verts):
        from ...model import RectifiedPerspectiveCameras
        dev = self.device
        frame_ids = frame_ids.long().to(dev)
        N = frame_ids.numel()
        gt_fl_pts, fl_masks = batch['fl_pts'].to(dev), batch['fl_masks'].to(dev)
        n_fl = fl_masks.shape[-1]
        focals, pps, Rs, Ts, H, W = self.dataset.get_camera_parameters(N, dev)
        img_size = torch.tensor([float(W), float(H)], device=dev).view(1, 2).expand(N, 2)
        cameras = RectifiedPerspectiveCameras(focals, pps, Rs, Ts, image_size=[(W, H)]).to(dev)
        grad_params = self.dataset.get_grad_parameters(frame_ids, dev)
        poses, trans = grad_params[0].detach(), grad_params[1].detach()     # the per-frame tensors are not registered here
        split = [v.shape[0] for v in line_verts]
        screen_pts = cameras.transform_points_screen(self._skin(line_verts, N, poses, trans), img_size)
        masks = [fl_masks[:, i:i + 1, None].expand(N, s, 3).float() for i, s in enumerate(split)]
        if self.visibility_lines is not None:
            with torch.no_grad():
                posed = self._skin(self.visibility_lines, N, poses, trans)
                at = cameras.transform_points_screen(posed, img_size)
                body_z = check_zbuf_body(self.smpl_mesh, N, self.deform_lbs, poses, trans, cameras, self.mask_render,
                                         img_size, at)
                visible = (posed[..., -1] - body_z < VISIBLE_BEHIND_BODY).float()
            masks = [m * vis[..., None] for m, vis in zip(masks, torch.split(visible, split, dim=1))]
        gt_list = torch.split(gt_fl_pts, [gt_fl_pts.shape[1] // n_fl for _ in range(n_fl)], dim=1)
        return fl_proj_loss(list(torch.split(screen_pts, split, dim=1)), list(gt_list), masks)


def _initial_pose(n_lines, device):
    pose = torch.zeros(n_lines, 6, device=device)
    pose[:, 0] = 1.
    pose[:, 4] = 1.
    return pose


def _line_meshes(fl_meshes, fl_infos, device):
    return [fl_meshes[name].to(device) for name in fl_infos]


def _fit(params, lr, epochs, loader, step_loss, log):
    opt = torch.optim.Adam(params, lr=lr, weight_decay=0.)
    for epoch in range(epochs):
        for batch_id, (frame_ids, batch) in enumerate(loader):
            opt.zero_grad()
            loss = step_loss(frame_ids, batch)
            loss.backward()
            opt.step()
            if log is not None:
                log("{}/{}/{}: fl_loss: {}".format(epoch, batch_id, len(loader), float(loss.detach())))


def scale_rigid_optimizer(deform_lbs, fl_meshes, smpl_mesh, mask_render, dataset, data_dataloader, save_path, fl_infos,
                          rigid_R_type='o6', device='cuda:0', log=print):
    """:111-519.  `fl_meshes`: {line name: mesh}; `smpl_mesh`: the canonical body (vertices, faces); `mask_render`: see
    check_zbuf_body; `dataset`: `get_init_fl_datasets`, `get_camera_parameters`, `get_grad_parameters`;
    `data_dataloader`: its batch size / sampler / worker count size the loader over the annotated frames.  Returns the
    registered line meshes (in `fl_infos` order) and writes `<save_path>/init_trans_matrix.pth`; when that file exists the
    stored transform is applied and nothing is optimised (:165-207)."""
    if rigid_R_type != 'o6':
        raise NotImplementedError
    matrix_path = os.path.join(save_path, 'init_trans_matrix.pth')
    n_lines = len(fl_meshes.keys())
    lines = _line_meshes(fl_meshes, fl_infos, device)
    if os.path.exists(matrix_path):
        stored = torch.load(matrix_path)
        moved = scale_icp_rotate_center_transform(lines, stored['rigid_R'].to(device), stored['rigid_T'].to(device),
                                                  stored['rigid_scale'].to(device))
        return [m.update_padded(v[None]) for m, v in zip(lines, moved)]
    loader = dataset.get_init_fl_datasets(data_dataloader.batch_size, data_dataloader.sampler, data_dataloader.num_workers)
    T_epoch = max(150 // len(loader), 2)
    S_epoch = min(max(150 // len(loader), 2), 10)
    os.makedirs(save_path, exist_ok=True)
    rigid_pose = _initial_pose(n_lines, device)
    rigid_T = torch.zeros(n_lines, 1, 3, device=device, requires_grad=True)
    rigid_scale = torch.tensor([INI_FL_SCALE[name] for name in fl_meshes.keys()], device=device).float().requires_grad_(True)
    identity = compute_rotation_matrix_from_ortho6d(rigid_pose)
    with torch.no_grad():                  # visibility is always judged on the lines at their initial scale (:164)
        start = scale_icp_rotate_transfrom(lines, identity, rigid_T, rigid_scale)
    proj = _Projection(deform_lbs, dataset, device, smpl_mesh, mask_render, visibility_lines=start)

    def scaled(frame_ids, batch):
        return proj.loss(frame_ids, batch, scale_icp_rotate_transfrom(lines, identity, rigid_T, rigid_scale))

    log and log("training global rigid_T!")
    _fit([rigid_T, rigid_scale], 0.005, T_epoch, loader, scaled, log)
    rigid_T.requires_grad = False
    log and log("training global rigid_scale!")
    _fit([rigid_scale], 0.005, S_epoch, loader, scaled, log)
    # a garment is symmetric: both cuffs (both trouser legs) take the larger of their two scales (:383-396)
    rigid_scale = rigid_scale.detach().clone()
    index = {name: i for i, name in enumerate(fl_infos)}
    for left, right in (('left_cuff', 'right_cuff'), ('left_pant', 'right_pant')):
        if left in index:
            both = torch.maximum(rigid_scale[index[left]], rigid_scale[index[right]])
            rigid_scale[index[left]] = both
            rigid_scale[index[right]] = both
    lines = update_scale_feature_line_mesh(lines, identity, rigid_T, rigid_scale)
    log and log("training global rigid_R!")
    rigid_pose = _initial_pose(n_lines, device).requires_grad_(True)
    no_T = torch.zeros(n_lines, 1, 3, device=device)

    def rotated(frame_ids, batch):
        return proj.loss(frame_ids, batch, center_transform(lines, compute_rotation_matrix_from_ortho6d(rigid_pose), no_T))

    _fit([rigid_pose], 0.001, T_epoch, loader, rotated, log)
    with torch.no_grad():
        rigid_R = compute_rotation_matrix_from_ortho6d(rigid_pose)
        moved = center_transform(lines, rigid_R, no_T)
    torch.save({'rigid_R': rigid_R.detach().cpu(), 'rigid_T': rigid_T.detach().cpu(),
                'rigid_scale': rigid_scale.detach().cpu()}, matrix_path)
    return [m.update_padded(v[None]) for m, v in zip(lines, moved)]


def rigid_optimizer(deform_lbs, fl_meshes, dataset, train_data_dataloader, save_path, fl_infos, rigid_R_type='o6',
                    device='cuda:0', log=print):
    """:520-677 — the registration without scale and without the body visibility test: two epochs of translation, two of
    rotation about the centroids, over `train_data_dataloader` itself."""
    if rigid_R_type != 'o6':
        raise NotImplementedError
    matrix_path = os.path.join(save_path, 'init_trans_matrix.pth')
    n_lines = len(fl_meshes.keys())
    lines = _line_meshes(fl_meshes, fl_infos, device)
    if os.path.exists(matrix_path):
        stored = torch.load(matrix_path)
        moved = icp_rotate_center_transform(lines, stored['rigid_R'].to(device), stored['rigid_T'].to(device))
        return [m.update_padded(v[None]) for m, v in zip(lines, moved)]
    os.makedirs(save_path, exist_ok=True)
    identity = compute_rotation_matrix_from_ortho6d(_initial_pose(n_lines, device))
    rigid_T = torch.zeros(n_lines, 1, 3, device=device, requires_grad=True)
    proj = _Projection(deform_lbs, dataset, device)
    log and log("training global rigid_T!")
    _fit([rigid_T], 0.005, 2, train_data_dataloader,
         lambda fids, batch: proj.loss(fids, batch, icp_rotate_transfrom(lines, identity, rigid_T)), log)
    rigid_T.requires_grad = False
    lines = update_feature_line_mesh(lines, identity, rigid_T)
    log and log("training global rigid_R!")
    rigid_pose = _initial_pose(n_lines, device).requires_grad_(True)
    no_T = torch.zeros(n_lines, 1, 3, device=device)
    last = {}

    def rotated(fids, batch):
        last['R'] = compute_rotation_matrix_from_ortho6d(rigid_pose)
        return proj.loss(fids, batch, center_transform(lines, last['R'], no_T))

    _fit([rigid_pose], 0.001, 2, train_data_dataloader, rotated, log)
    with torch.no_grad():
        # the reference stores (and applies) the rotation of the last step's forward pass, one Adam update behind the
        # parameter (:669-670); kept, so that the stored file and the returned meshes agree with its output
        rigid_R = last['R'].detach() if 'R' in last else compute_rotation_matrix_from_ortho6d(rigid_pose)
        moved = center_transform(lines, rigid_R, no_T)
    torch.save({'rigid_R': rigid_R.detach().cpu(), 'rigid_T': rigid_T.detach().cpu()}, matrix_path)
    return [m.update_padded(v[None]) for m, v in zip(lines, moved)]
