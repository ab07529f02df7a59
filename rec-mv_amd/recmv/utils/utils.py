"""Hot-path helpers of utils/utils.py (reference lines 8-264), on the recmv kernels.

  FastDiff3x3MinvFunction  utils/utils.py:8-18
  quat2mat                 :21-39
  annealing_weights        :40-46
  GMRobustError            :48-52
  sample_points            :101-111
  compute_Jacobian         :133-156   (batch_compute_Jacobian :158-186)
  compute_deformed_normals :198-230
  compute_cardinal_rays    :232-250
  compute_netRender_color  :252-264
"""
import numpy as np
import torch
from torch.autograd import Function

from ..FastMinv import Fast3x3Minv, Fast3x3Minv_backward

__all__ = ["save_model", "load_model", "set_hierarchical_config", "FastDiff3x3MinvFunction", "quat2mat", "annealing_weights", "GMRobustError", "sample_points",
           "compute_Jacobian", "batch_compute_Jacobian", "compute_deformed_normals", "compute_cardinal_rays",
           "compute_netRender_color", "scatter_mean", "write_ply", "read_ply", "smpl_tmp_Apose"]


class FastDiff3x3MinvFunction(Function):
    """Differentiable batched 3x3 inverse with a validity mask (utils/utils.py:8-18)."""

    @staticmethod
    def forward(ctx, input):
        invs, check = Fast3x3Minv(input.contiguous())
        ctx.save_for_backward(invs, check)
        ctx.mark_non_differentiable(check)
        return invs, check

    @staticmethod
    def backward(ctx, grad_input, grad_check):
        invs, check = ctx.saved_tensors
        return Fast3x3Minv_backward(grad_input.contiguous(), invs), None


def quat2mat(quat):
    """(w,x,y,z) quaternion -> rotation matrix, normalising first (utils/utils.py:21-39)."""
    norm_quat = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = norm_quat[:, 0], norm_quat[:, 1], norm_quat[:, 2], norm_quat[:, 3]
    B = quat.size(0)
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(B, 3, 3)


def annealing_weights(multires, ratio):
    """Coarse-to-fine weights of the positional-encoding bands (utils/utils.py:40-46)."""
    alpha = ratio * multires
    out = []
    for ind in range(multires):
        w = (1. - np.cos(np.pi * min(max(alpha - float(ind), 0.), 1.))) / 2.
        out.extend([w, w])
    return out


def GMRobustError(x, c, square=False):
    if square:
        return 2. * x / (c * c) / (x / (c * c) + 4)
    return 2. * x * x / (c * c) / (x * x / (c * c) + 4)


def sample_points(pc_input, global_sigma, local_sigma, ratio=6):
    """Local gaussian jitter + global uniform samples (utils/utils.py:101-111)."""
    sample_size, dim = pc_input.shape
    sample_local = pc_input + (torch.randn_like(pc_input) * local_sigma)
    if ratio > 0:
        sample_global = (torch.rand(sample_size // ratio, dim, device=pc_input.device) * (global_sigma * 2)) \
            - global_sigma
        return torch.cat([sample_local, sample_global], dim=0)
    return sample_local


def compute_Jacobian(ps, ds, retain_graph, create_graph, allow_unused=False):
    """[P,3,3] Jacobian d(ds)/d(ps), row i = grad of ds[...,i]; relies on per-point independence
    (utils/utils.py:133-156)."""
    jac = getattr(ds, '_recmv_jac', None)
    if jac is not None and jac[0] is ps:                 # carried forward by the MLP jet pass (forward(..., jet=True))
        return jac[1]
    lazy = getattr(ds, '_recmv_jac_lazy', None)
    if lazy is not None and lazy[0] is ps:
        # composite deformer: J = J_skinning(q) . J_offsetMLP(ps); only the skinning stage goes through autograd
        _, q, Jq = lazy
        Jl = compute_Jacobian(q, ds, retain_graph, create_graph, allow_unused)
        return (Jl.unsqueeze(-1) * Jq.unsqueeze(-3)).sum(-2)
    grad_d_p = []
    grad_outputs = torch.ones_like(ds[..., 0])
    outx = torch.autograd.grad(ds[..., 0], ps, grad_outputs, retain_graph=True, create_graph=create_graph,
                               allow_unused=allow_unused)
    grad_d_p.append(outx[0].view(-1, 1, 3))
    outy = torch.autograd.grad(ds[..., 1], ps, grad_outputs, retain_graph=True, create_graph=create_graph,
                               allow_unused=allow_unused)
    grad_d_p.append(outy[0].view(-1, 1, 3))
    outz = torch.autograd.grad(ds[..., 2], ps, grad_outputs, retain_graph=retain_graph,
                               create_graph=create_graph, allow_unused=allow_unused)
    grad_d_p.append(outz[0].view(-1, 1, 3))
    return torch.cat(grad_d_p, dim=1)


def batch_compute_Jacobian(ps, ds, retain_graph, create_graph, allow_unused=False):
    batch = ps.shape[0]
    grad_d_p = []
    grad_outputs = torch.ones_like(ds[..., 0])
    for i in range(3):
        out = torch.autograd.grad(ds[..., i], ps, grad_outputs, retain_graph=True if i < 2 else retain_graph,
                                  create_graph=create_graph, allow_unused=allow_unused)
        grad_d_p.append(out[0].view(batch, -1, 1, 3))
    return torch.cat(grad_d_p, dim=-2)


def _bmv(m, v):
    """Batched 3x3 @ 3 without BLAS: (m[P,3,3], v[P,3]) -> [P,3]."""
    return (m * v.unsqueeze(-2)).sum(-1)


def compute_deformed_normals(sdf, deformer, ps, defconds, batch_inds, ratio, phase, offset_type):
    """Deformed-space normal J^-T grad f, fallback J grad f where J is singular (utils/utils.py:198-230)."""
    sdfs = sdf(ps, ratio)
    check = True if phase == 'train' or phase == 'Train' else False
    onx = torch.autograd.grad(sdfs, ps, torch.ones_like(sdfs), retain_graph=check, create_graph=check)[0]
    ds = deformer(ps, defconds, batch_inds, ratio=ratio, offset_type=offset_type)
    grad_d_p = compute_Jacobian(ps, ds, check, check)
    grad_d_p_inv, inv_mask = FastDiff3x3MinvFunction.apply(grad_d_p)
    nx = _bmv(grad_d_p_inv.transpose(-2, -1), onx.view(-1, 3))
    # rows with a singular Jacobian fall back to J grad f (:221-227).  The reference tests `n_inv_mask.sum().item()`
    # on the host (a device sync per call) before scattering; selecting per row gives the same tensor without it.
    nx = torch.where(inv_mask.view(-1, 1), nx, _bmv(grad_d_p, onx.view(-1, 3)))
    nx = nx / nx.norm(dim=1, keepdim=True)
    return nx, ds


def compute_cardinal_rays(deformer, ps, rays, defconds, batch_inds, ratio, phase, offset_type=None):
    """Canonical-space ray J^-1 v, fallback v where J is singular (utils/utils.py:232-250)."""
    check = True if phase == 'train' or phase == 'Train' else False
    ds = deformer(ps, defconds, batch_inds, ratio=ratio, offset_type=offset_type, jet=True)
    grad_d_p = compute_Jacobian(ps, ds, check, check)
    grad_d_p_inv, inv_mask = FastDiff3x3MinvFunction.apply(grad_d_p)
    crays = _bmv(grad_d_p_inv, rays.view(-1, 3))
    # rows with a singular Jacobian fall back to the (detached) ray itself (:241-247), selected per row without the
    # reference's host-side `.sum().item()` test
    crays = torch.where(inv_mask.view(-1, 1), crays, rays.view(-1, 3).detach())
    crays = crays / crays.norm(dim=1, keepdim=True)
    return crays, ds


def compute_netRender_color(net, ps, ds, ns, vs, features, framefeatures, ratio):
    """`ds` and `framefeatures` are accepted and ignored, as in the reference (utils/utils.py:252-264)."""
    return net(ps, ns, vs, features, ratio)


def scatter_mean(src, index, dim_size):
    """torch_scatter.scatter(src, index, reduce='mean', dim_size=...) for 1-D src (OptimGarmentNetwork.py:1188, :1215),
    summed in a fixed order (ops.rows_sum_by_index) instead of with float atomics."""
    from ..ops import rows_sum_by_index
    out = rows_sum_by_index(src.view(-1, 1), index, dim_size).view(-1)
    cnt = rows_sum_by_index(torch.ones_like(src).detach().view(-1, 1), index, dim_size).view(-1)
    return out / cnt.clamp(min=1)


def set_hierarchical_config(conf, name, optNet, dataloader, resolutions):
    """utils/utils.py:330-348 of the reference: switch `optNet` to stage `name` ('coarse' | 'medium' | 'fine') — new
    batch size and Seg3dLossless pyramid at once, loss weights / point radius / re-mesh period parked until the next
    re-mesh (`next_conf`, `next_train_conf`) — and return `(optNet, dataloader)`.  The frame loader reads the batch size
    from `optNet`, so the same loader comes back (the reference rebuilds its DataLoader for the new batch size)."""
    optNet.set_stage(name, [tuple(int(v) for v in r) for r in resolutions] if resolutions is not None else None)
    return optNet, dataloader


def save_model(name, epoch, optNet, dataset):
    """Checkpoint with the reference's layout (utils/utils.py:350-357): epoch, model_state_dict, the camera parameters
    under their dataset names, poses / trans / shape and the per-frame codes dcond / rcond.  Optimiser state is not
    saved (the reference does not either)."""
    outdic = {"epoch": epoch, "model_state_dict": {k: v.detach().cpu() for k, v in optNet.state_dict().items()}}
    outdic.update({k: v.detach().cpu() for k, v in dataset.camera_params.items()})
    outdic.update({'poses': dataset.poses.detach().cpu(), 'trans': dataset.trans.detach().cpu(),
                   'shape': dataset.shape.detach().cpu(), 'dcond': dataset.conds[0].detach().cpu(),
                   'rcond': dataset.conds[1].detach().cpu()})
    torch.save(outdic, name)


def load_model(name, optNet, dataset, device, subsdfmodel=None, model_rm_prefix=None):
    """utils/utils.py:359-420: drops `engine.*` and the skinner's `ws` volume, optional prefix removal and SDF
    substitution, `alpha_curve` -> `inter_free_curve`, non-strict load; restores the per-frame tensors and camera
    parameters keeping each tensor's requires_grad; returns (optNet, dataset, epoch)."""
    saved = torch.load(name, map_location='cpu')
    state = {k: v for k, v in saved["model_state_dict"].items() if 'engine.' not in k}
    if model_rm_prefix:
        state = {k: v for k, v in state.items() if not any(k[:len(p)] == p for p in model_rm_prefix)}
    if subsdfmodel is not None:
        sdf_model = torch.load(subsdfmodel, map_location='cpu')
        state = {k: v for k, v in state.items() if 'sdf.' not in k}
        state.update({'sdf.' + k: v for k, v in sdf_model.items()})
    state = {k: v for k, v in state.items() if 'deformer.defs.1.ws' not in k}
    for key in list(state.keys()):
        if 'alpha_curve' in key:
            state[key.replace('alpha_curve', 'inter_free_curve')] = state.pop(key)
    optNet.load_state_dict(state, strict=False)
    optNet = optNet.to(device)

    def restore(old, new):
        return new.to(device).requires_grad_(old.requires_grad)

    if 'dcond' in saved:
        dataset.conds[0] = restore(dataset.conds[0], saved['dcond'])
    if 'rcond' in saved:
        dataset.conds[1] = restore(dataset.conds[1], saved['rcond'])
    dataset.poses = restore(dataset.poses, saved['poses'])
    assert dataset.frame_num <= dataset.poses.shape[0]
    dataset.trans = restore(dataset.trans, saved['trans'])
    assert dataset.frame_num <= dataset.trans.shape[0]
    dataset.shape = restore(dataset.shape, saved['shape'])
    dataset.camera_params = {k: restore(v, saved[k]) for k, v in dataset.camera_params.items()}
    return optNet, dataset, saved['epoch']


def write_ply(name, verts, faces):
    """A triangle mesh as a binary little-endian PLY (float32 vertices, int32 faces) — what train.py exports after the
    SDF pre-fit (`initial_sdf_idr_*.ply`, train.py:196-206 through trimesh) and getOptNet reads back (model/network.py:210)."""
    v = np.ascontiguousarray(torch.as_tensor(verts).detach().cpu().numpy(), dtype='<f4').reshape(-1, 3)
    f = np.ascontiguousarray(torch.as_tensor(faces).detach().cpu().numpy(), dtype='<i4').reshape(-1, 3)
    rows = np.empty(f.shape[0], dtype=[('n', 'u1'), ('idx', '<i4', (3,))])
    rows['n'], rows['idx'] = 3, f
    with open(name, 'wb') as fh:
        fh.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                  "property float z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n"
                  % (v.shape[0], f.shape[0])).encode('ascii'))
        fh.write(v.tobytes())
        fh.write(rows.tobytes())


def read_ply(name):
    """(vertices float32 [V,3], faces int64 [F,3]) of a PLY written by `write_ply` (x, y, z vertices; triangle faces)."""
    with open(name, 'rb') as fh:
        header = b''
        while not header.endswith(b'end_header\n'):
            line = fh.readline()
            if not line:
                raise ValueError(name + ': not a PLY file')
            header += line
        text = header.decode('ascii')
        if 'format binary_little_endian 1.0' not in text or 'property list uchar int vertex_indices' not in text:
            raise ValueError(name + ': only the layout write_ply produces is read')
        count = {ln.split()[1]: int(ln.split()[2]) for ln in text.splitlines() if ln.startswith('element ')}
        v = np.frombuffer(fh.read(12 * count['vertex']), dtype='<f4').reshape(-1, 3)
        rows = np.frombuffer(fh.read(13 * count['face']), dtype=[('n', 'u1'), ('idx', '<i4', (3,))])
    if not (rows['n'] == 3).all():
        raise ValueError(name + ': non-triangle face')
    return torch.from_numpy(v.copy()), torch.from_numpy(rows['idx'].astype(np.int64))


def smpl_tmp_Apose(init_pose_type=0):
    """utils/utils.py:68-99 — the canonical pose the skinning volume is baked in: legs spread by 10 / 7 / 15 / 15 degrees, arms
    lowered by 45 / 55 / 55 / 0 degrees for `train.skinner_pose_type` 0..3 (axis-angle [24,3], float32)."""
    assert init_pose_type in (0, 1, 2, 3)
    legs, arms = {0: (10., 45.), 1: (7., 55.), 2: (15., 55.), 3: (15., 0.)}[init_pose_type]
    pose = np.zeros((24, 3))
    pose[1], pose[2] = [0, 0, legs / 180. * np.pi], [0, 0, -legs / 180. * np.pi]
    pose[16], pose[17] = [0, 0, -arms / 180. * np.pi], [0, 0, arms / 180. * np.pi]
    return pose.astype(np.float32)
