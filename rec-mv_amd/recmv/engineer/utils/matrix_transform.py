"""Rigid / scale transforms of the feature-line templates (engineer/utils/matrix_transform.py of the reference): the
6-D rotation parameterisation (:154-193) and the per-line transforms the start-up registration optimises (:13-103).

A "mesh" argument is either a tensor [V,3] or anything with `verts_packed()` (the reference accepts pytorch3d `Meshes`;
here `FeatureLineMesh` below plays that part — the registration only reads the packed vertices and faces and writes
new vertices back).  All functions return a LIST of [V,3] tensors, one per line, as the reference does.
"""
import torch


class FeatureLineMesh:
    """The three members of pytorch3d's `Meshes` the feature-line registration touches (engineer/core/fl_optimizer.py:
    61-70, 164, 519): packed vertices, packed faces and `update_padded` (a new mesh with the same faces)."""

    def __init__(self, verts, faces):
        if isinstance(verts, (list, tuple)):
            assert len(verts) == 1, "one feature line per mesh"
            verts, faces = verts[0], faces[0]
        self._verts, self._faces = verts, faces

    def verts_packed(self):
        return self._verts

    def faces_packed(self):
        return self._faces

    def update_padded(self, new_verts_padded):
        return FeatureLineMesh(new_verts_padded.reshape(-1, 3), self._faces)

    def to(self, device):
        return FeatureLineMesh(self._verts.to(device), self._faces.to(device))


def _verts(meshes):
    return [m.verts_packed() if hasattr(m, 'verts_packed') else m for m in meshes]


def normalize_vector(v):
    """:154-160 — rows scaled to unit length, the length floored at 1e-8."""
    mag = torch.sqrt(v.pow(2).sum(1))
    mag = torch.max(mag, torch.full((1,), 1e-8, dtype=v.dtype, device=v.device))
    return v / mag.view(-1, 1).expand(v.shape[0], v.shape[1])


def cross_product(u, v):
    """:163-173"""
    i = u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1]
    j = u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2]
    k = u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]
    return torch.stack((i, j, k), 1)


def compute_rotation_matrix_from_ortho6d(poses):
    """:178-193 — [B,6] -> [B,3,3]; columns x = n(a), z = n(x × b), y = z × x."""
    x = normalize_vector(poses[:, 0:3])
    z = normalize_vector(cross_product(x, poses[:, 3:6]))
    y = cross_product(z, x)
    return torch.cat((x.view(-1, 3, 1), y.view(-1, 3, 1), z.view(-1, 3, 1)), 2)


def icp_rotate_transfrom(meshes, R_pack, T_Pack):
    """:92-103 — per line v -> R v + T."""
    return [(R @ v.T).T + T for v, R, T in zip(_verts(meshes), R_pack, T_Pack)]


def scale_icp_rotate_transfrom(meshes, R_pack, T_pack, S_pack):
    """:73-91 — per line: the distance of every vertex from the line's centroid is multiplied by max(scale, 0) along its
    own direction, then v -> R v + T."""
    out = []
    for v, R, T, scale in zip(_verts(meshes), R_pack, T_pack, S_pack):
        center = v.mean(0, keepdim=True)
        v_dirs = (v - center) / ((v - center).norm(dim=1, keepdim=True) + 1e-6)
        init_scale = ((v - center) * v_dirs).sum(dim=-1, keepdim=True)
        v = center + torch.clamp_min(scale, 0.) * init_scale * v_dirs
        out.append((R @ v.T).T + T)
    return out


def center_transform(meshes, R, T):
    """:42-71 — rotate every line about its own centroid (move to the origin, R v + T, move back)."""
    vs = _verts(meshes)
    center = torch.cat([v.mean(0, keepdim=True)[None] for v in vs], dim=0)          # [L,1,3]
    identity = torch.eye(3, device=R.device).expand(R.shape[0], 3, 3)
    out = icp_rotate_transfrom(vs, identity, -center)
    out = icp_rotate_transfrom(out, R, T)
    return icp_rotate_transfrom(out, identity, center)


def icp_rotate_center_transform(meshes, R, T):
    """:13-25 — translate by T, then rotate about the (translated) centroid."""
    identity = torch.eye(3, device=R.device).expand(R.shape[0], 3, 3)
    return center_transform(icp_rotate_transfrom(meshes, identity, T), R, torch.zeros_like(T))


def scale_icp_rotate_center_transform(meshes, R, T, scale):
    """:27-38 — scale about the centroid and translate, then rotate about the new centroid."""
    identity = torch.eye(3, device=R.device).expand(R.shape[0], 3, 3)
    return center_transform(scale_icp_rotate_transfrom(meshes, identity, T, scale), R, torch.zeros_like(T))
