"""Seeded builders shared by tests/golden/make_golden.py (reference side) and the tests (recmv side).

Networks are never stored in fixtures: both sides construct them with the same seed and the same
construction order, then apply the same seeded perturbation; `fingerprint` proves the parameters agree.
"""
import numpy as np
import torch

RATIOS = [-1.0, 0.0, 0.1, 0.37, 0.5, 0.62, 0.99, 1.0, 1.7]
SDF_GRAD_KEYS = ["lin0.weight_v", "lin4.weight_g", "lin8.bias", "lin3.bias"]
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
                        dtype=np.int64)


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def points(n, seed, scale=1.0):
    return torch.randn(n, 3, generator=_gen(seed)) * scale


def perturb(module, seed, scale):
    g = _gen(seed)
    with torch.no_grad():
        for _, p in module.named_parameters():
            p.add_(scale * torch.randn(p.shape, generator=g))
    return module


def fingerprint(module):
    rows = []
    for _, p in module.named_parameters():
        d = p.detach().double()
        rows.append([d.sum().item(), d.abs().sum().item(), float(p.numel())])
    return np.array(rows)


def build_sdf(getTmpSdf):
    torch.manual_seed(100)
    net = getTmpSdf("cpu", 6)
    return perturb(net, 101, 0.003)


def build_translator(cls):
    torch.manual_seed(200)
    net = cls(128, multires=6)
    return perturb(net, 201, 0.004)


def build_render(cls):
    torch.manual_seed(300)
    net = cls(256, "idr", 9, 3, [512, 512, 512, 512], True, multires_n=0, multires_v=4)
    return perturb(net, 301, 0.004)


def conds_and_inds(n, nframes, condlen, seed):
    conds = (0.1 * torch.randn(nframes, condlen, generator=_gen(seed))).requires_grad_(True)
    binds = torch.randint(0, nframes, (n,), generator=_gen(seed + 1000))
    return conds, binds


def render_inputs(n, seed):
    g = _gen(seed)
    p = torch.randn(n, 3, generator=g) * 0.5
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    view = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    feat = torch.randn(n, 256, generator=g) * 0.5
    return p, nrm, view, feat


def apose():
    pose = np.zeros((24, 3), dtype=np.float32)       # utils/utils.py:76-83 (init_pose_type 0)
    pose[1] = [0, 0, 10. / 180. * np.pi]
    pose[2] = [0, 0, -10. / 180. * np.pi]
    pose[16] = [0, 0, -45. / 180. * np.pi]
    pose[17] = [0, 0, 45. / 180. * np.pi]
    return pose


def skinner_args(dims=(9, 17, 11), seed=400):
    g = _gen(seed)
    D, H, W = dims
    ws = torch.softmax(2.0 * torch.randn(1, 24, D, H, W, generator=g), dim=1)
    Js = 0.3 * torch.randn(24, 3, generator=g)
    return dict(ws=ws, bmins=[-0.8, -1.1, -0.6], bmaxs=[0.8, 1.1, 0.6], Js=Js, parents=SMPL_PARENTS,
                init_pose=apose(), align_corners=False, extra_trans=None,
                bbox_extend=torch.tensor([1.6, 2.2, 1.2]), bbox_center=torch.tensor([0.0, -0.1, 0.05]))


def build_skinner(LBSkinner, batch_rodrigues=None, dims=(9, 17, 11)):
    return LBSkinner(**skinner_args(dims))


def poses_trans(n, seed):
    g = _gen(seed)
    poses = 0.2 * torch.randn(n, 24, 3, generator=g)
    trans = 0.01 * torch.randn(n, 3, generator=g)
    return poses, trans


def rootfind_init(sdf, n, seed):
    """Points close to the zero level set of `sdf` (a few Newton steps from a sphere of radius 0.6)."""
    p = torch.nn.functional.normalize(points(n, seed), dim=1) * 0.6
    for _ in range(4):
        p = p.detach().requires_grad_(True)
        f = sdf(p, 1.0)
        g = torch.autograd.grad(f.sum(), p)[0]
        p = (p - f * g / (g * g).sum(1, keepdim=True)).detach()
    return p


class TrimeshStandIn:
    """Stand-in for `trimesh.Trimesh(vertices, faces, process=False)` with the one method the hot path calls,
    `.sample(count)` = trimesh.sample.sample_surface (trimesh 3.10.5 — third party, absent from /root/reference and from
    this image; restated from its published algorithm: numpy, float64, area-weighted inverse-CDF face pick, two
    uniforms per sample reflected into the triangle).  Draws from `rng` (a numpy RandomState) instead of numpy's global
    state so that both sides of a parity test see the same points."""

    rng = None

    def __init__(self, vertices, faces, process=False):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)

    def sample(self, count):
        rng = TrimeshStandIn.rng
        tri = self.vertices[self.faces]
        area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
        weight_cum = np.cumsum(area)
        face_index = np.searchsorted(weight_cum, rng.random_sample(count) * weight_cum[-1])
        tri_origins = tri[:, 0][face_index]
        tri_vectors = (tri[:, 1:] - tri[:, :1])[face_index]
        random_lengths = rng.random_sample((count, 2, 1))
        random_test = random_lengths.sum(axis=1).reshape(-1) > 1.0
        random_lengths[random_test] -= 1.0
        random_lengths = np.abs(random_lengths)
        return (tri_vectors * random_lengths).sum(axis=1) + tri_origins


def curve_aware_ring(n=200):
    """The `upper_bottom` ring of the curve-aware fixture: a wobbly closed curve close to the zero level of
    build_sdf's net (radius ~0.6)."""
    t = torch.linspace(0, 2 * np.pi, n + 1)[:-1]
    rho = 0.52 + 0.03 * torch.sin(3 * t)
    return torch.stack([rho * torch.cos(t), -0.3 + 0.02 * torch.cos(2 * t), rho * torch.sin(t)], -1).float()
