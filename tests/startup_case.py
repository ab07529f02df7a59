"""The start-up cases shared by tests/golden/make_golden_startup.py (which runs the REFERENCE on them) and the tests (which run
recmv on them, on the CPU port and on the GPU): the capture, the template feature lines and the body of the registration, the
point cloud of the SDF pre-fit."""
import math
import os
import shutil

import torch

CONDS = {'deformer': 16, 'render': 8}
FL_SAMPLING = 30
BATCH = 4
LINE_NAMES = ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'left_pant', 'right_pant']     # FL_INFOS['female-3-casual']
PREFIT_BATCH, PREFIT_EPOCHS = 256, 3
PREFIT_KEYS = ["lin0.weight_v", "lin4.weight_g", "lin8.bias", "lin8.weight_v"]
PREFIT_ROWS = 24                      # leading rows of every compared parameter kept in the fixture


def write_capture(root):
    """tests/capture_fixture.py at 48 x 40 with the loop's pinhole, without the optional normal maps (a mini-batch of frames
    with and without them does not collate — in the reference either)."""
    import capture_fixture as cf
    cf.write_capture(root, seed=7, H=48, W=40, loop_camera=True)
    shutil.rmtree(os.path.join(root, 'normals'))
    return root


def _ribbon(center, axis, radius, n, half_width=0.012):
    """A closed band of 2n vertices / 2n triangles around `axis` (0 = x, 1 = y)."""
    t = torch.linspace(0, 2 * math.pi, n + 1)[:-1]
    c, s = radius * torch.cos(t), radius * torch.sin(t)
    rows = []
    for off in (-half_width, half_width):
        o = torch.full_like(t, off)
        rows.append(torch.stack([o, c, s], -1) if axis == 0 else torch.stack([c, o, s], -1))
    verts = torch.cat(rows, 0) + torch.tensor(center).view(1, 3)
    i = torch.arange(n)
    j = (i + 1) % n
    faces = torch.cat([torch.stack([i, j, i + n], -1), torch.stack([j, j + n, i + n], -1)], 0)
    return verts.float(), faces.long()


def _uv_sphere(radius, n_lat=9, n_lon=14):
    th = torch.linspace(0, math.pi, n_lat + 2)[1:-1]
    ph = torch.linspace(0, 2 * math.pi, n_lon + 1)[:-1]
    ring = torch.stack([torch.sin(th)[:, None] * torch.cos(ph)[None], torch.cos(th)[:, None].expand(-1, n_lon),
                        torch.sin(th)[:, None] * torch.sin(ph)[None]], -1).reshape(-1, 3)
    verts = torch.cat([ring, torch.tensor([[0., 1., 0.], [0., -1., 0.]])], 0) * radius
    faces = []
    for a in range(n_lat - 1):
        for b in range(n_lon):
            p, q = a * n_lon + b, a * n_lon + (b + 1) % n_lon
            faces += [[p, q, p + n_lon], [q, q + n_lon, p + n_lon]]
    top, bottom = n_lat * n_lon, n_lat * n_lon + 1
    for b in range(n_lon):
        faces.append([top, (b + 1) % n_lon, b])
        faces.append([bottom, (n_lat - 1) * n_lon + b, (n_lat - 1) * n_lon + (b + 1) % n_lon])
    return verts.float(), torch.tensor(faces).long()


def geometry(_unused=None):
    """Template feature lines (ribbons of different sizes around a body-sized blob) and the canonical body."""
    spec = {'neck': ((0., 0.30, 0.), 1, 0.12, 10), 'left_cuff': ((0.42, 0.10, 0.), 0, 0.07, 8),
            'right_cuff': ((-0.42, 0.10, 0.), 0, 0.07, 8), 'upper_bottom': ((0., -0.05, 0.), 1, 0.30, 16),
            'left_pant': ((0.12, -0.45, 0.), 1, 0.09, 9), 'right_pant': ((-0.12, -0.45, 0.), 1, 0.09, 9)}
    verts, faces = zip(*[_ribbon(*spec[n]) for n in LINE_NAMES])
    body_v, body_f = _uv_sphere(0.33)
    return dict(line_verts=torch.cat(verts, 0), line_faces=torch.cat(faces, 0), body_verts=body_v, body_faces=body_f,
                line_split=torch.tensor([0] + list(torch.tensor([v.shape[0] for v in verts]).cumsum(0))),
                face_split=torch.tensor([0] + list(torch.tensor([f.shape[0] for f in faces]).cumsum(0))))


def prefit_points(n=700):
    g = torch.Generator().manual_seed(51)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    vs = d * (0.5 + 0.03 * torch.randn(n, 1, generator=g))
    ns = torch.nn.functional.normalize(d + 0.1 * torch.randn(n, 3, generator=g), dim=1)
    return vs, ns


def prefit_probe(n=50):
    return torch.randn(n, 3, generator=torch.Generator().manual_seed(52)) * 0.5


class StandInSMPL:
    """What the start-up steps ask of the reference's SMPL object (smpl_pytorch.SMPL, un-vendored): a linear shape space on a
    6890-vertex template turned by the root rotation, a joint regressor to the 19 'cocoplus' joints, per-vertex blend weights,
    the kinematic tree, faces.  NOT a body model — both sides of a parity test are handed the same one."""

    def __init__(self, seed=61, rodrigues=None):
        import common_setup as cs
        g = torch.Generator().manual_seed(seed)
        d = torch.nn.functional.normalize(torch.randn(6890, 3, generator=g), dim=1)
        self.template = d * torch.tensor([0.28, 0.75, 0.16]) * (0.8 + 0.2 * torch.rand(6890, 1, generator=g))
        self.shapedirs = 0.01 * torch.randn(10, 6890 * 3, generator=g)
        self.joint_regressor = torch.softmax(4 * torch.randn(6890, 19, generator=g), dim=0)
        self.weight = torch.softmax(3 * torch.randn(6890, 24, generator=g), dim=1)
        self.J_template = 0.3 * torch.randn(24, 3, generator=g)
        self.J_dirs = 0.01 * torch.randn(10, 24 * 3, generator=g)
        self.parents = cs.SMPL_PARENTS
        self.faces = torch.randint(0, 6890, (200, 3), generator=g).numpy()
        self.rodrigues = rodrigues

    def to(self, device):
        for k in ('template', 'shapedirs', 'joint_regressor', 'weight', 'J_template', 'J_dirs'):
            setattr(self, k, getattr(self, k).to(device))
        return self

    def skeleton(self, betas, require_body=False):
        return (self.J_template[None] + (betas @ self.J_dirs).view(-1, 24, 3)), None

    def __call__(self, betas, poses, get_skin=False):
        from recmv.model import batch_rodrigues
        rod = self.rodrigues or batch_rodrigues
        v = self.template[None] + (betas @ self.shapedirs).view(-1, 6890, 3)
        R = rod(poses.reshape(-1, 24, 3)[:, 0])
        return torch.matmul(v, R.transpose(1, 2)), None, None


JOINTS = 17


def write_joint_capture(root):
    """`write_capture` with 17 COCO joints per frame in pixels (x, y, visible) — what `smpl_beta_optimizer` consumes (the base
    fixture's 49-joint TCMR rows only exercise the reader)."""
    import capture_fixture as cf
    import joblib
    import numpy as np
    write_capture(root)
    rng = np.random.RandomState(3)
    joints = np.concatenate([rng.uniform(5, 35, (cf.FRAMES, JOINTS, 2)), (rng.rand(cf.FRAMES, JOINTS, 1) > 0.2).astype(np.float64)],
                            -1).astype(np.float32)
    joblib.dump([None, {'gt_joints2d': joints, 'frame_ids': np.arange(cf.FRAMES), 'pose': rng.randn(cf.FRAMES, 72).astype(np.float32),
                        'betas': rng.randn(cf.FRAMES, 10).astype(np.float32)}], os.path.join(root, '%s_tcmr_output.pkl' % cf.GARMENT_TYPE))
    return root


# ----------------------------------------------------------------------------------------------- recmv side
def line_meshes(g, device):
    from recmv.engineer.utils.matrix_transform import FeatureLineMesh
    ls, fs = [int(v) for v in g['reg_line_split']], [int(v) for v in g['reg_face_split']]
    return {n: FeatureLineMesh(g['reg_line_verts'][ls[i]:ls[i + 1]].to(device), g['reg_line_faces'][fs[i]:fs[i + 1]].long().to(device))
            for i, n in enumerate(LINE_NAMES)}


def run_registration(g, root, device, rtol=2e-3):
    """recmv's scale_rigid_optimizer / rigid_optimizer on the capture under `root`, against startup.npz; returns the largest
    deviation of every compared quantity relative to its magnitude."""
    import random
    import capture_fixture as cf
    import common_setup as cs
    from recmv.dataset import RandomSampler, SceneDataset
    from recmv.engineer.core import fl_optimizer as fo
    from recmv.engineer.utils.matrix_transform import FeatureLineMesh
    from recmv.model import LBSkinner
    sk = cs.build_skinner(LBSkinner).to(device)
    torch.manual_seed(31)
    ds = SceneDataset(root, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=FL_SAMPLING, curve_sampling=1)
    for t in ds.conds + [ds.poses, ds.trans] + list(ds.camera_params.values()):
        t.data = t.data.to(device)
    body = FeatureLineMesh(g['reg_body_verts'].to(device), g['reg_body_faces'].long().to(device))
    cat = lambda meshes: torch.cat([m.verts_packed() for m in meshes], 0).cpu()
    worst = {}

    def close(got, want, what):
        worst[what] = float((got.cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-12)
        torch.testing.assert_close(got.cpu(), want, rtol=rtol, atol=rtol * float(want.abs().max()), msg=lambda m: what + ': ' + m)

    save_path = os.path.join(root, 'fl_init')
    random.seed(33)
    torch.manual_seed(33)
    data_loader = torch.utils.data.DataLoader(ds, BATCH, sampler=RandomSampler(ds, 1, False), num_workers=0)
    got = fo.scale_rigid_optimizer(sk, line_meshes(g, device), body, None, ds, data_loader, save_path, LINE_NAMES,
                                   device=device, log=None)
    stored = torch.load(os.path.join(save_path, 'init_trans_matrix.pth'))
    close(stored['rigid_scale'], g['srig_scale'], 'scale')
    close(stored['rigid_T'], g['srig_T'], 'translation')
    close(stored['rigid_R'], g['srig_R'], 'rotation')
    close(cat(got), g['srig_verts'], 'registered lines')
    again = fo.scale_rigid_optimizer(sk, line_meshes(g, device), body, None, ds, data_loader, save_path, LINE_NAMES,
                                     device=device, log=None)
    close(cat(again), g['srig_reapplied'], 're-applied stored transform')
    save_path = os.path.join(root, 'fl_init_rigid')
    random.seed(34)
    torch.manual_seed(34)
    train_loader = ds.get_init_fl_datasets(BATCH, None, 0)
    got = fo.rigid_optimizer(sk, line_meshes(g, device), ds, train_loader, save_path, LINE_NAMES, device=device, log=None)
    stored = torch.load(os.path.join(save_path, 'init_trans_matrix.pth'))
    close(stored['rigid_T'], g['rig_T'], 'rigid translation')
    close(stored['rigid_R'], g['rig_R'], 'rigid rotation')
    close(cat(got), g['rig_verts'], 'rigidly registered lines')
    again = fo.rigid_optimizer(sk, line_meshes(g, device), ds, train_loader, save_path, LINE_NAMES, device=device, log=None)
    close(cat(again), g['rig_reapplied'], 're-applied rigid transform')
    return worst


def run_prefit(g, device, rtol=5e-3, atol_rel=2e-3):
    """HotLoop.initializeSDF on recmv's SDF net against the reference method's parameters after the same three epochs."""
    import tempfile
    import types
    import common_setup as cs
    from composite_cases import host_draws
    from recmv.loop import HotLoop
    from recmv.model import getTmpSdf
    for with_normals in (True, False):
        net = cs.build_sdf(getTmpSdf).to(device)
        vs, ns = g['prefit_vs'].to(device), g['prefit_ns'].to(device)
        opt = torch.optim.Adam([{"params": net.parameters(), "lr": 0.005, "weight_decay": 0}])
        sche = torch.optim.lr_scheduler.StepLR(opt, 2, 0.5)
        with tempfile.TemporaryDirectory() as tmp:
            name = os.path.join(tmp, 'initial_sdf_idr_6_1.pth')
            torch.manual_seed(41)
            with host_draws():
                HotLoop.initializeSDF(types.SimpleNamespace(), net, opt, sche, PREFIT_BATCH, PREFIT_EPOCHS, device, vs, ns,
                                      with_normals, name, log=None)
            stored = torch.load(name, map_location='cpu')
        tag = 'prefit%d_' % int(with_normals)
        params = dict(net.named_parameters())
        for k in PREFIT_KEYS:
            want = g[tag + k.replace('.', '_')]
            assert torch.equal(stored[k], params[k].detach().cpu())
            got = params[k].detach().cpu()[:PREFIT_ROWS]
            assert torch.allclose(got, want, rtol=rtol, atol=atol_rel * float(want.abs().max())), (
                tag + k, float((got - want).abs().max()), float(want.abs().max()))
        assert abs(opt.param_groups[0]['lr'] - float(g[tag + 'lr'])) < 1e-8        # (stored as float32)
        with torch.no_grad():
            probe = net(prefit_probe().to(device), -1).cpu()
        torch.testing.assert_close(probe, g[tag + 'probe'], rtol=rtol, atol=atol_rel)


def run_skinner_baking(g, device, rtol=2e-4):
    """compute_lbswField / smooth_weights / initialLBSkinner on the stand-in model against the reference functions."""
    from recmv.model import compute_lbswField, initialLBSkinner, smooth_weights
    from recmv.utils import smpl_tmp_Apose
    torch.testing.assert_close(smooth_weights(g['bake_smooth_in'].clone().to(device), 4).cpu(), g['bake_smooth_out'], rtol=1e-5, atol=1e-6)
    field = compute_lbswField([-0.5, -0.8, -0.3], torch.tensor([0.5, 0.8, 0.3]).to(device), (7, 9, 5), g['bake_verts'].to(device),
                              g['bake_ws'].to(device), align_corners=False, mean_neighbor=5, smooth_times=3)
    torch.testing.assert_close(field.cpu(), g['bake_field'], rtol=rtol, atol=1e-6)
    pose = torch.from_numpy(smpl_tmp_Apose(0)).float().view(1, 24, 3).to(device)
    sk, verts, faces = initialLBSkinner('female', g['bake_shape'].to(device), pose, (9, 13, 7), None, None,
                                        torch.tensor([[0.01, 0.02, -0.01]]), smpl=StandInSMPL())
    for name, got in (('ws', sk.ws), ('b_min', sk.b_min), ('b_max', sk.b_max), ('Js', sk.Js), ('init_pose', sk.init_pose),
                      ('extend', sk.bbox_extend), ('center', sk.bbox_center), ('verts', verts), ('faces', faces.float())):
        torch.testing.assert_close(got.detach().cpu().float().reshape(g['bake_sk_' + name].shape), g['bake_sk_' + name],
                                   rtol=rtol, atol=1e-6, msg=lambda m: name + ': ' + m)
    return sk.to(device)


def run_beta_fit(g, root, device, rtol=2e-3):
    """smpl_beta_optimizer on the joint capture under `root` against the reference function's betas / translation."""
    import random
    import capture_fixture as cf
    from recmv.dataset import SceneDataset
    from recmv.engineer.core.beta_optimizer import smpl_beta_optimizer
    torch.manual_seed(35)
    ds = SceneDataset(root, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=FL_SAMPLING, curve_sampling=1)
    for t in ds.conds + [ds.poses, ds.trans] + list(ds.camera_params.values()):
        t.data = t.data.to(device)
    random.seed(36)
    torch.manual_seed(36)
    betas, extra = smpl_beta_optimizer(ds.gender, None, ds, device, smpl=StandInSMPL(), log=None)
    worst = {}
    for name, got, want in (('betas', betas, g['beta_betas']), ('extra_trans', extra, g['beta_extra'])):
        worst[name] = float((got.cpu() - want).abs().max()) / float(want.abs().max())
        torch.testing.assert_close(got.cpu(), want, rtol=rtol, atol=rtol * float(want.abs().max()), msg=lambda m: name + ': ' + m)
    return worst
