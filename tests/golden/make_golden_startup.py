"""Golden vectors for the start-up stage that precedes the optimisation loop, from the REAL reference code:

  startup.npz   engineer/utils/matrix_transform.py  — 6-D rotation, per-line rigid / scale transforms (functions called
                directly on seeded inputs);
                dataset/dataset.py `get_init_fl_datasets` / `Init_Fl_SceneDataset` — which frames, their feature lines;
                engineer/core/fl_optimizer.py `scale_rigid_optimizer` (:111-519) and `rigid_optimizer` (:520-677) run
                for real on the synthetic capture of tests/capture_fixture.py with the reference's LBSkinner and the
                reference's dataset classes: the stored `init_trans_matrix.pth` and the registered line vertices, plus
                the path that re-applies a stored file;
                engineer/networks/OptimGarmentNetwork.py `initializeSDF` (:387-443) run for real on the reference's SDF
                net: parameters after three epochs (Adam + StepLR, the method's own shuffling and sampling).

What is stood in (none of it is the arithmetic under test): the functions hard-code `device='cuda:0'` / `.cuda()` — mapped
to the CPU; pytorch3d is absent — `Meshes` is a plain holder, `chamfer_distance` the restatement of recmv.curves (so the
chamfer arithmetic stays parity-unpinned, as in curves.npz), the camera recmv's restatement (pinned by camera_ndc.npz),
`mask_render` the C oracle rasteriser; OpenCV is absent — images are decoded by Pillow, the debug drawing calls are no-ops;
`np.int` / `np.bool` (removed from numpy 2) are the builtins.

    python tests/golden/make_golden_startup.py
"""
import random
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import PIL.Image  # noqa: E402,F401
PIL.Image.init()
import joblib  # noqa: E402,F401
import ref_loader  # noqa: E402
from recmv.dataset import read_image_bgr  # noqa: E402

ref_loader.install()
cv2 = types.ModuleType("cv2")
cv2.imread = lambda path, *a: read_image_bgr(path)
cv2.IMREAD_UNCHANGED = -1
cv2.circle = lambda img, *a, **k: img
cv2.imwrite = lambda *a, **k: True
sys.modules["cv2"] = cv2
import capture_fixture as cf  # noqa: E402
import common_setup as cs  # noqa: E402
import startup_case as sc  # noqa: E402
from make_golden import save  # noqa: E402


class Meshes:
    """The members of pytorch3d's Meshes the registration touches."""

    def __init__(self, verts, faces):
        self.verts, self.faces = list(verts), list(faces)

    def verts_packed(self):
        return torch.cat(self.verts, 0)

    def faces_packed(self):
        return self.faces[0]

    def update_padded(self, padded):
        return Meshes([padded[0]], self.faces)

    def to(self, device):
        return self


class TorchOnCpu:
    """`torch`, with the constructors the reference calls with device='cuda:0' building on the CPU."""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def full(*a, **k):
        k.pop('device', None)
        return torch.full(*a, **k)

    @staticmethod
    def tensor(*a, **k):
        k.pop('device', None)
        return torch.tensor(*a, **k)


def transforms():
    ref_loader.ref_module("model.network")       # the reference's own entry order (its packages import each other)
    MT = ref_loader.ref_module("engineer.utils.matrix_transform")
    g = torch.Generator().manual_seed(21)
    poses = torch.randn(5, 6, generator=g)
    poses[3, 3:] = poses[3, :3] * 2 + 1e-3 * torch.randn(3, generator=g)         # nearly parallel axes
    lines = [torch.randn(n, 3, generator=g) * 0.2 + torch.randn(1, 3, generator=g) for n in (7, 12, 5, 9, 30)]
    R = MT.compute_rotation_matrix_from_ortho6d(poses)
    T = 0.1 * torch.randn(5, 1, 3, generator=g)
    S = torch.tensor([1.5, 0.7, -0.3, 2.0, 1.0])                                   # a negative scale: clamped to 0
    cat = lambda lst: torch.cat(lst, 0)
    # closed 3-D polylines: fewer points than asked for (both outcomes of the closing test), more (farthest-point branch)
    PG = ref_loader.ref_module("engineer.utils.polygons")
    extra = {}
    for tag, n, newn, flip in (('few', 23, 60, 1.), ('few_flipped', 23, 60, -1.), ('many', 90, 40, 1.)):
        t = torch.sort(torch.rand(n, generator=g) * 2 * np.pi).values
        ring = torch.stack([flip * (0.3 + 0.05 * torch.sin(3 * t)) * torch.cos(t), 0.1 * torch.cos(2 * t),
                            (0.25 + 0.03 * torch.cos(5 * t)) * torch.sin(t)], -1) + torch.tensor([0.1, -0.2, 0.05])
        extra['us3d_%s_in' % tag] = ring
        extra['us3d_%s_out' % tag] = torch.from_numpy(np.ascontiguousarray(PG.uniformsample3d(ring.numpy(), newn))).float()
        extra['us3d_%s_n' % tag] = torch.tensor([float(newn)])
    cloud = torch.randn(2, 50, 3, generator=g)
    extra['fps_in'], extra['fps_out'] = cloud, PG.farthest_point_sample(cloud, 12).float()
    # Inverse_Fl_Body (model/Deformer.py:36-122) on the lines above, their scale-registered versions as input
    Dref = ref_loader.ref_module("model.Deformer")
    names = ['a', 'b', 'c', 'd', 'e']
    holders = [Meshes([l], [torch.zeros(1, 3, dtype=torch.long)]) for l in lines]
    inv = Dref.Inverse_Fl_Body(holders, names, T, S.abs() + 0.2)
    registered = MT.scale_icp_rotate_center_transform(lines, R, T, S.abs() + 0.2)
    inv.set_rigid_center([v.mean(0, keepdim=True) for v in registered], names)
    extra['inv_S'] = S.abs() + 0.2
    extra['inv_in'], extra['inv_out'] = cat(registered), cat(inv(registered, names))
    return dict(extra, mt_poses=poses, mt_lines=cat(lines), mt_split=torch.tensor([float(l.shape[0]) for l in lines]), mt_T=T, mt_S=S,
                mt_R=R, mt_icp=cat(MT.icp_rotate_transfrom(lines, R, T)),
                mt_scale_icp=cat(MT.scale_icp_rotate_transfrom(lines, R, T, S)),
                mt_center=cat(MT.center_transform(lines, R, T)),
                mt_icp_center=cat(MT.icp_rotate_center_transform(lines, R, T)),
                mt_scale_icp_center=cat(MT.scale_icp_rotate_center_transform(lines, R, T, S)))


def registration(out):
    ref_loader.ref_module("model.network")       # the reference's own entry order (its packages import each other)
    Fo = ref_loader.ref_module("engineer.core.fl_optimizer")
    MT = ref_loader.ref_module("engineer.utils.matrix_transform")
    Dref = ref_loader.ref_module("model.Deformer")
    refds = ref_loader.ref_module("dataset.dataset")
    from oracle import oracle as orc
    from recmv import curves as ours
    from recmv.model import RectifiedPerspectiveCameras as OurCameras
    Fo.torch = TorchOnCpu()
    Fo.tocuda = lambda x: x
    Fo.Meshes = lambda verts, faces: Meshes(verts, faces)
    Fo.RectifiedPerspectiveCameras = OurCameras
    Fo.chamfer_distance = lambda a, b, point_reduction='sum': (ours.chamfer_distance_sum(a, b), None)
    MT.pytorch3d = types.SimpleNamespace(structures=types.SimpleNamespace(Meshes=Meshes))
    np.int, np.bool = int, bool

    class MaskRender:
        rasterizer = types.SimpleNamespace(cameras=None)

        def __call__(self, meshes):
            cam = self.rasterizer.cameras
            verts = torch.stack([v.detach() for v in meshes.verts])
            faces = meshes.faces[0]
            N, F = verts.shape[0], faces.shape[0]
            W, H = int(cam.image_size[0, 0]), int(cam.image_size[0, 1])
            ndc = cam.transform_points_ndc(verts.reshape(-1, 3)).view(N, -1, 3)
            fv = ndc[:, faces.reshape(-1)].reshape(-1, 3, 3)
            p2f, zbuf, bary, dists = orc.rasterize_meshes(fv, torch.arange(N) * F, torch.full((N,), F), (H, W))
            return None, types.SimpleNamespace(zbuf=zbuf, pix_to_face=p2f)

    sk = cs.build_skinner(Dref.LBSkinner, Dref.batch_rodrigues)
    geo = sc.geometry(orc.mc)
    out.update({'reg_' + k: v for k, v in geo.items()})
    names = sc.LINE_NAMES
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with tempfile.TemporaryDirectory() as root:
            sc.write_capture(root)
            torch.manual_seed(31)
            ds = refds.SceneDataset(root, dict(sc.CONDS), cf.GARMENT_TYPE, fl_sampling=sc.FL_SAMPLING, curve_sampling=1)
            # Init_Fl loader contents (frames, flags, points)
            random.seed(32)
            torch.manual_seed(32)
            loader = ds.get_init_fl_datasets(3, None, 0)
            init_ds = loader.dataset
            out['init_idx'] = torch.tensor([float(i) for i in init_ds.idx])
            out['init_len'] = torch.tensor([float(len(init_ds)), float(len(loader))])
            for k in (0, 4, 7):
                fid, sample = init_ds[k]
                out['init_%d_fid' % k] = torch.tensor([float(fid)])
                out['init_%d_fl_pts' % k] = sample['fl_pts']
                out['init_%d_fl_masks' % k] = sample['fl_masks'].float()
                out['init_%d_mask' % k] = sample['mask']
            order = [int(f) for fids, _ in loader for f in fids]
            out['init_order'] = torch.tensor([float(f) for f in order])
            ps = refds.People_Snapshot_SceneDataset(root, dict(sc.CONDS), cf.GARMENT_TYPE, fl_sampling=sc.FL_SAMPLING,
                                                    curve_sampling=1, a_pose=False)
            ps_loader = ps.get_init_fl_datasets(2, None, 0)
            out['init_ps_idx'] = torch.tensor([float(i) for i in ps_loader.dataset.idx])
            out['init_ps_masks'] = torch.stack([ps_loader.dataset[k][1]['fl_masks'].float() for k in range(len(ps_loader.dataset))])

            def meshes():
                return {n: Meshes([geo['line_verts'][geo['line_split'][i]:geo['line_split'][i + 1]]],
                                  [geo['line_faces'][geo['face_split'][i]:geo['face_split'][i + 1]]])
                        for i, n in enumerate(names)}

            smpl_mesh = Meshes([geo['body_verts']], [geo['body_faces']])
            # ---- scale_rigid_optimizer
            save_path = str(Path(root) / 'fl_init')
            random.seed(33)
            torch.manual_seed(33)
            data_loader = torch.utils.data.DataLoader(ds, sc.BATCH, sampler=refds.RandomSampler(ds, 1, False), num_workers=0)
            got = Fo.scale_rigid_optimizer(sk, meshes(), smpl_mesh, MaskRender(), ds, data_loader, save_path, names, device='cpu')
            stored = torch.load(str(Path(save_path) / 'init_trans_matrix.pth'))
            out['srig_R'], out['srig_T'], out['srig_scale'] = stored['rigid_R'], stored['rigid_T'], stored['rigid_scale']
            out['srig_verts'] = torch.cat([m.verts_packed() for m in got], 0)
            again = Fo.scale_rigid_optimizer(sk, meshes(), smpl_mesh, MaskRender(), ds, data_loader, save_path, names, device='cpu')
            out['srig_reapplied'] = torch.cat([m.verts_packed() for m in again], 0)
            # ---- rigid_optimizer
            save_path = str(Path(root) / 'fl_init_rigid')
            random.seed(34)
            torch.manual_seed(34)
            train_loader = ds.get_init_fl_datasets(sc.BATCH, None, 0)
            got = Fo.rigid_optimizer(sk, meshes(), ds, train_loader, save_path, names, device='cpu')
            stored = torch.load(str(Path(save_path) / 'init_trans_matrix.pth'))
            out['rig_R'], out['rig_T'] = stored['rigid_R'], stored['rigid_T']
            out['rig_verts'] = torch.cat([m.verts_packed() for m in got], 0)
            again = Fo.rigid_optimizer(sk, meshes(), ds, train_loader, save_path, names, device='cpu')
            out['rig_reapplied'] = torch.cat([m.verts_packed() for m in again], 0)
    finally:
        torch.Tensor.cuda = real_cuda
    print("scale registration: scale", out['srig_scale'].tolist())
    print("                    |T| ", out['srig_T'].norm(dim=-1).view(-1).tolist())


def sdf_prefit(out):
    N = ref_loader.ref_module("model.network")
    OGN = ref_loader.ref_module("engineer.networks.OptimGarmentNetwork")
    for with_normals in (True, False):
        net = cs.build_sdf(N.getTmpSdf)
        vs, ns = sc.prefit_points()
        opt = torch.optim.Adam([{"params": net.parameters(), "lr": 0.005, "weight_decay": 0}])
        sche = torch.optim.lr_scheduler.StepLR(opt, 2, 0.5)
        with tempfile.TemporaryDirectory() as tmp:
            name = str(Path(tmp) / 'initial_sdf_idr_6_1.pth')
            torch.manual_seed(41)
            OGN.OptimGarmentNetwork.initializeSDF(types.SimpleNamespace(), net, opt, sche, sc.PREFIT_BATCH, sc.PREFIT_EPOCHS,
                                                  'cpu', vs, ns, with_normals, name)
            stored = torch.load(name)
        tag = 'prefit%d_' % int(with_normals)
        for k in sc.PREFIT_KEYS:
            out[tag + k.replace('.', '_')] = dict(net.named_parameters())[k].detach()[:sc.PREFIT_ROWS].clone()
            assert torch.equal(stored[k], dict(net.named_parameters())[k].detach())
        out[tag + 'lr'] = torch.tensor([opt.param_groups[0]['lr']])
        probe = sc.prefit_probe()
        with torch.no_grad():
            out[tag + 'probe'] = net(probe, -1)
    out['prefit_vs'], out['prefit_ns'] = sc.prefit_points()


def skinner_baking(out):
    """model/Deformer.py `smooth_weights`, `compute_lbswField`, `initialLBSkinner` (on tests/startup_case.StandInSMPL),
    utils/utils.py `smpl_tmp_Apose`."""
    ref_loader.ref_module("model.network")
    Dref = ref_loader.ref_module("model.Deformer")
    U = ref_loader.ref_module("utils.utils")
    g = torch.Generator().manual_seed(71)
    w = torch.softmax(torch.randn(1, 24, 6, 7, 5, generator=g), dim=1)
    out['bake_smooth_in'], out['bake_smooth_out'] = w.clone(), Dref.smooth_weights(w.clone(), 4)
    verts = torch.randn(80, 3, generator=g) * torch.tensor([0.3, 0.5, 0.2])
    ws = torch.softmax(2 * torch.randn(80, 24, generator=g), dim=1)
    out['bake_verts'], out['bake_ws'] = verts, ws
    out['bake_field'] = Dref.compute_lbswField([-0.5, -0.8, -0.3], torch.tensor([0.5, 0.8, 0.3]), (7, 9, 5), verts, ws,
                                               align_corners=False, mean_neighbor=5, smooth_times=3)
    for t in range(4):
        out['apose_%d' % t] = torch.from_numpy(U.smpl_tmp_Apose(t))
    Dref.getSMPL = lambda gender: sc.StandInSMPL(rodrigues=Dref.batch_rodrigues)
    shape = 0.5 * torch.randn(10, generator=g)
    pose = torch.from_numpy(U.smpl_tmp_Apose(0)).view(1, 24, 3)
    sk, v, f = Dref.initialLBSkinner('female', shape, pose, (9, 13, 7), None, None, torch.tensor([[0.01, 0.02, -0.01]]))
    out['bake_shape'] = shape
    for name, t in (('ws', sk.ws), ('b_min', sk.b_min), ('b_max', sk.b_max), ('Js', sk.Js), ('init_pose', sk.init_pose),
                    ('extend', sk.bbox_extend), ('center', sk.bbox_center), ('verts', v), ('faces', f.float())):
        out['bake_sk_' + name] = t.detach().float()
    # the baked skinner at work: posed points (reference forward), for the GPU test of the fused kernels on a single extent
    pts = 0.3 * torch.randn(2, 60, 3, generator=g)
    poses, trans = cs.poses_trans(2, seed=9)
    out['bake_pts'], out['bake_poses'], out['bake_trans'] = pts, poses, trans
    out['bake_posed'] = sk(pts, [poses, trans])


def beta_fit(out):
    """engineer/core/beta_optimizer.py `smpl_beta_optimizer` (:132-245) run for real: 150 Adam steps on the 2-D joints of the
    capture, SMPL replaced by the stand-in on both sides."""
    import os
    ref_loader.ref_module("model.network")
    Bo = ref_loader.ref_module("engineer.core.beta_optimizer")
    Dref = ref_loader.ref_module("model.Deformer")
    refds = ref_loader.ref_module("dataset.dataset")
    from recmv.model import RectifiedPerspectiveCameras as OurCameras
    Bo.torch = TorchOnCpu()
    Bo.RectifiedPerspectiveCameras = OurCameras
    Bo.getSMPL = lambda gender: sc.StandInSMPL(rodrigues=Dref.batch_rodrigues)
    np.int, np.bool = int, bool
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as root, tempfile.TemporaryDirectory() as scratch:
        sc.write_joint_capture(root)
        torch.manual_seed(35)
        ds = refds.SceneDataset(root, dict(sc.CONDS), cf.GARMENT_TYPE, fl_sampling=sc.FL_SAMPLING, curve_sampling=1)
        os.chdir(scratch)                                    # the function writes ./debug/smpl_beta/
        try:
            random.seed(36)
            torch.manual_seed(36)
            betas, extra = Bo.smpl_beta_optimizer(ds.gender, None, ds, 'cpu')
        finally:
            os.chdir(cwd)
        out['beta_start'] = ds.shape.clone()
    out['beta_betas'], out['beta_extra'] = betas, extra
    print("beta fit: |betas - start| max", float((betas - out['beta_start']).abs().max()), " extra_trans", extra.view(-1).tolist())


def main():
    torch.set_num_threads(8)
    out = {}
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self               # normalize_vector builds its floor with .cuda()
    try:
        out.update(transforms())
    finally:
        torch.Tensor.cuda = real_cuda
    registration(out)
    sdf_prefit(out)
    skinner_baking(out)
    beta_fit(out)
    save("startup", **out)


if __name__ == "__main__":
    main()
