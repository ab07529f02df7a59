"""Golden vectors for the feature-curve branch, from the REAL reference classes imported from /root/reference:

  curves.npz   engineer/utils/garment_structure.py `Intersect_Free_Curve` (initialize_parameters / forward /
               regularization on seeded closed curves; constructed without its mesh-extraction front end) and
               engineer/core/fl_optimizer.py `fl_proj_loss` (its loop / normalisation logic; pytorch3d's
               chamfer_distance — absent here — is replaced by the restatement in recmv.curves, so the chamfer
               arithmetic itself stays parity-unpinned)

    python tests/golden/make_golden_curves.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import ref_loader  # noqa: E402

ref_loader.install()
import common_setup as cs  # noqa: E402
from make_golden import save  # noqa: E402


def rings(seed, n_lines=3, n=40):
    g = torch.Generator().manual_seed(seed)
    out = []
    for k in range(n_lines):
        t = torch.linspace(0, 2 * np.pi, n + 1)[:-1]
        r = 0.3 + 0.1 * k + 0.02 * torch.randn(n, generator=g)
        c = torch.tensor([0.05 * k, 0.4 - 0.3 * k, 0.02])
        tilt = 0.2 * k
        p = torch.stack([r * torch.cos(t), 0.05 * torch.sin(3 * t) + tilt * r * torch.cos(t), r * torch.sin(t)], -1)
        out.append((p + c).float())
    return out


def main():
    ref_loader.ref_module("model.network")       # the reference's own entry order (its packages import each other)
    G = ref_loader.ref_module("engineer.utils.garment_structure")
    Fo = ref_loader.ref_module("engineer.core.fl_optimizer")
    from recmv import curves as ours
    names = ['neck', 'left_cuff', 'upper_bottom']
    curves = rings(3)
    smpl = [0.93 * c for c in curves]
    ref = object.__new__(G.Intersect_Free_Curve)
    torch.nn.Module.__init__(ref)
    ref.cano2canosmpl = lambda lst, nm: [0.93 * c for c in lst]
    ref.fl_names = names
    ref.sample_num = 40
    ref.initialize_parameters([c.clone() for c in curves])
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        ref.scale.copy_(1.0 + 0.3 * torch.randn(ref.scale.shape, generator=g))      # some scales negative -> ReLU path
        ref.scale[0, :5] = -0.2
        ref.nx_scale.copy_(0.05 * torch.randn(ref.nx_scale.shape, generator=g))
    verts = ref()
    fl_masks = torch.tensor([[1., 1., 0.], [1., 0., 1.]])
    reg = ref.regularization(fl_masks)
    (reg['diff_a_loss'] + verts.sum()).backward()
    q = ref.query_canosmpl_verts(['upper_bottom', 'neck'])
    # 4 lines: torch.cross without `dim` (garment_structure.py:89) then acts on the last axis; with exactly 3 lines
    # (above) its legacy default picks the FIRST axis of size 3, i.e. the line axis — both behaviours are pinned
    curves4 = rings(4, n_lines=4, n=24)
    ref4 = object.__new__(G.Intersect_Free_Curve)
    torch.nn.Module.__init__(ref4)
    ref4.cano2canosmpl = lambda lst, nm: [0.9 * c for c in lst]
    ref4.fl_names = ['neck', 'left_cuff', 'right_cuff', 'upper_bottom']
    ref4.sample_num = 24
    ref4.initialize_parameters([c.clone() for c in curves4])
    # ---- fl_proj_loss (reference loop, restated chamfer)
    Fo.chamfer_distance = lambda a, b, point_reduction='sum': (ours.chamfer_distance_sum(a, b), None)
    N, S, M = 2, 40, 25
    pts = [torch.rand(N, S, 3, generator=g) * 100 for _ in range(3)]
    gts = [torch.rand(N, M, 2, generator=g) * 100 for _ in range(3)]
    masks = []
    for k in range(3):
        m = torch.rand(N, S, generator=g) > 0.4
        if k == 1:
            m[1] = False                                  # a frame that does not see the line
        if k == 2:
            m[:] = False                                  # a line nobody sees
        masks.append(m[..., None].expand(N, S, 3).clone())
    w = [1.0, 2.5, 0.7]
    loss = Fo.fl_proj_loss(pts, gts, masks, w)
    save("curves", curves=torch.stack(curves), smpl=torch.stack(smpl), scale=ref.scale,
         nx_scale=ref.nx_scale, verts=verts, center=ref.cano_verts_center, nx=ref.cano_nx, dirs=ref.cano_v_dirs,
         init_scale=ref.init_scale, fl_masks=fl_masks, reg_diff=reg['diff_a_loss'], reg_center=reg['center_offset'],
         g_scale=ref.scale.grad, g_nx=ref.nx_scale.grad, q0=q[0], q1=q[1],
         curves4=torch.stack(curves4), nx4=ref4.cano_nx, verts4=ref4(),
         proj_pts=torch.stack(pts), proj_gts=torch.stack(gts), proj_masks=torch.stack(masks), proj_w=np.array(w),
         proj_loss=loss)




def proj_loss_fixture():
    """OptimGarmentNetwork.compute_fl_proj_loss (:1605-1711) for real on a stand-in self: screen projection, z-buffer
    thresholds per feature line, label masks, fl_proj_loss (restated chamfer), curve regulariser with config weights."""
    import types
    ref_loader.ref_module("model.network")
    G = ref_loader.ref_module("engineer.utils.garment_structure")
    OGN = ref_loader.ref_module("engineer.networks.OptimGarmentNetwork")
    from recmv import curves as ours
    from recmv.hocon import ConfigFactory
    from recmv.model import RectifiedPerspectiveCameras as OurCameras
    OGN.fl_proj_loss.__globals__["chamfer_distance"] = lambda a, b, point_reduction='sum': (ours.chamfer_distance_sum(a, b), None)
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf")).get_config('loss_coarse')
    names = ['neck', 'left_cuff', 'right_cuff', 'upper_bottom']          # FL_EXTRACT['short_sleeve_upper']
    cur = rings(8, n_lines=4, n=30)
    ref = object.__new__(G.Intersect_Free_Curve)
    torch.nn.Module.__init__(ref)
    ref.cano2canosmpl = lambda lst, nm: [0.9 * c for c in lst]
    ref.fl_names, ref.sample_num = names, 30
    ref.initialize_parameters([c.clone() for c in cur])
    g = torch.Generator().manual_seed(61)
    N, S, M = 3, 30, 20
    cam = OurCameras(torch.tensor([[300., 295.]]), torch.tensor([[64., 60.]]),
                     torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3), torch.tensor([[0.05, -0.1, 2.5]]),
                     image_size=[(128, 120)])
    base = ref()                                                                                  # [4,S,3]
    defs = [(base[i][None] + 0.03 * torch.randn(N, S, 3, generator=g)).detach().requires_grad_(True) for i in range(4)]
    checks = [torch.rand(N, S, 2, generator=g) * 0.2 - 0.05 for _ in range(4)]                      # around the thresholds
    fl_masks = torch.tensor([[1., 1., 0., 1.], [1., 1., 1., 1.], [0., 1., 1., 0.]])
    gt = torch.rand(N, 4 * M, 2, generator=g) * 100 + 10
    fake = types.SimpleNamespace(conf=conf, info={'fl_loss': {}}, inter_free_curve=ref,
                                 dataset=types.SimpleNamespace(fl_weights={'neck': 1.0, 'left_cuff': 2.0,
                                                                           'right_cuff': 0.5, 'upper_bottom': 1.5}))
    proj_size = torch.tensor([[128., 120.]]).repeat(N, 1)
    loss, vis, lab = OGN.OptimGarmentNetwork.compute_fl_proj_loss(
        fake, defs, checks, None, fl_masks, gt, None, 'short_sleeve_upper', None, None, [S] * 4, 0, cam, proj_size)
    grads = torch.autograd.grad(loss, defs + [ref.scale, ref.nx_scale])
    save("curve_proj", curves=torch.stack(cur), defs=torch.stack([d.detach() for d in defs]), checks=torch.stack(checks),
         fl_masks=fl_masks, gt=gt, loss=loss, g_defs=torch.stack(grads[:4]), g_scale=grads[4], g_nx=grads[5],
         vis=torch.stack([v[..., 0] for v in vis]))




def visibility_fixture():
    """OptimGarmentNetwork.fl_visible_by_body_zbuff (:1374-1448) for real on a stand-in self: reference deformer and
    skinner, reference depth logic (background fill, uv mapping, bilinear z-buffer read, sign conventions).  The
    rasteriser behind `maskRender` is the C oracle (pytorch3d is absent) and `Meshes` a plain holder."""
    import types
    ref_loader.ref_module("model.network")
    Dref = ref_loader.ref_module("model.Deformer")
    OGN = ref_loader.ref_module("engineer.networks.OptimGarmentNetwork")
    from oracle import oracle as orc
    from recmv.model import RectifiedPerspectiveCameras as OurCameras

    class Meshes:
        def __init__(self, verts, faces):
            self.verts, self.faces = verts, faces

    OGN.Meshes = Meshes
    H, W, N = 60, 48, 3
    cam = OurCameras(torch.tensor([[70., 68.]]), torch.tensor([[24., 30.]]),
                     torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3), torch.tensor([[0.02, -0.05, 2.4]]),
                     image_size=[(W, H)])

    def mask_render(meshes):
        verts = torch.stack([v.detach() for v in meshes.verts])                      # [N,V,3]
        faces = meshes.faces[0]
        ndc = cam.transform_points_ndc(verts.reshape(-1, 3)).view(N, -1, 3)
        F = faces.shape[0]
        fv = ndc[:, faces.reshape(-1)].reshape(-1, 3, 3)
        p2f, zbuf, bary, dists = orc.rasterize_meshes(fv, torch.arange(N) * F, torch.full((N,), F), (H, W))
        return None, types.SimpleNamespace(zbuf=zbuf, pix_to_face=p2f)

    tr = cs.build_translator(Dref.MLPTranslator)
    sk = cs.build_skinner(Dref.LBSkinner, Dref.batch_rodrigues)
    comp = Dref.CompositeDeformer([tr, sk])

    def sphere(res, radius):
        ax = torch.linspace(-0.6, 0.6, res)
        x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
        step = 1.2 / (res - 1)
        return orc.mc(((x * x + y * y + z * z).sqrt() - radius).contiguous(), step, step, step, -0.6, -0.6, -0.6, 0.0)

    gv, gf = sphere(21, 0.42)
    bv, bf = sphere(17, 0.36)
    conds, _ = cs.conds_and_inds(8, nframes=N, condlen=128, seed=4)
    poses, trans = cs.poses_trans(N, seed=7)
    conds, poses, trans = conds.detach(), poses.detach(), trans.detach()
    t = torch.linspace(0, 2 * 3.14159265, 41)[:-1]
    ring = lambda y, r: torch.stack([r * torch.cos(t), torch.full_like(t, y), r * torch.sin(t)], -1)
    curves = [ring(0.25, 0.33), ring(-0.2, 0.37)]
    smpl_curves = [c * (0.36 / 0.42) for c in curves]
    ratio = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}
    with torch.no_grad():
        def_fl = [comp(c.view(-1, 3).expand(N, -1, 3), [conds, [poses, trans]], ratio=ratio, offset_type=n)
                  for c, n in zip(curves, ("neck", "upper_bottom"))]
    fake = types.SimpleNamespace(deformer=comp, garment_vs=[gv], garment_fs=[gf], tmpBodyVs=bv, tmpBodyFs=bf,
                                 maskRender=mask_render)
    fake.maskRender = types.SimpleNamespace(__call__=None)
    class MR:
        rasterizer = types.SimpleNamespace(cameras=cam)
        def __call__(self, meshes):
            return mask_render(meshes)
    fake.maskRender = MR()
    proj_size = torch.tensor([[float(W), float(H)]]).repeat(N, 1)
    with torch.no_grad():
        out = OGN.OptimGarmentNetwork.fl_visible_by_body_zbuff(
            fake, cam, [None, conds], [poses, trans], ratio, None, [d.clone() for d in def_fl],
            [c.view(1, -1, 3) for c in smpl_curves], ["neck", "upper_bottom"], 0, "upper", proj_size, N)
    print("visibility checks: garment %.3f..%.3f  body %.3f..%.3f" % (out[..., 0].min(), out[..., 0].max(),
                                                                     out[..., 1].min(), out[..., 1].max()))
    save("curve_vis", gv=gv, gf=gf, bv=bv, bf=bf, conds=conds, poses=poses, trans=trans, curves=torch.stack(curves),
         smpl=torch.stack(smpl_curves), def_fl=torch.stack(def_fl), checks=out, H=H, W=W)


if __name__ == "__main__":
    main()
    proj_loss_fixture()
    visibility_fixture()
