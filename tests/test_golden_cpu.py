"""CPU-side parity of the host logic against the golden vectors generated from the REAL reference code
(tests/golden/make_golden.py): constructors (parameter-for-parameter initialisation), annealing weights,
positional encoding (torch path), cameras, Seg3dLossless (torch route) and the hot loop on the CPU port."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLD))
import common_setup as cs  # noqa: E402


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(GOLD / f"{name}.npz").items()}


def test_constructors_reproduce_reference_parameters():
    from recmv.model import MLPTranslator, RenderingNetwork_view_norm, getTmpSdf
    for name, net in (("sdf", cs.build_sdf(getTmpSdf)), ("translator", cs.build_translator(MLPTranslator)),
                      ("render", cs.build_render(RenderingNetwork_view_norm))):
        np.testing.assert_allclose(cs.fingerprint(net), load(name)["fingerprint"].numpy(), rtol=1e-6, atol=1e-5)
    sdf = cs.build_sdf(getTmpSdf)
    keys = list(sdf.state_dict().keys())
    assert "lin0.weight_g" in keys and "lin0.weight_v" in keys and "lin8.bias" in keys     # checkpoint names
    assert sdf.lin3.weight_v.shape == (473, 512) and sdf.lin8.weight_v.shape == (257, 512)


def test_annealing_weights_and_embedder_torch_path():
    from recmv.model import get_embedder
    from recmv.utils import annealing_weights
    g = load("embedder")
    for r, row in zip(g["ratios"].tolist(), g["ws_grid"]):
        if r > 0:
            np.testing.assert_allclose(annealing_weights(6, r), row.numpy(), rtol=0, atol=0)
    embed, dim = get_embedder(6)
    assert dim == int(g["out_dim"])
    assert torch.equal(embed(g["x"]), g["plain"])
    torch.testing.assert_close(embed(g["x"], g["ws"].tolist()), g["weighted"], rtol=0, atol=0)
    assert torch.equal(get_embedder(4)[0](g["x"]), g["embed4"])


def test_lbs_init_pose_and_skeleton_cpu():
    """init_pose_inverse + kinematic chain (python route) vs the reference LBSkinner."""
    from recmv.model import LBSkinner
    g = load("lbs")
    sk = cs.build_skinner(LBSkinner)
    torch.testing.assert_close(sk.init_pose, g["init_pose"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(sk.posedSkeleton([g["poses"], g["trans"]]), g["skel"], rtol=1e-5, atol=1e-6)
    assert sk.ws.stride(1) == 1, "skinning volume is stored channels-last for the HIP sampler"


def test_camera_formulas():
    from recmv.model import RectifiedPerspectiveCameras
    cam = RectifiedPerspectiveCameras(torch.tensor([[1000., 990.]]), torch.tensor([[256., 250.]]),
                                      torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3),
                                      torch.tensor([[0.1, -0.2, 3.0]]), image_size=[(512, 512)])
    pix = torch.tensor([[10., 20., 1.], [300., 400., 1.]])
    rays = cam.view_rays(pix)
    torch.testing.assert_close(rays.norm(dim=1), torch.ones(2))
    c = cam.cam_pos()
    pts = c.view(1, 3) + 2.5 * rays                       # a point on each ray must project back to its pixel
    torch.testing.assert_close(cam.project(pts), pix[:, :2], rtol=1e-4, atol=1e-3)
    assert 0.02 < cam.angThreshold(0.5) < 0.03            # ~half a pixel at f=1000: 0.0286 deg at the centre


def test_seg3d_lossless_torch_route_matches_reference_on_analytic_field():
    """Seg3dLossless restatement (use_cuda_impl=False) vs the lossless property: equals dense evaluation near the
    surface, so MC of the sparse result == MC of the dense result."""
    from oracle import oracle as orc
    from recmv.MCAcc import Seg3dLossless, create_grid3D

    def field(p):
        return (p.norm(dim=-1) - 0.6 + 0.05 * torch.sin(7 * p[..., 0]) * torch.cos(5 * p[..., 1]))

    calls = []

    def query(points):
        calls.append(points.shape[1])
        return field(points.reshape(-1, 3)).reshape(1, 1, -1)

    res = [(9, 11, 7), (17, 21, 13), (33, 41, 25), (65, 81, 49)]
    eng = Seg3dLossless(query_func=query, b_min=[-1.0, -1.1, -0.9], b_max=[1.0, 1.1, 0.9], resolutions=res,
                        align_corners=False, balance_value=0.0, use_cuda_impl=False, faster=False)
    grid = eng.forward()
    W, H, D = res[-1]
    assert grid.shape == (1, 1, D, H, W)
    coords = create_grid3D(0, (W - 1, H - 1, D - 1), (W, H, D), device="cpu")
    dense = eng.batch_eval(coords.unsqueeze(0)).view(1, 1, D, H, W)
    assert sum(calls[:-1]) < 0.35 * W * H * D, "only a shell around the surface is queried"
    assert torch.equal(grid < 0, dense < 0), "lossless: inside/outside identical to dense evaluation"
    v1, f1 = orc.mc(grid[0, 0].permute(2, 1, 0).contiguous(), eng.spacing_x, eng.spacing_y, eng.spacing_z, eng.bx,
                    eng.by, eng.bz, 0.0)
    v2, f2 = orc.mc(dense[0, 0].permute(2, 1, 0).contiguous(), eng.spacing_x, eng.spacing_y, eng.spacing_z, eng.bx,
                    eng.by, eng.bz, 0.0)
    assert torch.equal(f1, f2) and torch.equal(v1, v2)


def test_golden_seg3d_mesh_is_closed_sphere_like():
    g = load("seg3d")
    from test_mc_oracle import mesh_invariants
    assert mesh_invariants(g["verts"], g["faces"]) == 2


class _Frags:
    def __init__(self, p2f, bary):
        self.pix_to_face, self.bary_coords = p2f, bary


def test_find_surface_ps_matches_the_reference_function():
    """utils/FindSurfacePs.py:7-37 run for real (tests/golden/make_golden_raster.py): same pixels, same faces (bit-exact
    indices — the "surface-point indices" of the north star), same canonical points; K = 1 shortcut and K = 3 path."""
    from recmv.utils import FindSurfacePs
    g = load("findsurface")
    for suf, p2f, bary in (("", g["pix_to_face"], g["bary"]), ("3", g["pix_to_face3"], g["bary3"])):
        b, r, c, p0, f = FindSurfacePs(g["verts"], g["faces"], _Frags(p2f, bary))
        for got, key in ((b, "batch"), (r, "row"), (c, "col"), (f, "finds")):
            assert torch.equal(got, g[key + suf]), key + suf
        torch.testing.assert_close(p0, g["init" + suf], rtol=0, atol=1e-7)
    assert g["batch"].numel() > 1000 and g["batch3"].numel() > 100


def test_oracle_rasteriser_reproduces_the_fixture_fragments(oracle):
    g = load("findsurface")
    F = g["faces"].shape[0]
    p2f, zbuf, bary, dists = oracle.rasterize_meshes(g["fv"], torch.tensor([0, F]), torch.tensor([F, F]),
                                                     (int(g["H"]), int(g["W"])))
    assert torch.equal(p2f, g["pix_to_face"])
    assert torch.equal(bary.view(torch.int32), g["bary"].view(torch.int32))


def test_camera_ndc_matches_the_reference_calibration_matrix():
    """transform_points_ndc vs the projection through the reference's own `_get_sfm_calibration_matrix`
    (model/CameraMine.py:210-300) and MeshRasterizer.transform's view-space depth."""
    from recmv.model import RectifiedPerspectiveCameras
    g = load("camera_ndc")
    cam = RectifiedPerspectiveCameras(g["focal"], g["pp"], g["R"], g["T"], image_size=[(int(g["W"]), int(g["H"]))])
    out = cam.transform_points_ndc(g["pts"])
    torch.testing.assert_close(out[:, :2], g["ndc_xy"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out[:, 2], g["view_z"], rtol=0, atol=1e-6)
    # and the pixel convention that ties it to the rays: NDC x of pixel column c is 1 - (2c+1)/W
    W, H = int(g["W"]), int(g["H"])
    pix = cam.project(g["pts"])
    torch.testing.assert_close(1 - (2 * pix[:, 0] + 1) / W, out[:, 0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(1 - (2 * pix[:, 1] + 1) / H, out[:, 1], rtol=1e-4, atol=1e-5)


def test_intersect_free_curve_matches_the_reference_class():
    """engineer/utils/garment_structure.py:36-147 run for real (tests/golden/make_golden_curves.py)."""
    from recmv.curves import Intersect_Free_Curve
    g = load("curves")
    names = ['neck', 'left_cuff', 'upper_bottom']
    c = Intersect_Free_Curve(list(g["curves"]), list(g["smpl"]), names)
    for got, key in ((c.cano_verts_center, "center"), (c.cano_nx, "nx"), (c.cano_v_dirs, "dirs"),
                     (c.init_scale, "init_scale")):
        torch.testing.assert_close(got, g[key], rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        c.scale.copy_(g["scale"])
        c.nx_scale.copy_(g["nx_scale"])
    verts = c()
    torch.testing.assert_close(verts, g["verts"], rtol=1e-6, atol=1e-7)
    reg = c.regularization(g["fl_masks"])
    torch.testing.assert_close(reg["diff_a_loss"], g["reg_diff"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(reg["center_offset"], g["reg_center"], rtol=0, atol=0)
    (reg["diff_a_loss"] + verts.sum()).backward()
    torch.testing.assert_close(c.scale.grad, g["g_scale"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(c.nx_scale.grad, g["g_nx"], rtol=1e-4, atol=1e-6)
    q = c.query_canosmpl_verts(['upper_bottom', 'neck'])
    assert torch.equal(q[0], g["q0"]) and torch.equal(q[1], g["q1"])
    c4 = Intersect_Free_Curve(list(g["curves4"]), list(0.9 * g["curves4"]),
                              ['neck', 'left_cuff', 'right_cuff', 'upper_bottom'])
    torch.testing.assert_close(c4.cano_nx, g["nx4"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(c4(), g["verts4"], rtol=1e-6, atol=1e-7)


def test_fl_proj_loss_matches_the_reference_function():
    """engineer/core/fl_optimizer.py:72-110: visible-sample selection, per-frame chamfer, normalisation by the frames
    that see a line and by its visible samples — incl. a frame without the line and a line nobody sees."""
    from recmv.curves import fl_proj_loss
    g = load("curves")
    loss = fl_proj_loss(list(g["proj_pts"]), list(g["proj_gts"]), list(g["proj_masks"]), g["proj_w"].tolist())
    torch.testing.assert_close(loss, g["proj_loss"], rtol=1e-5, atol=1e-6)


def test_propagate_tmp_ps_grad_matches_the_reference_method():
    """OptimGarmentNetwork.propagateTmpPsGrad (:2159-2313) run for real (tests/golden/make_golden_propagate.py) vs
    HotLoop.propagateTmpPsGrad on the CPU port: the gradients injected into the SDF net, the offset MLP, the per-frame
    codes, the poses / translations and the camera (focal, principal point, T) agree."""
    from oracle import cpu_port
    import propagate_case as pc
    from recmv.model import CompositeDeformer, LBSkinner, MLPTranslator, getTmpSdf
    g = load("propagate")
    cpu_port.install()
    try:
        sdf, tr = cs.build_sdf(getTmpSdf), cs.build_translator(MLPTranslator)
        comp = CompositeDeformer([tr, cs.build_skinner(LBSkinner)])
        out, n_total, n_ok = pc.run(g, sdf, tr, comp, "cpu")
    finally:
        cpu_port.uninstall()
    assert (n_total, n_ok) == (int(g["inv_total"]), int(g["inv_ok"]))
    pc.compare(out, g, rtol=2e-3, atol_rel=2e-4)


def test_dct_pose_loss_and_small_host_pieces_match_the_reference():
    """tests/golden/make_golden_misc.py: DCTNullSpace (utils/utils.py:293-305), the frame windows of
    get_batchframe_data (dataset/dataset.py:438-457), dct_poses_loss (:1221-1250) with its gradients, GMRobustError."""
    import types
    from recmv.loop import HotLoop, SyntheticFrames, dct_nullspace
    from recmv.model import LBSkinner
    from recmv.utils import GMRobustError
    g = load("misc")
    torch.testing.assert_close(dct_nullspace(30, 10), g["dctnull"], rtol=0, atol=2e-6)
    poses, trans = g["poses"].clone().requires_grad_(True), g["trans"].clone().requires_grad_(True)
    ds = types.SimpleNamespace(F=40, poses=poses, trans=trans)
    ds.get_batchframe_data = lambda name, fids, nlen: SyntheticFrames.get_batchframe_data(ds, name, fids, nlen)
    win, idx = ds.get_batchframe_data('poses', g["frame_ids"], 30)
    assert torch.equal(win.detach(), g["window"])
    assert torch.equal(g["frame_ids"] - idx[:, 0], g["rel"])            # the reference returns fids - starts
    fake = types.SimpleNamespace(dctnull=g["dctnull"], dataset=ds, info={},
                                 deformer=types.SimpleNamespace(defs=[None, cs.build_skinner(LBSkinner)]),
                                 conf=types.SimpleNamespace(get_float=lambda k: 2.0))
    fid = g["frame_ids"]
    loss = HotLoop.dct_poses_loss(fake, poses[fid], trans[fid], fid, 3)
    torch.testing.assert_close(loss, g["dct_loss"], rtol=1e-5, atol=1e-7)
    gp, gt = torch.autograd.grad(loss, [poses, trans], allow_unused=True)
    torch.testing.assert_close(gp, g["g_poses"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(torch.zeros_like(trans) if gt is None else gt, g["g_trans"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(GMRobustError(g["gm_x"], 0.01, True), g["gm_true"], rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(GMRobustError(g["gm_x"], 0.5, False), g["gm_false"], rtol=1e-6, atol=1e-8)


def test_compute_garment_pc_loss_matches_the_reference_method():
    """OptimGarmentNetwork.compute_garment_pc_loss (:621-667) run for real: silhouette IoU + LBS-consistency term and
    their gradients w.r.t. the silhouette and the explicit vertices."""
    import types
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    from recmv.model import LBSkinner
    g = load("misc")
    conf = ConfigFactory.parse_file(str(GOLD.parent.parent / "configs" / "synthetic" / "people_snapshot_like.conf"))
    sk = cs.build_skinner(LBSkinner)
    fake = types.SimpleNamespace(conf=conf.get_config('loss_coarse'), info={},
                                 deformer=types.SimpleNamespace(defs=[None, sk]))
    imgs, verts = g["pc_imgs"].clone().requires_grad_(True), g["pc_verts"].clone().requires_grad_(True)
    N = imgs.shape[0]
    from oracle import cpu_port
    cpu_port.install()              # the skinner's weight sampler runs on the C oracle here
    try:
        _pc_loss_body(g, fake, sk, imgs, verts, N)
    finally:
        cpu_port.uninstall()


def _pc_loss_body(g, fake, sk, imgs, verts, N):
    from recmv.loop import HotLoop
    # the fixture's deformed vertices = skinning + a fixed perturbation; rebuild them on our skinner so that the
    # consistency term differentiates through the same expression
    base = sk(verts.view(1, -1, 3).expand(N, -1, 3), [g["pc_poses"], g["pc_trans"]])
    with torch.no_grad():
        ref_base = sk(g["pc_verts"].view(1, -1, 3).expand(N, -1, 3), [g["pc_poses"], g["pc_trans"]])
        noise = g["pc_def"] - ref_base
    loss = HotLoop.compute_garment_pc_loss(fake, base + noise, [None, [g["pc_poses"], g["pc_trans"]]], imgs,
                                           g["pc_gt"], 'upper', verts)
    torch.testing.assert_close(loss, g["pc_loss"], rtol=1e-5, atol=1e-6)
    g_img, g_v = torch.autograd.grad(loss, [imgs, verts], allow_unused=True)
    torch.testing.assert_close(g_img, g["pc_g_img"], rtol=1e-4, atol=1e-8)
    # d(def - skin)/d verts cancels exactly in both implementations
    torch.testing.assert_close(torch.zeros_like(verts) if g_v is None else g_v, g["pc_g_verts"], rtol=1e-4, atol=1e-6)


def test_surface_render_loss_matches_the_reference_method():
    """OptimGarmentNetwork.surface_render_loss (:1083-1219) run for real (tests/golden/make_golden_render_loss.py) vs
    HotLoop.surface_render_loss on the CPU port with the same seed: value, per-term info, and the gradients that
    `backward()` leaves on the SDF net, the offset MLP, the colour net, the per-frame codes, the poses and the surface
    points.  (Differences by construction: closed-form singular values instead of the CPU SVD, jet pass instead of
    double backward.)"""
    import types
    from oracle import cpu_port
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    from recmv.model import (CompositeDeformer, LBSkinner, MLPTranslator, RenderingNetwork_view_norm, getTmpSdf)
    g = load("render_loss")
    conf = ConfigFactory.parse_file(str(GOLD.parent.parent / "configs" / "synthetic" / "people_snapshot_like.conf"))
    cpu_port.install()
    try:
        sdf, tr = cs.build_sdf(getTmpSdf), cs.build_translator(MLPTranslator)
        comp = CompositeDeformer([tr, cs.build_skinner(LBSkinner)])
        rn = cs.build_render(RenderingNetwork_view_norm)
        leaf = lambda t: t.detach().clone().requires_grad_(True)
        leaves = dict(conds=leaf(g["conds"]), poses=leaf(g["poses"]), trans=leaf(g["trans"]),
                      rendcond=leaf(g["in_rendcond"]))
        fake = types.SimpleNamespace(conf=conf.get_config('loss_coarse'), device='cpu', garment_size=1,
                                     garment_names=['upper'], garment_nets=[sdf], deformer=comp, netRender=rn, info={})
        fake.garment_vs = [g["in_verts"].clone().requires_grad_(True)]
        fake.dataset = types.SimpleNamespace(images=lambda fids: (g["in_gtC"], g["in_gtN"]))
        fake.get_grad_parameters = lambda fids, dev: ([None, leaves["conds"]], leaves["poses"], leaves["trans"],
                                                      leaves["rendcond"])
        fake._ray_valid = [g["in_check"].sum()]
        cameras = types.SimpleNamespace(R=g["in_R"])
        samples = [(g["in_binds"], g["in_row"], g["in_col"], None, g["in_rays"])]
        torch.manual_seed(int(g["seed"]))
        loss = HotLoop.surface_render_loss(fake, 3, cameras, torch.arange(3), {"sdfRatio": 0.8, "deformerRatio": 0.7,
                                                                               "renderRatio": 1.0},
                                           [g["in_check"]], [g["in_init"].clone()], samples)
        loss.backward()
    finally:
        cpu_port.uninstall()
    torch.testing.assert_close(loss.detach(), g["loss"], rtol=2e-4, atol=1e-5)
    for key, ours in (("upper_grad_loss", "upper_grad_loss"), ("def_upper_loss", "def_upper_loss"),
                      ("upper_color_loss", "upper_color_loss"), ("upper_normal_loss", "upper_normal_loss")):
        torch.testing.assert_close(fake.info[ours].detach().float(), g["info_" + key], rtol=5e-4, atol=1e-6)
    sp, tp, rp = dict(sdf.named_parameters()), dict(tr.named_parameters()), dict(rn.named_parameters())
    got = {"g_sdf_" + k.replace(".", "_"): sp[k].grad for k in ["lin0.weight_v", "lin4.weight_g", "lin8.bias", "lin8.weight_v"]}
    got.update({"g_tr_" + k.replace(".", "_"): tp[k].grad for k in ["lin0.weight", "lin4.weight"]})
    got.update({"g_rn_" + k.replace(".", "_"): rp[k].grad for k in ["lin0.weight_v", "lin4.bias"]})
    for k, v in leaves.items():
        if "g_" + k in g:
            got["g_" + k] = v.grad
        else:           # the reference leaves no gradient on this leaf (translation, per-frame colour code): neither do we
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
    got["g_TmpPs"] = fake.TmpPs[0].grad
    import propagate_case as pc
    pc.compare(got, g, rtol=5e-3, atol_rel=5e-4)


def test_sample_train_ray_matches_the_reference_method():
    """OptimGarmentNetwork.sample_train_ray (:983-1055) run for real with a seeded host generator: same pixels kept
    (mask selection + Bernoulli subset drawn by torch's CPU generator), same rays."""
    import types
    from recmv.loop import HotLoop
    from recmv.model import RectifiedPerspectiveCameras
    g = load("sample_rays")
    found = [(g[f"in{i}_b"], g[f"in{i}_r"], g[f"in{i}_c"], g[f"in{i}_p"], None) for i in range(2)]
    fake = types.SimpleNamespace(conf={}, sample_pix=1024, garment_size=2, device='cpu', info={},
                                 _surface_inputs=(None, None), find_surface_ps=lambda d, t, c: found,
                                 dataset=types.SimpleNamespace(garment_masks=lambda g_i, fids: g["masks"][g_i]))
    cams = RectifiedPerspectiveCameras(g["focal"], g["pp"], g["R"], g["T"], image_size=[(40, 48)])
    torch.manual_seed(52)
    out = HotLoop.sample_train_ray(fake, 3, torch.arange(3), cams)
    for i, (b, r, c, p, rays) in enumerate(out):
        assert torch.equal(b, g[f"out{i}_b"]) and torch.equal(r, g[f"out{i}_r"]) and torch.equal(c, g[f"out{i}_c"])
        assert torch.equal(p, g[f"out{i}_p"])
        torch.testing.assert_close(rays, g[f"out{i}_rays"], rtol=1e-6, atol=1e-7)
    assert out[0][0].numel() < found[0][0].numel() and out[1][0].numel() > 500


def test_compute_fl_proj_loss_matches_the_reference_method():
    """OptimGarmentNetwork.compute_fl_proj_loss (:1605-1711) run for real: which samples count as visible (per-line
    z-buffer thresholds x label masks), the weighted chamfer normalisation, the curve regulariser with the config's
    weights — value and gradients w.r.t. the deformed samples and the curve parameters."""
    import types
    from recmv import curves as fl
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    from recmv.model import RectifiedPerspectiveCameras
    g = load("curve_proj")
    conf = ConfigFactory.parse_file(str(GOLD.parent.parent / "configs" / "synthetic" / "people_snapshot_like.conf"))
    names = ['neck', 'left_cuff', 'right_cuff', 'upper_bottom']
    curve = fl.Intersect_Free_Curve(list(g["curves"]), list(0.9 * g["curves"]), names)
    cam = RectifiedPerspectiveCameras(torch.tensor([[300., 295.]]), torch.tensor([[64., 60.]]),
                                      torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3),
                                      torch.tensor([[0.05, -0.1, 2.5]]), image_size=[(128, 120)])
    fake = types.SimpleNamespace(conf=conf.get_config('loss_coarse'), info={'fl_loss': {}}, inter_free_curve=curve,
                                 fl_extract={'upper': names},
                                 dataset=types.SimpleNamespace(H=120, W=128, fl_weights={'neck': 1.0, 'left_cuff': 2.0,
                                                                                          'right_cuff': 0.5,
                                                                                          'upper_bottom': 1.5}))
    defs = [d.clone().requires_grad_(True) for d in g["defs"]]
    checks = torch.cat(list(g["checks"]), dim=1)
    loss = HotLoop.compute_fl_proj_loss(fake, defs, checks, g["fl_masks"], g["gt"], 'upper', [30] * 4, cam)
    torch.testing.assert_close(loss, g["loss"], rtol=2e-5, atol=1e-6)
    grads = torch.autograd.grad(loss, defs + [curve.scale, curve.nx_scale])
    torch.testing.assert_close(torch.stack(grads[:4]), g["g_defs"], rtol=1e-3, atol=1e-8)
    torch.testing.assert_close(grads[4], g["g_scale"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(grads[5], g["g_nx"], rtol=1e-4, atol=1e-8)
    vis = float(fake.info['fl_loss']['upper_visible'])
    assert 0.2 < vis < 0.95 and abs(vis - float((checks[..., 1] < torch.tensor(
        [fl.ZBUF_THRESHOLD[n] for n in names]).repeat_interleave(30).view(1, -1)).float().mean())) < 1e-6


def test_fl_visibility_by_body_zbuffer_matches_the_reference_method():
    """OptimGarmentNetwork.fl_visible_by_body_zbuff (:1374-1448) run for real (reference deformer + depth logic, oracle
    rasteriser behind maskRender): [N,P,2] signed depth of every curve sample behind the garment surface and of its
    canonical-SMPL counterpart behind the body surface."""
    import types
    from oracle import cpu_port
    from recmv.loop import HotLoop
    from recmv.model import CompositeDeformer, LBSkinner, MLPTranslator, RectifiedPerspectiveCameras
    g = load("curve_vis")
    H, W, N = int(g["H"]), int(g["W"]), 3
    cam = RectifiedPerspectiveCameras(torch.tensor([[70., 68.]]), torch.tensor([[24., 30.]]),
                                      torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3),
                                      torch.tensor([[0.02, -0.05, 2.4]]), image_size=[(W, H)])
    ratio = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}
    cpu_port.install()
    try:
        comp = CompositeDeformer([cs.build_translator(MLPTranslator), cs.build_skinner(LBSkinner)])
        smpl_conds = [g["poses"], g["trans"]]
        fake = types.SimpleNamespace(deformer=comp, garment_fs=[g["gf"]], tmpBodyVs=g["bv"], tmpBodyFs=g["bf"],
                                     dataset=types.SimpleNamespace(H=H, W=W), _frag_cache={})
        fake._garment_fragments = types.MethodType(HotLoop._garment_fragments, fake)
        with torch.no_grad():
            fake._shared_def_vs = [comp(g["gv"][None].expand(N, -1, 3), [g["conds"], smpl_conds], ratio=ratio,
                                        offset_type="upper")]
            out = HotLoop.fl_visible_by_body_zbuff(fake, cam, g["conds"], smpl_conds, ratio, list(g["def_fl"]),
                                                   [c.view(1, -1, 3) for c in g["smpl"]], 0, "upper", N)
    finally:
        cpu_port.uninstall()
    assert out.shape == g["checks"].shape
    # a sample whose pixel sits on a silhouette edge reads a different mix of surface / background depth when the
    # deformed vertices differ in the last bits: allow a handful of such samples
    diff = (out - g["checks"]).abs()
    assert float((diff > 2e-4).float().mean()) < 0.03, float((diff > 2e-4).float().mean())
    assert float(diff.median()) < 1e-5
