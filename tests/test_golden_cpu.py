"""CPU-side parity of the host logic against the golden vectors generated from the REAL reference code
(tests/golden/make_golden.py): constructors (parameter-for-parameter initialisation), annealing weights,
positional encoding (torch path), cameras, Seg3dLossless (torch route) and the hot loop on the CPU port."""
import sys
from pathlib import Path

import numpy as np
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


def test_single_net_root_finders_match_the_reference_functions():
    """utils/FindSurfacePs.py:145-207 `OptimizeSurfacePs` and :210-272 `OptimizeGarmentSurfaceSinlge` run for real
    (tests/golden/make_golden_rootfind_single.py) against the same names here, on the CPU port: one step to rounding, and the whole
    iteration with the reference's call-site settings (1e-4 / 30 steps; 5e-5 / 10 steps) — the same rays converge to the same points."""
    from oracle import cpu_port
    from recmv.model import CompositeDeformer, LBSkinner, MLPTranslator, getTmpSdf
    from recmv.utils import OptimizeGarmentSurfaceSinlge, OptimizeSurfacePs
    g, gt, gl = load("rootfind"), load("translator"), load("lbs")
    gs = {k: torch.from_numpy(v) for k, v in np.load(GOLD / "rootfind_single.npz").items() if v.dtype.kind != "U"}
    ratio = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}
    cpu_port.install()
    try:
        sdf, tr, sk = cs.build_sdf(getTmpSdf), cs.build_translator(MLPTranslator), cs.build_skinner(LBSkinner)
        comp = CompositeDeformer([tr, sk])
        conds = [gt["conds"], [gl["poses"], gl["trans"]]]
        for tag, times in (("single1", 1), ("single", 30)):
            p, ok = OptimizeGarmentSurfaceSinlge(g["cam_pos"], g["rays"], g["start"].clone(), g["binds"], sdf, ratio, comp, conds,
                                                 dthreshold=1.e-4, athreshold=0.02, w1=3.05, w2=1., times=times, offset_type="upper")
            assert (ok == gs[tag + "_ok"]).float().mean() > (0.999 if times == 1 else 0.95), tag
            both = ok & gs[tag + "_ok"]
            assert both.sum() >= 15 and (p - gs[tag + "_p"])[both].abs().max() < (2e-6 if times == 1 else 2e-4), tag
        assert gs["single_ok"].sum() > 80
        for tag, times in (("base1", 1), ("base", 10)):
            p, ok = OptimizeSurfacePs(g["cam_pos"], gs["rays_lbs"], g["start"].clone(), g["binds"], sdf, ratio, sk,
                                      [gl["poses"], gl["trans"]], dthreshold=5.e-5, athreshold=0.02, w1=3.05, w2=1., times=times)
            assert (ok == gs[tag + "_ok"]).float().mean() > 0.98, tag
            both = ok & gs[tag + "_ok"]
            assert both.sum() >= 10 and (p - gs[tag + "_p"])[both].abs().max() < 2e-5, tag
    finally:
        cpu_port.uninstall()
    # the names resolve through the reference's module paths as well (recmv.namespace)
    import importlib
    F = importlib.import_module("recmv.utils.FindSurfacePs")
    assert {"OptimizeSurfacePs", "OptimizeGarmentSurfaceSinlge", "OptimizeGarmentSurfacePs", "FindSurfacePs"} <= set(F.__all__)
    import recmv.utils as U
    assert U.OptimizeSurfacePs is F.OptimizeSurfacePs and U.OptimizeGarmentSurfaceSinlge is F.OptimizeGarmentSurfaceSinlge


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
    HotLoop.propagateTmpPsGrad on the CPU port (tests/composite_cases.py; the same driver runs on the GPU)."""
    from oracle import cpu_port
    import composite_cases as cc
    cpu_port.install()
    try:
        cc.run_propagate("cpu")
    finally:
        cpu_port.uninstall()


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
    their gradients (tests/composite_cases.py)."""
    from oracle import cpu_port
    import composite_cases as cc
    cpu_port.install()
    try:
        cc.run_pc_loss("cpu")
    finally:
        cpu_port.uninstall()


def test_surface_render_loss_matches_the_reference_method():
    """OptimGarmentNetwork.surface_render_loss (:1083-1219) run for real (tests/golden/make_golden_render_loss.py) vs
    HotLoop.surface_render_loss on the CPU port with the same seed: value, per-term info, gradients
    (tests/composite_cases.py).  Differences by construction: closed-form singular values instead of the CPU SVD,
    jet pass instead of double backward."""
    from oracle import cpu_port
    import composite_cases as cc
    cpu_port.install()
    try:
        cc.run_render_loss("cpu")
    finally:
        cpu_port.uninstall()


def test_sample_train_ray_matches_the_reference_method():
    """OptimGarmentNetwork.sample_train_ray (:983-1055) run for real with a seeded host generator: same pixels kept,
    same rays (tests/composite_cases.py)."""
    from oracle import cpu_port
    import composite_cases as cc
    cpu_port.install()
    try:
        cc.run_sample_rays("cpu")
    finally:
        cpu_port.uninstall()


def test_compute_fl_proj_loss_matches_the_reference_method():
    """OptimGarmentNetwork.compute_fl_proj_loss (:1605-1711) run for real: value and gradients
    (tests/composite_cases.py)."""
    from oracle import cpu_port
    import composite_cases as cc
    cpu_port.install()
    try:
        cc.run_fl_proj("cpu")
    finally:
        cpu_port.uninstall()


def test_fl_visibility_by_body_zbuffer_matches_the_reference_method():
    """OptimGarmentNetwork.fl_visible_by_body_zbuff (:1374-1448) run for real (tests/composite_cases.py)."""
    from oracle import cpu_port
    import composite_cases as cc
    cpu_port.install()
    try:
        cc.run_fl_visibility("cpu")
    finally:
        cpu_port.uninstall()


def test_curve_aware_loss_matches_the_reference_method():
    """OptimGarmentNetwork.curve_aware_loss (:787-839) run for real (tests/golden/make_golden_curve_aware.py): the fan
    mesh of the `upper_bottom` curve, 50 000 samples, |SDF| of the last garment net — value, info, gradients; plus the
    product's device-side sampler against the trimesh stand-in (tests/composite_cases.py)."""
    from oracle import cpu_port
    import composite_cases as cc
    cpu_port.install()
    try:
        cc.run_curve_aware("cpu")
    finally:
        cpu_port.uninstall()


def test_fl_proj_loss_batched_and_per_pair_routes_agree():
    """curves.fl_proj_loss takes all (line, frame) chamfers as one distance tensor when every line has the same number of
    samples and of ground-truth points, and falls back to one chamfer per pair otherwise: same value and gradients either
    way (hidden samples, a frame that does not see a line, a line no frame sees, non-unit weights)."""
    from recmv import curves
    g = torch.Generator().manual_seed(21)
    L, N, S, M = 3, 4, 17, 11
    pts = [torch.randn(N, S, 3, generator=g).requires_grad_(True) for _ in range(L)]
    gts = [torch.randn(N, M, 2, generator=g) for _ in range(L)]
    masks = [(torch.rand(N, S, 1, generator=g) < 0.6).float().expand(N, S, 2).clone() for _ in range(L)]
    masks[0][1] = 0.                      # frame 1 does not see line 0
    masks[2][:] = 0.                      # nobody sees line 2
    w = [1.0, 2.5, 0.7]
    batched = curves.fl_proj_loss(pts, gts, masks, w)
    gb = torch.autograd.grad(batched, pts, allow_unused=True)
    # a ragged copy of the same problem (one extra hidden sample on line 1) takes the per-pair route
    pts2 = [p.detach().clone().requires_grad_(True) for p in pts]
    extra = torch.zeros(N, 1, 3)
    ragged_pts = [pts2[0], torch.cat([pts2[1], extra], dim=1), pts2[2]]
    ragged_masks = [masks[0], torch.cat([masks[1], torch.zeros(N, 1, 2)], dim=1), masks[2]]
    looped = curves.fl_proj_loss(ragged_pts, gts, ragged_masks, w)
    gl = torch.autograd.grad(looped, pts2, allow_unused=True)
    assert torch.allclose(batched, looped, rtol=1e-6, atol=1e-7)
    for a, b in zip(gb, gl):
        if a is None or b is None:
            assert (a is None or float(a.abs().max()) == 0.0) and (b is None or float(b.abs().max()) == 0.0)
        else:
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_mask_loss_matches_the_reference_method():
    """OptimGarmentNetwork.mask_loss (:841-981) run for real as a whole (tests/golden/make_golden_mask_loss.py): both garments
    deformed, the reference's PointsRendererWithFrags_Split around the merged cloud, dilated ground-truth masks, IoU + LBS
    consistency per garment, the SGD step on the explicit vertices, the |SDF| terms — value, info, moved vertices, and the
    gradients left for the main optimiser (tests/mask_loss_case.py)."""
    from oracle import cpu_port
    import mask_loss_case as mlc
    cpu_port.install()
    try:
        worst = mlc.run(load("mask_loss"), "cpu")
        assert max(worst.values()) < 1e-4, worst                  # (measured: <= 9e-6, the pose / translation gradients)
    finally:
        cpu_port.uninstall()


def test_project_2d_loss_matches_the_reference_method():
    """OptimGarmentNetwork.project_2d_loss (:1772-1883) run for real as a whole (tests/golden/make_golden_project2d.py): the
    reference's deform_feature_line, body / garment z-buffer visibility, compute_fl_proj_loss and curve SDF terms for two
    garments and six curves, one AdamW step — per-garment losses, the total, the curve-parameter gradients and the stepped
    parameters (tests/project2d_case.py)."""
    from oracle import cpu_port
    import project2d_case as p2c
    cpu_port.install()
    try:
        worst = p2c.run(load("project2d"), "cpu")
        assert max(worst.values()) < 1e-5, worst                  # (measured: <= 1.2e-7)
    finally:
        cpu_port.uninstall()


def test_one_whole_iteration_matches_the_reference():
    """OptimGarmentNetwork.forward (:1885-1969) -> loss.backward() -> propagateTmpPsGrad (:2159-2313) run for real with every
    method underneath the reference's own (tests/golden/make_golden_forward.py), against HotLoop.forward / backward /
    propagateTmpPsGrad on the same state with the same host random draws: the loss, every per-term info entry, the rays entering
    and converging per garment, the explicit vertices after their SGD step, the curve parameters after their AdamW step, and the
    gradients the main optimiser consumes — both SDF nets, offset MLP, colour net, per-frame codes, poses, translations, camera;
    then the main optimiser steps and a SECOND iteration's loss is compared (SGD momentum, AdamW state, forward_time carried over)
    (tests/forward_case.py)."""
    from oracle import cpu_port
    import forward_case as fwc
    cpu_port.install()
    try:
        worst = fwc.run(load("forward"), "cpu", rtol=1e-4, rtol_grad=5e-3)
        big = {k: v for k, v in worst.items() if v > 5e-5}
        assert set(big) <= {'g_focal', 'g_pp'}, big          # (measured: everything <= 4.3e-5 but the two intrinsics' gradients, ~1e-3)
        assert 'loss of the second iteration' in worst       # ... and the iteration AFTER the Adam step: 3.5e-5
    finally:
        cpu_port.uninstall()


def test_one_whole_large_pose_iteration_matches_the_reference():
    """The same whole iteration on the large-pose stage: OptimGarmentNetwork_LargePose.forward (OptimGarmentNetwork_Large_Pose.py:
    242-323, its zero-weighted project_2d_loss :150-240 and its propagateTmpPsGrad :326-475, SDF nets frozen by freeze_sdf
    :130-137) vs the large-pose HotLoop; the frozen nets receive no gradient on either side."""
    from oracle import cpu_port
    import forward_case as fwc
    cpu_port.install()
    try:
        worst = fwc.run(load("forward_large"), "cpu", rtol=1e-4, rtol_grad=5e-3, large_pose=True, inputs=load("forward"))
        big = {k: v for k, v in worst.items() if v > 5e-5}
        assert set(big) <= {'g_focal', 'g_pp'}, big
    finally:
        cpu_port.uninstall()


def test_one_whole_iteration_with_the_remesh_inside_matches_the_reference():
    """The whole iteration again, starting at forward_time = 0: the reference's marching_cube_update (:678-740) -> discretizeSDF
    (:581-618) — its own Seg3dLossless pyramid over the body net and both garment nets, MC through the oracle — then the iteration
    on the freshly extracted meshes.  recmv's lockstep device-side pyramid + MC gives the same faces bit for bit and the same
    vertices to f32 interpolation error; everything downstream agrees as in the fixed-mesh case."""
    from oracle import cpu_port
    import forward_case as fwc
    cpu_port.install()
    try:
        worst = fwc.run(load("forward_remesh"), "cpu", rtol=5e-4, rtol_grad=2e-2, inputs=load("forward"), remesh=True)
        big = {k: v for k, v in worst.items() if v > 2e-4}
        assert set(big) <= {'g_focal', 'g_pp'} and max(big.values(), default=0.) < 5e-2, big
    finally:
        cpu_port.uninstall()


def test_one_whole_single_garment_iteration_matches_the_reference():
    """The whole iteration for a capture with ONE one-piece garment — `leyang_jump` = ['dress'] with train.is_upper_bottom
    (utils/constant.py:105, configs/female_large_pose/leyang_jump*.conf:5): the reference's forward takes the union region
    `datas['upper_bottom']` (:1901-1904), splits the deformer code in body + one garment (:670-676), deforms the dress's four lines
    (FL_EXTRACT), adds no curve-aware disc; HotLoop builds the same loop from train.garment_type alone."""
    from oracle import cpu_port
    import forward_case as fwc
    cpu_port.install()
    try:
        worst = fwc.run(load("forward_single"), "cpu", rtol=1e-4, rtol_grad=5e-3, single=True)
        big = {k: v for k, v in worst.items() if v > 5e-5}
        assert set(big) <= {'g_focal', 'g_pp'}, big
        worst = fwc.run(load("forward_single_large"), "cpu", rtol=1e-4, rtol_grad=5e-3, single=True, large_pose=True,
                        inputs=load("forward_single"))           # (OptimGarmentNetwork_Large_Pose.py:250-258, same switch)
        big = {k: v for k, v in worst.items() if v > 5e-5}
        assert set(big) <= {'g_focal', 'g_pp'}, big
    finally:
        cpu_port.uninstall()
