"""Device-parametric drivers of the composite-method golden tests.

Each `run_*` rebuilds a fixture's state on recmv modules ON `device`, calls the HotLoop method on a stand-in `self`
exactly as tests/golden/make_golden_*.py called the reference's method, and compares with the fixture.  The CPU tests
(tests/test_golden_cpu.py, through oracle/cpu_port) and the GPU tests (tests/test_gpu_composite.py, through
librecmv_hip.so — the fused / jet / chain branches of the product) share these drivers, so the HIP path meets the SAME
reference numbers as the CPU port.
"""
import contextlib
import sys
import types
from pathlib import Path

import numpy as np
import torch

GOLD = Path(__file__).resolve().parent / "golden"
REPO = GOLD.parent.parent
if str(GOLD) not in sys.path:
    sys.path.insert(0, str(GOLD))
import common_setup as cs  # noqa: E402
import propagate_case as pc  # noqa: E402

RATIO = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}
CONF = str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf")


def load(name, device="cpu"):
    return {k: torch.from_numpy(v).to(device) for k, v in np.load(GOLD / f"{name}.npz").items()}


def loss_conf(stage="coarse"):
    from recmv.hocon import ConfigFactory
    return ConfigFactory.parse_file(CONF).get_config("loss_" + stage)


@contextlib.contextmanager
def host_draws():
    """The fixtures were generated with torch's HOST generator.  On the GPU the product draws from the device generator
    (a different stream), so for parity the random draws are taken from the host generator and moved: same numbers on
    both devices, nothing else changes."""
    names = ("rand", "randn", "rand_like", "randn_like")
    real = {n: getattr(torch, n) for n in names}

    def wrap(fn, like):
        def f(*a, **k):
            if like:
                src = a[0]
                return fn(src.detach().cpu(), *a[1:], **k).to(src.device)
            dev = k.pop("device", None)
            out = fn(*a, **k)
            return out.to(dev) if dev is not None else out
        return f

    for n in names:
        setattr(torch, n, wrap(real[n], n.endswith("_like")))
    try:
        yield
    finally:
        for n in names:
            setattr(torch, n, real[n])


def build_nets(device):
    from recmv.model import CompositeDeformer, LBSkinner, MLPTranslator, RenderingNetwork_view_norm, getTmpSdf
    sdf = cs.build_sdf(getTmpSdf).to(device)
    tr = cs.build_translator(MLPTranslator).to(device)
    sk = cs.build_skinner(LBSkinner).to(device)
    rn = cs.build_render(RenderingNetwork_view_norm).to(device)
    return dict(sdf=sdf, tr=tr, sk=sk, rn=rn, comp=CompositeDeformer([tr, sk]))


# ------------------------------------------------------------------------------------------------ propagateTmpPsGrad
def run_propagate(device, rtol=2e-3, atol_rel=2e-4, large_pose=False):
    """OptimGarmentNetwork.propagateTmpPsGrad (:2159-2313); large_pose: the OptimGarmentNetwork_LargePose variant
    (OptimGarmentNetwork_Large_Pose.py:326-475) with its frozen SDF nets."""
    g = load("propagate_large" if large_pose else "propagate")
    n = build_nets(device)
    out, n_total, n_ok = pc.run(g, n["sdf"], n["tr"], n["comp"], device, large_pose=large_pose)
    assert (n_total, n_ok) == (int(g["inv_total"]), int(g["inv_ok"]))
    pc.compare(out, g, rtol=rtol, atol_rel=atol_rel)


# ------------------------------------------------------------------------------------------------ surface_render_loss
def run_render_loss(device, rtol_loss=2e-4, rtol_info=5e-4, rtol_grad=5e-3, atol_rel=5e-4):
    """OptimGarmentNetwork.surface_render_loss (:1083-1219): value, per-term info, and the gradients `backward()` leaves
    on the SDF net, the offset MLP, the colour net, the per-frame codes, the poses and the surface points."""
    from recmv.loop import HotLoop
    g = load("render_loss", device)
    n = build_nets(device)
    sdf, tr, comp, rn = n["sdf"], n["tr"], n["comp"], n["rn"]
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    leaves = dict(conds=leaf(g["conds"]), poses=leaf(g["poses"]), trans=leaf(g["trans"]), rendcond=leaf(g["in_rendcond"]))
    fake = types.SimpleNamespace(conf=loss_conf(), device=device, garment_size=1, garment_names=['upper'],
                                 garment_nets=[sdf], deformer=comp, netRender=rn, info={})
    fake.garment_vs = [g["in_verts"].clone().requires_grad_(True)]
    fake._gt_images = lambda fids: (g["in_gtC"], g["in_gtN"])
    fake.get_grad_parameters = lambda fids, dev: ([None, leaves["conds"]], leaves["poses"], leaves["trans"],
                                                  leaves["rendcond"])
    fake._ray_valid = [g["in_check"].sum()]
    fake._garment_render_terms = types.MethodType(HotLoop._garment_render_terms, fake)
    cameras = types.SimpleNamespace(R=g["in_R"])
    samples = [(g["in_binds"], g["in_row"], g["in_col"], None, g["in_rays"])]
    torch.manual_seed(int(g["seed"]))
    with host_draws():
        loss = HotLoop.surface_render_loss(fake, 3, cameras, torch.arange(3, device=device), RATIO, [g["in_check"]],
                                           [g["in_init"].clone()], samples)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), g["loss"].cpu(), rtol=rtol_loss, atol=1e-5)
    for key in ("upper_grad_loss", "def_upper_loss", "upper_color_loss", "upper_normal_loss"):
        torch.testing.assert_close(fake.info[key].detach().float().cpu(), g["info_" + key].cpu(), rtol=rtol_info, atol=1e-6)
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
    pc.compare(got, {k: v.cpu() for k, v in g.items()}, rtol=rtol_grad, atol_rel=atol_rel)


# ------------------------------------------------------------------------------------------------ compute_garment_pc_loss
def run_pc_loss(device):
    """OptimGarmentNetwork.compute_garment_pc_loss (:621-667): silhouette IoU + LBS-consistency term and their gradients
    w.r.t. the silhouette and the explicit vertices."""
    from recmv.loop import HotLoop
    from recmv.model import LBSkinner
    g = load("misc", device)
    sk = cs.build_skinner(LBSkinner).to(device)
    fake = types.SimpleNamespace(conf=loss_conf(), info={}, deformer=types.SimpleNamespace(defs=[None, sk]))
    imgs, verts = g["pc_imgs"].clone().requires_grad_(True), g["pc_verts"].clone().requires_grad_(True)
    N = imgs.shape[0]
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


# ------------------------------------------------------------------------------------------------ sample_train_ray
def run_sample_rays(device):
    """OptimGarmentNetwork.sample_train_ray (:983-1055) with a seeded host generator: same pixels kept (mask selection +
    Bernoulli subset drawn by torch's CPU generator, as the reference draws it), same rays."""
    from recmv.loop import HotLoop
    from recmv.model import RectifiedPerspectiveCameras
    g = load("sample_rays", device)
    found = [(g[f"in{i}_b"], g[f"in{i}_r"], g[f"in{i}_c"], g[f"in{i}_p"], None) for i in range(2)]
    fake = types.SimpleNamespace(conf={}, sample_pix=1024, garment_size=2, device=device, info={},
                                 _surface_inputs=(None, None), find_surface_ps=lambda d, t, c: found,
                                 _gt_garment_mask=lambda g_i, fids: g["masks"][g_i])
    if torch.device(device).type == "cuda":
        fake._surface_stream = None
        fake._surface_ready = torch.cuda.Event()
        fake._surface_ready.record()
    cams = RectifiedPerspectiveCameras(g["focal"], g["pp"], g["R"], g["T"], image_size=[(40, 48)])
    torch.manual_seed(52)
    out = HotLoop.sample_train_ray(fake, 3, torch.arange(3, device=device), cams)
    for i, (b, r, c, p, rays) in enumerate(out):
        assert torch.equal(b, g[f"out{i}_b"]) and torch.equal(r, g[f"out{i}_r"]) and torch.equal(c, g[f"out{i}_c"])
        assert torch.equal(p, g[f"out{i}_p"])
        torch.testing.assert_close(rays, g[f"out{i}_rays"], rtol=1e-6, atol=1e-7)
    assert out[0][0].numel() < found[0][0].numel() and out[1][0].numel() > 500


# ------------------------------------------------------------------------------------------------ compute_fl_proj_loss
def run_fl_proj(device):
    """OptimGarmentNetwork.compute_fl_proj_loss (:1605-1711): which samples count as visible (per-line z-buffer
    thresholds x label masks), the weighted chamfer normalisation, the curve regulariser with the config's weights —
    value and gradients w.r.t. the deformed samples and the curve parameters."""
    from recmv import curves as fl
    from recmv.loop import HotLoop
    from recmv.model import RectifiedPerspectiveCameras
    g = load("curve_proj", device)
    T = lambda *a: torch.tensor(*a, device=device)
    names = ['neck', 'left_cuff', 'right_cuff', 'upper_bottom']
    curve = fl.Intersect_Free_Curve(list(g["curves"]), list(0.9 * g["curves"]), names).to(device)
    cam = RectifiedPerspectiveCameras(T([[300., 295.]]), T([[64., 60.]]), torch.diag(T([-1., -1., 1.])).view(1, 3, 3),
                                      T([[0.05, -0.1, 2.5]]), image_size=[(128, 120)])
    fake = types.SimpleNamespace(conf=loss_conf(), info={'fl_loss': {}}, inter_free_curve=curve,
                                 fl_extract={'upper': names},
                                 dataset=types.SimpleNamespace(H=120, W=128, fl_weights={'neck': 1.0, 'left_cuff': 2.0,
                                                                                          'right_cuff': 0.5,
                                                                                          'upper_bottom': 1.5}))
    defs = [d.clone().requires_grad_(True) for d in g["defs"]]
    checks = torch.cat(list(g["checks"]), dim=1)
    loss = HotLoop.compute_fl_proj_loss(fake, defs, checks, g["fl_masks"], g["gt"], 'upper', [30] * 4, cam)
    torch.testing.assert_close(loss, g["loss"], rtol=2e-5, atol=1e-6)
    grads = torch.autograd.grad(loss, defs + [curve.scale, curve.nx_scale])
    # f32: elements that are a difference of nearly equal chamfer pulls carry the summation-order error of the whole
    # tensor, so the absolute tolerance is relative to the tensor's scale (1e-4 of max|g|)
    for got, key, rtol in ((torch.stack(grads[:4]), "g_defs", 1e-3), (grads[4], "g_scale", 1e-4), (grads[5], "g_nx", 1e-4)):
        torch.testing.assert_close(got, g[key], rtol=rtol, atol=1e-4 * float(g[key].abs().max()))
    vis = float(fake.info['fl_loss']['upper_visible'])
    thr = T([fl.ZBUF_THRESHOLD[n] for n in names]).repeat_interleave(30).view(1, -1)
    assert 0.2 < vis < 0.95 and abs(vis - float((checks[..., 1] < thr).float().mean())) < 1e-6


# ------------------------------------------------------------------------------------------------ fl_visible_by_body_zbuff
def run_fl_visibility(device):
    """OptimGarmentNetwork.fl_visible_by_body_zbuff (:1374-1448): [N,P,2] signed depth of every curve sample behind the
    garment surface and of its canonical-SMPL counterpart behind the body surface (reference deformer + depth logic; the
    fixture's rasteriser was the C oracle, here the product's — the HIP kernel on the GPU)."""
    from recmv.loop import HotLoop
    from recmv.model import RectifiedPerspectiveCameras
    g = load("curve_vis", device)
    T = lambda *a: torch.tensor(*a, device=device)
    H, W, N = int(g["H"]), int(g["W"]), 3
    cam = RectifiedPerspectiveCameras(T([[70., 68.]]), T([[24., 30.]]), torch.diag(T([-1., -1., 1.])).view(1, 3, 3),
                                      T([[0.02, -0.05, 2.4]]), image_size=[(W, H)])
    comp = build_nets(device)["comp"]
    smpl_conds = [g["poses"], g["trans"]]
    fake = types.SimpleNamespace(deformer=comp, garment_fs=[g["gf"]], tmpBodyVs=g["bv"], tmpBodyFs=g["bf"],
                                 dataset=types.SimpleNamespace(H=H, W=W), _frag_cache={}, device=device)
    fake._garment_fragments = types.MethodType(HotLoop._garment_fragments, fake)
    with torch.no_grad():
        fake._shared_def_vs = [comp(g["gv"][None].expand(N, -1, 3), [g["conds"], smpl_conds], ratio=RATIO,
                                    offset_type="upper")]
        out = HotLoop.fl_visible_by_body_zbuff(fake, cam, g["conds"], smpl_conds, RATIO, list(g["def_fl"]),
                                               [c.view(1, -1, 3) for c in g["smpl"]], 0, "upper", N)
    assert out.shape == g["checks"].shape
    # a sample whose pixel sits on a silhouette edge reads a different mix of surface / background depth when the
    # deformed vertices differ in the last bits: allow a handful of such samples
    diff = (out - g["checks"]).abs()
    assert float((diff > 2e-4).float().mean()) < 0.03, float((diff > 2e-4).float().mean())
    assert float(diff.median()) < 1e-5


# ------------------------------------------------------------------------------------------------ curve_aware_loss
def run_curve_aware(device, rtol=2e-4, rtol_grad=5e-3):
    """OptimGarmentNetwork.curve_aware_loss (:787-839): the same 50 000 fan-mesh samples (TrimeshStandIn on the fixture's
    seed) through HotLoop.curve_aware_loss: value, info entry, gradients on the last garment net only."""
    from recmv import curves as fl
    from recmv.loop import HotLoop, fan_mesh, sample_fan_mesh
    from recmv.model import getTmpSdf
    g = load("curve_aware", device)
    names = ['neck', 'upper_bottom', 'left_pant']
    ring = cs.curve_aware_ring()
    curves = [ring * 0.5 + torch.tensor([0., 0.6, 0.]), ring, ring * 0.4 + torch.tensor([0.1, -0.3, 0.])]
    curve = fl.Intersect_Free_Curve(curves, [0.9 * c for c in curves], names).to(device)
    with torch.no_grad():
        curve.scale.copy_(g["scale"])
        curve.nx_scale.copy_(g["nx_scale"])
    net, other = cs.build_sdf(getTmpSdf).to(device), cs.build_sdf(getTmpSdf).to(device)
    np.testing.assert_allclose(cs.fingerprint(net), g["fingerprint"].cpu().numpy(), rtol=1e-5, atol=1e-4)
    cs.TrimeshStandIn.rng = np.random.RandomState(int(g["seed"]))

    def sampler(verts, faces, n):
        pts = cs.TrimeshStandIn(verts.detach().cpu().numpy(), faces.cpu().numpy()).sample(n)
        return torch.from_numpy(pts).float().to(device)

    fake = types.SimpleNamespace(conf=loss_conf(), fl_names=names, inter_free_curve=curve, garment_nets=[other, net],
                                 sdfShrinkRadius=0.0, info={}, garment_type='female-3-casual', isfine=False, curves=True,
                                 CURVE_AWARE=HotLoop.CURVE_AWARE, CURVE_AWARE_SAMPLES=HotLoop.CURVE_AWARE_SAMPLES)
    loss = HotLoop.curve_aware_loss(fake, {"sdfRatio": 1.0, "deformerRatio": 0.7, "renderRatio": 1.0}, sampler=sampler)
    loss.backward()
    torch.testing.assert_close(loss.detach(), g["loss"], rtol=rtol, atol=1e-6)
    torch.testing.assert_close(fake.info['pc_upper_bottom_circle_loss_sdf'].float(), g["info"], rtol=rtol, atol=1e-7)
    assert all(p.grad is None for p in other.parameters()) and curve.scale.grad is None
    params = dict(net.named_parameters())
    got = {"g_" + k.replace(".", "_"): params[k].grad for k in cs.SDF_GRAD_KEYS}
    pc.compare(got, {k: v.cpu() for k, v in g.items()}, rtol=rtol_grad, atol_rel=5e-4)
    # the product's own device-side sampler: same distribution (points in the fan's triangles, face frequencies ~ area)
    verts, faces = fan_mesh(curve()[1].detach())
    gen = torch.Generator(device=device).manual_seed(3)
    pts = sample_fan_mesh(verts, faces, 200000, generator=gen)
    ref = sampler(verts, faces, 200000)
    assert pts.shape == ref.shape and torch.isfinite(pts).all()
    # moments of the two sample sets agree (mean to 3 sigma of the sampling error, covariance loosely)
    err = (pts.mean(0) - ref.mean(0)).abs() / (ref.std(0) / (200000 ** 0.5) * 2 ** 0.5)
    assert float(err.max()) < 4.5, err
    assert float((pts.var(0) / ref.var(0) - 1).abs().max()) < 0.02
    # every sample lies in the plane-ish fan: distance to the nearest fan triangle's plane is ~0
    tri = verts[faces]
    nrm = torch.nn.functional.normalize(torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=-1), dim=-1)
    d = ((pts[:4096, None, :] - tri[None, :, 0]) * nrm[None]).sum(-1).abs().min(1).values
    assert float(d.max()) < 1e-5
