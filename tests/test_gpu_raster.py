"""HIP first-hit rasteriser vs the CPU oracle, through the C ABI — bit-exact (indices AND floats: both sides run
the same un-contracted IEEE sequence per (pixel, face) pair), plus the full-size properties of the surface-point path
(find_surface_ps -> FindSurfacePs, OptimGarmentNetwork.py:742-767)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _soup(seed, n_faces, size, zlo=0.5, zhi=3.0, lo=-1.2, hi=1.2):
    rng = np.random.default_rng(seed)
    c = rng.uniform(lo, hi, size=(n_faces, 1, 2))
    xy = c + rng.uniform(-size, size, size=(n_faces, 3, 2))
    z = rng.uniform(zlo, zhi, size=(n_faces, 3, 1))
    return torch.from_numpy(np.concatenate([xy, z], -1).astype(np.float32))


def _same(frag, ref):
    names = ("pix_to_face", "zbuf", "bary_coords", "dists")
    for name, a, b in zip(names, frag, ref):
        a = a.cpu()
        assert a.shape == b.shape and a.dtype == b.dtype, name
        if a.dtype.is_floating_point:
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), name + " must be bit-identical"
        else:
            assert torch.equal(a, b), name + " must be identical"


@pytest.mark.parametrize("case", [
    dict(seed=0, faces=[300], size=0.05, hw=(64, 64)),                 # sub-pixel .. few-pixel faces
    dict(seed=1, faces=[200, 0, 333], size=0.2, hw=(48, 80)),          # ragged batch with an empty mesh, H != W
    dict(seed=2, faces=[40], size=1.5, hw=(96, 96)),                   # faces covering most of the image (wave path)
    dict(seed=3, faces=[500, 500], size=0.1, hw=(33, 17), blur=3e-3),  # soft coverage
    dict(seed=4, faces=[256], size=0.3, hw=(64, 64), cull=True),
    dict(seed=5, faces=[256], size=0.3, hw=(64, 64), persp=False),
    dict(seed=6, faces=[128], size=0.4, hw=(40, 40), zlo=-1.0),        # faces straddling / behind the camera plane
])
def test_rasterize_bit_exact(oracle, case):
    from recmv import raster
    nf = case["faces"]
    fv = _soup(case["seed"], sum(nf), case["size"], zlo=case.get("zlo", 0.5))
    if sum(nf) > 20:                      # adversarial faces: duplicate (tie), zero area, vertex exactly on a centre
        fv[5] = fv[4]
        fv[7, 2] = fv[7, 1]
        H, W = case["hw"]
        fv[9, 0, 0], fv[9, 0, 1] = 1 - (2 * 3 + 1) / W, 1 - (2 * 2 + 1) / H
    first = torch.tensor([0] + list(np.cumsum(nf)[:-1]), dtype=torch.int64)
    num = torch.tensor(nf, dtype=torch.int64)
    kw = dict(blur_radius=case.get("blur", 0.0), perspective_correct=case.get("persp", True),
              cull_backfaces=case.get("cull", False))
    ref = oracle.rasterize_meshes(fv, first, num, case["hw"], **kw)
    got = raster.rasterize_meshes(fv.to(DEV), first.to(DEV), num.to(DEV), case["hw"], max_faces_per_mesh=max(nf), **kw)
    _same(got, ref)
    assert (ref[0] >= 0).sum() > 10
    # the default launch bound (max_faces_per_mesh = F) gives the same answer
    _same(raster.rasterize_meshes(fv.to(DEV), first.to(DEV), num.to(DEV), case["hw"], **kw), ref)


def test_rasterize_argument_errors():
    from recmv import raster
    fv = torch.zeros(4, 3, 3, device=DEV)
    first, num = torch.tensor([0], device=DEV), torch.tensor([4], device=DEV)
    with pytest.raises(ValueError):
        raster.rasterize_meshes(fv, first, num, (8, 8), faces_per_pixel=2)
    with pytest.raises(ValueError):
        raster.rasterize_meshes(fv.double(), first, num, (8, 8))
    with pytest.raises(ValueError):
        raster.rasterize_meshes(fv, first.int(), num, (8, 8))
    with pytest.raises(RuntimeError):
        raster.rasterize_meshes(fv.cpu(), first, num, (8, 8))
    out = raster.rasterize_meshes(fv, first, num, (8, 8))          # all faces degenerate: nothing drawn
    assert (out.pix_to_face == -1).all() and (out.zbuf == -1).all()
    none = raster.rasterize_meshes(fv[:0], first[:0], num[:0], (8, 8))
    assert none.pix_to_face.shape == (0, 8, 8, 1)


def _sphere_mesh(res=65, radius=0.45):
    from recmv import MCGpu
    ax = torch.linspace(-0.6, 0.6, res, device=DEV)
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    sdf = (x * x + y * y * 1.3 + z * z).sqrt() - radius
    step = 1.2 / (res - 1)
    return MCGpu.mc_gpu(sdf.contiguous(), step, step, step, -0.6, -0.6, -0.6, 0.0)


def _camera(H, W, n=1):
    from recmv.model import RectifiedPerspectiveCameras
    f = torch.tensor([[1.8 * W, 1.8 * W]], device=DEV)
    pp = torch.tensor([[W / 2 - 0.5 + 3.25, H / 2 - 0.5 - 2.5]], device=DEV)
    R = torch.tensor([[[-1., 0., 0.], [0., 1., 0.], [0., 0., -1.]]], device=DEV)
    T = torch.tensor([[0.05, -0.02, 2.6]], device=DEV)
    return RectifiedPerspectiveCameras(f, pp, R, T, image_size=[(W, H)])


def test_mesh_rasterizer_matches_oracle_on_a_garment_mesh(oracle):
    """An MC mesh through the camera path: same fragments as the oracle, hence the same FindSurfacePs indices."""
    from recmv import raster, utils
    verts, faces = _sphere_mesh(33)
    H, W = 72, 56
    cam = _camera(H, W)
    offs = torch.tensor([[0., 0., 0.], [0.07, -0.03, 0.2]], device=DEV)
    def_vs = verts[None] + offs[:, None]
    rast = raster.MeshRasterizer(cam, (H, W))
    frags = rast(def_vs, faces)
    ndc = cam.transform_points_ndc(def_vs.reshape(-1, 3)).view(2, -1, 3)
    fv = ndc[:, faces.reshape(-1)].reshape(-1, 3, 3).cpu()
    F = faces.shape[0]
    ref = oracle.rasterize_meshes(fv, torch.tensor([0, F]), torch.tensor([F, F]), (H, W))
    _same(frags, ref)
    got = utils.FindSurfacePs(verts, faces, frags)
    want = utils.FindSurfacePs(verts.cpu(), faces.cpu(), raster.Fragments(*ref))
    for a, b in zip(got[:3] + got[4:], want[:3] + want[4:]):
        assert torch.equal(a.cpu(), b)                     # batch / row / col / face indices bit-exact
    assert torch.allclose(got[3].cpu(), want[3], atol=1e-6)
    assert got[0].numel() > 500


def test_surface_points_project_to_their_pixel_centres_full_size():
    """512 x 512, 3 frames, ~80k-vertex mesh (the bench workload's shape): every visible canonical point, moved by
    the frame's translation, projects onto the centre of the pixel it was found through, and the ray of that pixel
    (view_rays) passes through it; the frontmost face wins (no hit lies behind another face of the ray)."""
    from recmv import raster, utils
    verts, faces = _sphere_mesh(129)
    H = W = 512
    cam = _camera(H, W)
    offs = torch.tensor([[0., 0., 0.], [0.07, -0.03, 0.2], [-0.1, 0.05, -0.1]], device=DEV)
    def_vs = verts[None] + offs[:, None]
    frags = raster.MeshRasterizer(cam, (H, W))(def_vs, faces)
    b, r, c, p0, finds = utils.FindSurfacePs(verts, faces, frags)
    assert b.numel() > 100000
    world = p0 + offs[b]
    pix = cam.project(world)
    assert (pix[:, 0] - c).abs().max() < 2e-2 and (pix[:, 1] - r).abs().max() < 2e-2
    rays = cam.view_rays(torch.stack([c, r, torch.ones_like(c)], -1).float())
    d = world - cam.cam_pos().view(1, 3)
    ang = torch.linalg.cross(d, rays, dim=1).norm(dim=1) / d.norm(dim=1)
    assert ang.max() < 2e-5
    # convex body: exactly the front half is visible -> the face normal of every hit looks at the camera
    tri = def_vs[b[:, None], faces[finds]]                                    # [M,3,3]
    n = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=1)
    front = (n * d).sum(-1)
    assert ((front > 0).float().mean() > 0.999) or ((front < 0).float().mean() > 0.999)
    # determinism: the scatter formulation gives the same image every time
    again = raster.MeshRasterizer(cam, (H, W))(def_vs, faces)
    assert torch.equal(again.pix_to_face, frags.pix_to_face) and torch.equal(again.zbuf, frags.zbuf)


# ------------------------------------------------------------------------------------------- points
def _cloud(seed, n, zlo=0.5, zhi=3.0):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-1.05, 1.05, size=(n, 2))
    z = rng.uniform(zlo, zhi, size=(n, 1))
    return torch.from_numpy(np.concatenate([xy, z], 1).astype(np.float32))


def _bits(a, b, name):
    a = a.cpu()
    assert a.shape == b.shape and a.dtype == b.dtype, name
    if a.dtype.is_floating_point:
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), name + " must be bit-identical"
    else:
        assert torch.equal(a, b), name + " must be identical"


@pytest.mark.parametrize("case", [
    dict(seed=0, clouds=[4000], hw=(48, 48), K=8, r=0.05),
    dict(seed=1, clouds=[1500, 0, 2500], hw=(40, 72), K=50, r=0.11),          # ragged + empty cloud, H != W, K = 50
    dict(seed=2, clouds=[6000], hw=(24, 24), K=3, r=0.3, zlo=-0.3),           # saturated lists, points behind the camera
    dict(seed=3, clouds=[3000, 3000], hw=(64, 64), K=50, r=0.006 * 8),        # the loop's splat size, scaled to 64 px
])
def test_rasterize_points_and_composite_vs_oracle(oracle, case):
    from recmv import raster
    n = case["clouds"]
    pts = _cloud(case["seed"], sum(n), zlo=case.get("zlo", 0.5))
    if sum(n) > 100:
        pts[11] = pts[10]                                                     # equal depth and position: index decides
        H, W = case["hw"]
        pts[12, 0], pts[12, 1] = 1 - (2 * 5 + 1) / W, 1 - (2 * 4 + 1) / H     # exactly on a pixel centre (alpha = 1)
    first = torch.tensor([0] + list(np.cumsum(n)[:-1]), dtype=torch.int64)
    num = torch.tensor(n, dtype=torch.int64)
    K, r = case["K"], case["r"]
    ref = oracle.rasterize_points(pts, first, num, case["hw"], r, K)
    p_gpu = pts.to(DEV).requires_grad_(True)
    frags = raster.rasterize_points(p_gpu, first.to(DEV), num.to(DEV), case["hw"], r, K, max_points_per_cloud=max(n))
    for name, a, b in zip(("idx", "zbuf", "dists"), frags, ref):
        _bits(a.detach(), b, name)
    assert (ref[0] >= 0).sum() > 100
    # compositor: the forward is a fixed-order sum -> bit-exact; grad_features / grad_points use atomics
    rng = np.random.default_rng(case["seed"] + 100)
    feats = torch.from_numpy(rng.uniform(0, 1, size=(2, sum(n))).astype(np.float32))
    valid = ref[0] >= 0
    alphas_ref = (1 - ref[2] / (r * r)) * valid
    alphas_ref[valid] = alphas_ref[valid].clamp(max=0.999)
    if valid.sum() > 50:
        alphas_ref.view(-1)[valid.view(-1).nonzero()[3]] = 1.0       # a point exactly on a pixel centre: must stay finite
    img_ref = oracle.alpha_composite_forward(ref[0], alphas_ref, feats)
    f_gpu = feats.to(DEV).requires_grad_(True)
    a_gpu = alphas_ref.to(DEV).requires_grad_(True)
    img = raster.alpha_composite(frags.idx, a_gpu, f_gpu)
    _bits(img.detach(), img_ref, "images")
    g = torch.from_numpy(rng.normal(size=tuple(img_ref.shape)).astype(np.float32))
    img.backward(g.to(DEV))
    ga_ref, gf_ref = oracle.alpha_composite_backward(ref[0], alphas_ref, feats, g)
    # grad_alphas: the kernel evaluates the published O(K^2) double loop through an O(K) suffix recursion
    # (different summation order) -> tolerance, not bits
    assert torch.isfinite(ga_ref).all() and torch.isfinite(a_gpu.grad).all()
    assert torch.allclose(a_gpu.grad.cpu(), ga_ref, rtol=1e-4, atol=2e-6 * float(ga_ref.abs().max()))
    assert torch.allclose(f_gpu.grad.cpu(), gf_ref, rtol=1e-4, atol=1e-5)
    # fused opacities (1 - dists / r^2 inside the kernels) == the composition of torch ops, values and gradients
    d1 = frags.dists.detach().clone().requires_grad_(True)
    d2 = frags.dists.detach().clone().requires_grad_(True)
    im1 = raster.alpha_composite(frags.idx, 1 - d1 / (r * r), f_gpu.detach())
    im2 = raster.alpha_composite_dists(frags.idx, d2, r, f_gpu.detach())
    assert torch.allclose(im1, im2, rtol=1e-5, atol=1e-6)     # torch divides by a scalar as x * (1/s): 1 ulp apart
    im1.backward(g.to(DEV))
    im2.backward(g.to(DEV))
    vg = valid.to(DEV)
    assert torch.allclose(d1.grad[vg], d2.grad[vg], rtol=1e-4, atol=1e-6 * float(d1.grad[vg].abs().max()))
    assert (d2.grad[~vg] == 0).all()
    gd = torch.from_numpy(rng.normal(size=tuple(ref[2].shape)).astype(np.float32)) * valid
    gz = torch.from_numpy(rng.normal(size=tuple(ref[2].shape)).astype(np.float32)) * valid
    (frags.dists * gd.to(DEV) + frags.zbuf * gz.to(DEV)).sum().backward()
    gp_ref = oracle.rasterize_points_backward(pts, ref[0], gd, gz)
    # one thread per point, pixels summed in row-major order = the oracle's order: bit-exact (upstream and round 1 used
    # float atomics here: same value up to summation order, different bits from run to run)
    assert torch.equal(p_gpu.grad.cpu(), gp_ref)


def test_points_renderer_split_full_size_properties():
    """3 x 512 x 512, two garments of ~75k vertices, K = 50, radius 0.006 (the coarse stage's splat): silhouettes in
    [0,1], the two garments' images never add up to more than 1, gradients reach the points and are finite, and the
    forward is reproducible bit for bit."""
    from recmv import raster
    upper, _ = _sphere_mesh(129, radius=0.45)
    lower, _ = _sphere_mesh(129, radius=0.40)
    lower = lower + torch.tensor([0.0, -0.25, 0.0], device=DEV)
    H = W = 512
    cam = _camera(H, W)
    offs = torch.tensor([[0., 0., 0.], [0.07, -0.03, 0.2], [-0.1, 0.05, -0.1]], device=DEV)
    cloud = (torch.cat([upper, lower])[None] + offs[:, None]).requires_grad_(True)
    rend = raster.PointsRendererWithFrags_Split(cam, (H, W), radius=0.006, points_per_pixel=50)
    imgs, frags = rend(cloud, split_size=upper.shape[0])
    assert imgs[0].shape == (3, H, W, 1) and frags.idx.shape == (3, H, W, 50)
    for im in imgs:
        assert im.min() >= 0 and im.max() <= 1 + 1e-5
    assert (imgs[0] + imgs[1]).max() <= 1 + 1e-5
    cover = ((imgs[0] + imgs[1])[..., 0] > 0.5).float().mean()
    assert 0.1 < cover < 0.6
    # lists are packed, sorted by depth, and every listed point is within the radius
    valid = frags.idx >= 0
    assert (valid[..., 1:] <= valid[..., :-1]).all()
    z = torch.where(valid, frags.zbuf, torch.full_like(frags.zbuf, float("inf")))
    assert (z[..., 1:] >= z[..., :-1]).all()
    assert (frags.dists[valid] < 0.006 ** 2).all() and (frags.dists[valid] >= 0).all()
    gt = torch.zeros(3, H, W, device=DEV)
    gt[:, 100:400, 150:380] = 1
    m = imgs[0][..., -1]
    loss = (1. - (m * gt).view(3, -1).sum(1) / (m + gt - m * gt).abs().view(3, -1).sum(1)).mean()   # :626
    loss.backward()
    assert torch.isfinite(cloud.grad).all() and cloud.grad.abs().sum() > 0
    again, frags2 = rend(cloud.detach(), split_size=upper.shape[0])
    assert torch.equal(frags2.idx, frags.idx) and torch.equal(again[0], imgs[0].detach())
    # the whole forward + backward is reproducible bit for bit (no float atomics on the way to the points: the explicit
    # vertices that the mask loss moves — and with them every later re-mesh — are the same from run to run)
    g1 = cloud.grad.clone()
    cloud.grad = None
    imgs2, _ = rend(cloud, split_size=upper.shape[0])
    m2 = imgs2[0][..., -1]
    (1. - (m2 * gt).view(3, -1).sum(1) / (m2 + gt - m2 * gt).abs().view(3, -1).sum(1)).mean().backward()
    assert torch.equal(cloud.grad, g1)


def test_fixture_fragments_and_reference_surface_points_on_device():
    """Golden path (tests/golden/findsurface.npz, made with the reference's own FindSurfacePs): the HIP rasteriser gives
    the fixture's fragments bit for bit, and FindSurfacePs on the device gives the reference's indices."""
    from pathlib import Path
    from recmv import raster, utils
    g = {k: torch.from_numpy(v) for k, v in np.load(Path(__file__).parent / "golden" / "findsurface.npz").items()}
    F = g["faces"].shape[0]
    H, W = int(g["H"]), int(g["W"])
    frags = raster.rasterize_meshes(g["fv"].to(DEV), torch.tensor([0, F], device=DEV), torch.tensor([F, F], device=DEV),
                                    (H, W), max_faces_per_mesh=F)
    assert torch.equal(frags.pix_to_face.cpu(), g["pix_to_face"])
    assert torch.equal(frags.bary_coords.cpu().view(torch.int32), g["bary"].view(torch.int32))
    b, r, c, p0, f = utils.FindSurfacePs(g["verts"].to(DEV), g["faces"].to(DEV), frags)
    for got, key in ((b, "batch"), (r, "row"), (c, "col"), (f, "finds")):
        assert torch.equal(got.cpu(), g[key]), key
    assert torch.allclose(p0.cpu(), g["init"], rtol=0, atol=1e-6)
    fr3 = raster.Fragments(g["pix_to_face3"].to(DEV), None, g["bary3"].to(DEV), None)
    b, r, c, p0, f = utils.FindSurfacePs(g["verts"].to(DEV), g["faces"].to(DEV), fr3)
    for got, key in ((b, "batch3"), (r, "row3"), (c, "col3"), (f, "finds3")):
        assert torch.equal(got.cpu(), g[key]), key
