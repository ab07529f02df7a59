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
