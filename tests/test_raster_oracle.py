"""CPU checks of the rasteriser restatement in oracle/recmv_oracle.c (`oracle_rasterize_meshes`).

pytorch3d is not available in this tree, so the oracle is PARITY UNPINNED against it; what can be
checked on its own is checked here: an independent float64 numpy restatement of the coverage / nearest-face rule,
the geometric meaning of the outputs, the documented tie rule and the fill values.
"""
import numpy as np
import torch


def _random_soup(rng, n_faces, lo=-1.1, hi=1.1, size=0.3, zlo=0.5, zhi=3.0):
    c = rng.uniform(lo, hi, size=(n_faces, 1, 2))
    xy = c + rng.uniform(-size, size, size=(n_faces, 3, 2))
    z = rng.uniform(zlo, zhi, size=(n_faces, 3, 1))
    return torch.from_numpy(np.concatenate([xy, z], -1).astype(np.float32))


def _numpy_first_hit(fv, H, W):
    """Nearest strictly-covering face per pixel centre in float64 (perspective-correct depth)."""
    fv = fv.double().numpy()
    out = -np.ones((H, W), dtype=np.int64)
    zb = np.full((H, W), np.inf)
    for r in range(H):
        for c in range(W):
            px, py = 1 - (2 * c + 1) / W, 1 - (2 * r + 1) / H
            for f, (a, b, cc) in enumerate(fv):
                def e(p, q, s):
                    return (p[0] - q[0]) * (s[1] - q[1]) - (p[1] - q[1]) * (s[0] - q[0])
                area = e(cc, a, b)
                if abs(area) < 1e-7:
                    continue
                w = np.array([e((px, py), b, cc), e((px, py), cc, a), e((px, py), a, b)]) / area
                if (w <= 1e-6).any():
                    continue
                top = w * np.array([b[2] * cc[2], a[2] * cc[2], a[2] * b[2]])
                bz = top / top.sum()
                z = (bz * np.array([a[2], b[2], cc[2]])).sum()
                if z < zb[r, c] - 1e-6:
                    zb[r, c], out[r, c] = z, f
    return out, zb


def test_oracle_matches_numpy_first_hit(oracle):
    rng = np.random.default_rng(0)
    fv = _random_soup(rng, 40)
    H, W = 24, 20
    first, num = torch.tensor([0]), torch.tensor([40])
    p2f, zbuf, bary, dists = oracle.rasterize_meshes(fv, first, num, (H, W))
    ref, zref = _numpy_first_hit(fv, H, W)
    got = p2f[0, :, :, 0].numpy()
    # pixels whose centre is within rounding of an edge or whose two nearest faces tie may differ: none with this seed
    assert (got == ref).mean() > 0.995
    hit = (got >= 0) & (got == ref)
    assert np.allclose(zbuf[0, :, :, 0].numpy()[hit], zref[hit], rtol=1e-5)
    assert (p2f[p2f >= 0] < 40).all() and hit.sum() > 50


def test_oracle_outputs_mean_what_they_say(oracle):
    rng = np.random.default_rng(1)
    fv = _random_soup(rng, 60)
    H, W = 32, 32
    p2f, zbuf, bary, dists = oracle.rasterize_meshes(fv, torch.tensor([0, 25]), torch.tensor([25, 35]), (H, W))
    assert p2f.shape == (2, H, W, 1) and bary.shape == (2, H, W, 1, 3)
    empty = p2f < 0
    assert (zbuf[empty] == -1).all() and (dists[empty] == -1).all() and (bary[empty] == -1).all()
    # mesh 0 only sees faces [0,25), mesh 1 only [25,60)
    assert (p2f[0][p2f[0] >= 0] < 25).all() and (p2f[1][p2f[1] >= 0] >= 25).all()
    hit = ~empty
    b = bary[hit]                                              # [M,3], perspective-corrected
    assert torch.allclose(b.sum(-1), torch.ones(b.shape[0]), atol=1e-5)
    assert (dists[hit] <= 0).all()                             # blur_radius = 0: only interior pixels
    tri = fv[p2f[hit]]                                         # [M,3,3]
    z = (b * tri[:, :, 2]).sum(-1)
    assert torch.allclose(z, zbuf[hit], rtol=1e-6)
    # un-correcting the barycentrics recovers the pixel centre: b_i ~ w_i / z_i, so b_i z_i normalised are the
    # screen-space weights
    w = b * tri[:, :, 2]
    w = w / w.sum(-1, keepdim=True)
    xy = (w[:, :, None] * tri[:, :, :2]).sum(1)
    n, r, c, _ = hit.nonzero(as_tuple=True)
    centre = torch.stack([1 - (2 * c.float() + 1) / W, 1 - (2 * r.float() + 1) / H], -1)
    assert torch.allclose(xy, centre, atol=2e-5)


def test_oracle_tie_rule_rejections_and_blur(oracle):
    tri = torch.tensor([[[-0.9, -0.9, 1.0], [0.9, -0.9, 1.0], [0.0, 0.9, 1.0]]])
    behind = tri.clone()
    behind[..., 2] = -1.0
    flat = torch.tensor([[[-0.5, 0.0, 0.5], [0.0, 0.0, 0.5], [0.5, 0.0, 0.5]]])          # zero area, nearer
    fv = torch.cat([behind, flat, tri, tri])                                             # duplicates: 2 and 3
    p2f, zbuf, bary, dists = oracle.rasterize_meshes(fv, torch.tensor([0]), torch.tensor([4]), (16, 16))
    hit = p2f[p2f >= 0]
    assert hit.numel() > 30 and (hit == 2).all(), "equal depth -> lowest face index; behind / zero-area never win"
    # back-face culling drops clockwise faces (negative edge function of v0,v1,v2)
    cw = tri[:, [0, 2, 1]]
    sign_keep = oracle.rasterize_meshes(tri, torch.tensor([0]), torch.tensor([1]), (16, 16), cull_backfaces=True)[0]
    sign_drop = oracle.rasterize_meshes(cw, torch.tensor([0]), torch.tensor([1]), (16, 16), cull_backfaces=True)[0]
    assert ((sign_keep >= 0).sum() == 0) != ((sign_drop >= 0).sum() == 0)
    # a blur radius only ever adds pixels, with positive distance outside the face
    hard = oracle.rasterize_meshes(tri, torch.tensor([0]), torch.tensor([1]), (16, 16))
    soft = oracle.rasterize_meshes(tri, torch.tensor([0]), torch.tensor([1]), (16, 16), blur_radius=0.02)
    assert ((hard[0] >= 0) <= (soft[0] >= 0)).all() and (soft[0] >= 0).sum() > (hard[0] >= 0).sum()
    extra = (soft[0] >= 0) & (hard[0] < 0)
    assert (soft[3][extra] > 0).all() and (soft[3][extra] < 0.02).all()
    # no mesh / empty mesh
    e = oracle.rasterize_meshes(tri, torch.tensor([0, 1]), torch.tensor([1, 0]), (8, 8))
    assert (e[0][1] == -1).all() and (e[0][0] >= 0).any()


def test_scan_port_equals_per_pixel_loop(oracle):
    """The face-ordered CPU port used for `cpu_baseline` gives bit-identical fragments."""
    rng = np.random.default_rng(7)
    for size, blur, hw in ((0.05, 0.0, (40, 56)), (0.6, 0.0, (33, 31)), (0.2, 4e-3, (24, 24))):
        fv = _random_soup(rng, 150, size=size, zlo=-0.3)
        fv[5] = fv[4]
        first, num = torch.tensor([0, 60, 60]), torch.tensor([60, 0, 90])
        a = oracle.rasterize_meshes(fv, first, num, hw, blur_radius=blur)
        b = oracle.rasterize_meshes(fv, first, num, hw, blur_radius=blur, scan=True)
        for x, y in zip(a, b):
            assert torch.equal(x, y) if x.dtype == torch.int64 else torch.equal(x.view(torch.int32), y.view(torch.int32))
        assert (a[0] >= 0).sum() > 20


def test_find_surface_ps_single_fragment_path(oracle):
    """utils.FindSurfacePs (utils/FindSurfacePs.py:7-37): the K = 1 shortcut returns what the general scatter-min
    formulation returns when a second, empty fragment layer is appended."""
    from recmv import raster, utils
    rng = np.random.default_rng(11)
    F = 120
    verts = torch.from_numpy(rng.normal(size=(3 * F, 3)).astype(np.float32))
    faces = torch.arange(3 * F).view(F, 3)
    fv = _random_soup(rng, 2 * F, size=0.25)
    p2f, zbuf, bary, dists = oracle.rasterize_meshes(fv, torch.tensor([0, F]), torch.tensor([F, F]), (40, 48))
    one = utils.FindSurfacePs(verts, faces, raster.Fragments(p2f, zbuf, bary, dists))
    pad = lambda t: torch.cat([t, torch.full_like(t, -1)], dim=3)
    two = utils.FindSurfacePs(verts, faces, raster.Fragments(pad(p2f), pad(zbuf), pad(bary), pad(dists)))
    assert one[0].numel() > 100
    for a, b in zip(one, two):
        assert torch.equal(a, b)
    assert (one[4] < F).all() and (one[4] >= 0).all()
