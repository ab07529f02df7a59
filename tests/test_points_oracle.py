"""CPU checks of the point-rasteriser / alpha-compositor restatement in oracle/recmv_oracle.c.

pytorch3d is not available here, so the oracle is PARITY UNPINNED against it.  Checked on its own:
a float64 numpy brute force of "K nearest in depth within the radius", the port variant, the analytic gradients
against torch autograd of a plain-torch restatement, and the fill values."""
import numpy as np
import torch


def _cloud(rng, n, zlo=0.5, zhi=3.0):
    xy = rng.uniform(-1.05, 1.05, size=(n, 2))
    z = rng.uniform(zlo, zhi, size=(n, 1))
    return torch.from_numpy(np.concatenate([xy, z], 1).astype(np.float32))


def test_points_oracle_matches_numpy_k_nearest(oracle):
    rng = np.random.default_rng(0)
    pts = _cloud(rng, 900, zlo=-0.2)
    H, W, K, r = 20, 24, 5, 0.13
    first, num = torch.tensor([0, 500]), torch.tensor([500, 400])
    idx, zbuf, dists = oracle.rasterize_points(pts, first, num, (H, W), r, K)
    assert idx.dtype == torch.int32 and idx.shape == (2, H, W, K)
    p64 = pts.double().numpy()
    full = 0
    for n, (f0, cnt) in enumerate(((0, 500), (500, 400))):
        for row in range(H):
            for col in range(W):
                xf, yf = 1 - (2 * col + 1) / W, 1 - (2 * row + 1) / H
                sub = p64[f0:f0 + cnt]
                d2 = (sub[:, 0] - xf) ** 2 + (sub[:, 1] - yf) ** 2
                ok = np.nonzero((d2 < r * r - 1e-7) & (sub[:, 2] >= 0))[0]
                order = ok[np.argsort(sub[ok, 2], kind="stable")][:K] + f0
                got = idx[n, row, col].numpy()
                got = got[got >= 0]
                # borderline points (|d2 - r^2| < 1e-7) may differ; none with this seed
                assert list(got) == list(order), (n, row, col)
                full += len(order) == K
                assert np.allclose(dists[n, row, col, :len(order)].numpy(), d2[order - f0], atol=1e-6)
                assert np.allclose(zbuf[n, row, col, :len(order)].numpy(), p64[order, 2], atol=0)
                assert (zbuf[n, row, col, len(order):] == -1).all() and (dists[n, row, col, len(order):] == -1).all()
    assert full > 20, "some pixels must saturate K so that the eviction path is exercised"


def test_points_scan_port_equals_per_pixel_loop(oracle):
    rng = np.random.default_rng(3)
    pts = _cloud(rng, 3000, zlo=-0.1)
    pts[7] = pts[6]                                                   # equal depth: index decides
    first, num = torch.tensor([0, 1000, 1000]), torch.tensor([1000, 0, 2000])
    for (H, W, K, r) in ((33, 29, 4, 0.09), (16, 40, 50, 0.2)):
        a = oracle.rasterize_points(pts, first, num, (H, W), r, K)
        b = oracle.rasterize_points(pts, first, num, (H, W), r, K, scan=True)
        assert torch.equal(a[0], b[0])
        assert torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))
        assert torch.equal(a[2].view(torch.int32), b[2].view(torch.int32))
        assert (a[0][1] == -1).all() and (a[0][0] >= 0).any()


def _torch_composite(idx, alphas, features):
    """Plain-torch alpha compositing in the dtype of `alphas` (autograd gives the reference gradients)."""
    N, H, W, K = idx.shape
    valid = (idx >= 0)
    a = torch.where(valid, alphas, torch.zeros_like(alphas))
    cum = torch.cumprod(torch.cat([torch.ones_like(a[..., :1]), 1 - a[..., :-1]], -1), -1)       # prod_{l<k}(1-a_l)
    f = features[:, idx.clamp(min=0).long()]                                                      # [C,N,H,W,K]
    return (f * (cum * a)[None]).sum(-1).permute(1, 0, 2, 3)                                      # [N,C,H,W]


def test_alpha_composite_and_gradients(oracle):
    rng = np.random.default_rng(5)
    pts = _cloud(rng, 1500)
    H, W, K, r = 18, 22, 6, 0.16
    first, num = torch.tensor([0, 700]), torch.tensor([700, 800])
    idx, zbuf, dists = oracle.rasterize_points(pts, first, num, (H, W), r, K)
    alphas = (1 - dists / (r * r)) * (idx >= 0)
    features = torch.from_numpy(rng.uniform(0, 1, size=(3, 1500)).astype(np.float32))
    img = oracle.alpha_composite_forward(idx, alphas, features)
    a64 = alphas.double().requires_grad_(True)
    f64 = features.double().requires_grad_(True)
    ref = _torch_composite(idx, a64, f64)
    assert torch.allclose(img.double(), ref, atol=1e-6)
    g = torch.from_numpy(rng.normal(size=tuple(img.shape)).astype(np.float32))
    ref.backward(g.double())
    ga, gf = oracle.alpha_composite_backward(idx, alphas, features, g)
    valid = idx >= 0
    assert torch.allclose(ga[valid].double(), a64.grad[valid], rtol=1e-4, atol=1e-5)
    assert torch.allclose(gf.double(), f64.grad, rtol=1e-4, atol=1e-5)
    assert (ga[~valid] == 0).all()
    # d dists / d points
    gd = torch.from_numpy(rng.normal(size=tuple(dists.shape)).astype(np.float32)) * valid
    gz = torch.from_numpy(rng.normal(size=tuple(dists.shape)).astype(np.float32)) * valid
    gp = oracle.rasterize_points_backward(pts, idx, gd, gz)
    p64 = pts.double().requires_grad_(True)
    col = torch.arange(W).view(1, 1, W, 1).double()
    row = torch.arange(H).view(1, H, 1, 1).double()
    xf, yf = 1 - (2 * col + 1) / W, 1 - (2 * row + 1) / H
    sel = p64[idx.clamp(min=0).long()]                                                            # [N,H,W,K,3]
    d2 = (xf - sel[..., 0]) ** 2 + (yf - sel[..., 1]) ** 2
    ((d2 * gd.double() + sel[..., 2] * gz.double()) * valid).sum().backward()
    assert torch.allclose(gp.double(), p64.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(d2[valid].float(), dists[valid], atol=1e-6)
