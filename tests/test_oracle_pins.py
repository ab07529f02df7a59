"""Pin the CPU oracle against everything the reference itself pins.

  * FastMinv/check.py:7-20           -> inv*m == I on randn(10000,3,3)
  * MCAcc/check_grid_sampler_mine.py -> equality with F.grid_sample(border, align_corners=False),
                                        gradcheck of the Function and of the backward Function (f64)
  * MCAcc/seg3d_lossless.py:273-282  -> interp2x == F.interpolate(trilinear, align_corners=True) + (0<valid<1)
"""
import pytest
import torch
import torch.nn.functional as F


def test_inv3x3_identity_like_reference_check(oracle):
    torch.manual_seed(0)
    N = 10000  # FastMinv/check.py:7
    ms = torch.randn(N, 3, 3)
    invs, checks = oracle.inv3x3_forward(ms)
    assert checks.dtype == torch.bool and invs.shape == (N, 3, 3)
    err = (invs[checks].matmul(ms[checks]) - torch.eye(3).view(1, 3, 3)).norm(dim=(1, 2))
    # det >= 1e-4 bounds the conditioning; f32 cofactor inverse
    assert err.mean().item() < 1e-4 and err.max().item() < 0.5
    det = torch.linalg.det(ms.double())
    clear = (det.abs() - 1e-4).abs() > 1e-6
    assert torch.equal(checks[clear], (det.abs() >= 1e-4)[clear])
    assert (invs[~checks] == 0).all()


def test_inv3x3_f64_vs_linalg_and_backward(oracle):
    torch.manual_seed(1)
    ms = torch.randn(2000, 3, 3, dtype=torch.float64)
    invs, checks = oracle.inv3x3_forward(ms)
    ref = torch.linalg.inv(ms[checks])
    torch.testing.assert_close(invs[checks], ref, rtol=1e-9, atol=1e-9)
    g = torch.randn_like(ms)
    out = oracle.inv3x3_backward(g, invs)
    m = ms[checks].clone().requires_grad_(True)
    (torch.linalg.inv(m) * g[checks]).sum().backward()
    torch.testing.assert_close(out[checks], m.grad, rtol=1e-8, atol=1e-8)
    assert (out[~checks] == 0).all()  # singular rows: inverse 0 -> gradient 0


def test_inv3x3_singular_rows(oracle):
    ms = torch.eye(3).repeat(4, 1, 1)
    ms[0] *= 0.0                     # det = 0
    ms[1, 0, 0] = 0.99e-4            # det just below threshold
    ms[2, 0, 0] = 1.01e-4            # just above
    ms[3, 0, 0] = -0.5e-4            # negative, |det| below
    invs, checks = oracle.inv3x3_forward(ms)
    assert checks.tolist() == [False, False, True, False]
    assert (invs[0] == 0).all() and (invs[1] == 0).all() and (invs[3] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_sampler_equals_torch_grid_sample(oracle, dtype):
    torch.manual_seed(0)
    inp = torch.randn(1, 5, 15, 15, 15, dtype=dtype)                       # check_grid_sampler_mine.py:5
    grid = (torch.rand(1, 1, 1, 10, 3, dtype=dtype) - 0.5) * 2.2            # :6  (~9 % clipped)
    out = oracle.gs3d_forward(inp, grid)
    ref = F.grid_sample(inp, grid, mode="bilinear", padding_mode="border", align_corners=False)
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    torch.testing.assert_close(out, ref, rtol=tol, atol=tol)
    # first derivative against torch's own backward
    inp2 = inp.clone().requires_grad_(True)
    grid2 = grid.clone().requires_grad_(True)
    go = torch.randn_like(ref)
    F.grid_sample(inp2, grid2, mode="bilinear", padding_mode="border", align_corners=False).backward(go)
    gi, gg = oracle.gs3d_backward(inp, grid, go)
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    torch.testing.assert_close(gi, inp2.grad, rtol=tol, atol=tol)
    torch.testing.assert_close(gg, grid2.grad, rtol=tol, atol=tol)


def test_sampler_strided_and_multi_batch(oracle):
    torch.manual_seed(3)
    inp = torch.randn(2, 6, 5, 7, 9, dtype=torch.float64)
    inp_cl = inp.contiguous(memory_format=torch.channels_last_3d)
    grid = (torch.rand(2, 2, 3, 4, 3, dtype=torch.float64) - 0.5) * 2.4
    ref = F.grid_sample(inp, grid, mode="bilinear", padding_mode="border", align_corners=False)
    torch.testing.assert_close(oracle.gs3d_forward(inp, grid), ref, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(oracle.gs3d_forward(inp_cl, grid), ref, rtol=1e-12, atol=1e-12)


def test_sampler_gradcheck_and_double_backward(oracle):
    """MCAcc/check_grid_sampler_mine.py:5-16 (gradcheck of the Function and of its backward Function, 10 points, coordinates up
    to +-1.1) on a 6 x 7 x 5 volume — finite differences over the reference's 15^3 volume take two minutes and show nothing more;
    that exact shape is compared bit for bit with the reference's kernels in tests/test_oracle_ref.py."""
    torch.manual_seed(0)
    inp = torch.randn(1, 5, 6, 7, 5, dtype=torch.double, requires_grad=True)
    grid = ((torch.rand(1, 1, 1, 10, 3, dtype=torch.double) - 0.5) * 2.2).requires_grad_(True)
    assert torch.autograd.gradcheck(oracle.OracleGridSample3dFunction.apply, (inp, grid))        # :11
    go = torch.randn(1, 5, 1, 1, 10, dtype=torch.double, requires_grad=True)
    assert torch.autograd.gradcheck(oracle.OracleGridSample3dBackwardFunction.apply, (inp, grid, go))  # :16


def test_sampler_clip_edges(oracle):
    # exactly on / beyond the border: gradient mask must be 0 (GridSamplerMineKernel.cu:47-58)
    inp = torch.arange(2 * 3 * 4 * 5, dtype=torch.float64).view(1, 2, 3, 4, 5)
    grid = torch.tensor([[-1.5, 0.0, 0.0], [1.5, 0.3, -0.2], [0.1, -2.0, 2.0], [-1.0, -1.0, -1.0],
                         [1.0, 1.0, 1.0]], dtype=torch.float64).view(1, 1, 1, 5, 3)
    go = torch.ones(1, 2, 1, 1, 5, dtype=torch.float64)
    _, gg = oracle.gs3d_backward(inp, grid, go)
    gg = gg.view(5, 3)
    assert gg[0, 0] == 0 and gg[1, 0] == 0 and gg[2, 1] == 0 and gg[2, 2] == 0
    assert (gg[3] == 0).all() and (gg[4] == 0).all()
    ref = F.grid_sample(inp, grid, mode="bilinear", padding_mode="border", align_corners=False)
    torch.testing.assert_close(oracle.gs3d_forward(inp, grid), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_interp2x_equals_interpolate(oracle, dtype):
    torch.manual_seed(0)
    x = torch.randn(1, 1, 6, 9, 5, dtype=dtype)
    bal = 0.1
    out, bnd = oracle.interp2x_forward(x, bal)
    D, H, W = 11, 17, 9
    ref = F.interpolate(x, size=(D, H, W), mode="trilinear", align_corners=True)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    valid = F.interpolate((x > bal).to(dtype), size=(D, H, W), mode="trilinear", align_corners=True)
    assert torch.equal(bnd, (valid > 0.0) & (valid < 1.0))
    # backward == autograd of interpolate
    go = torch.randn_like(ref)
    xx = x.clone().requires_grad_(True)
    F.interpolate(xx, size=(D, H, W), mode="trilinear", align_corners=True).backward(go)
    gi = oracle.interp2x_backward(go)
    torch.testing.assert_close(gi, xx.grad, rtol=1e-5, atol=1e-5)


def test_def_regu_oracle_against_the_reference_gm_output_and_finite_differences(oracle):
    """oracle.def_regu (the checker of recmv_def_regu) pinned: (i) its Geman-McClure stage against the REFERENCE's own
    utils.GMRobustError output stored in tests/golden/misc.npz (J = diag(e^sqrt(x), 1, 1) has sum log^2 sigma = x exactly),
    (ii) its gradient against central differences of its own value in float64."""
    import numpy as np
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "misc.npz")
    x = torch.from_numpy(g["gm_x"]).double().abs().reshape(-1)
    J = torch.zeros(x.numel(), 3, 3, dtype=torch.float64)
    J[:, 0, 0] = torch.exp(torch.sqrt(x))
    J[:, 1, 1] = 1.0
    J[:, 2, 2] = 1.0
    y, _ = oracle.def_regu(J, 0.01)
    ref = torch.from_numpy(g["gm_true"]).double().reshape(-1)
    keep = torch.from_numpy(g["gm_x"]).double().reshape(-1) >= 0          # (the fixture's x may be signed: GM(x, square) is not even)
    torch.testing.assert_close(y[keep], ref[keep], rtol=1e-6, atol=1e-9)
    gen = torch.Generator().manual_seed(3)
    A = torch.eye(3, dtype=torch.float64) + 0.2 * torch.randn(5, 3, 3, generator=gen, dtype=torch.float64)
    _, grad = oracle.def_regu(A, 0.3)
    h = 1e-6
    for i in range(3):
        for j in range(3):
            Ap, Am = A.clone(), A.clone()
            Ap[:, i, j] += h
            Am[:, i, j] -= h
            fd = (oracle.def_regu(Ap, 0.3)[0] - oracle.def_regu(Am, 0.3)[0]) / (2 * h)
            torch.testing.assert_close(grad[:, i, j], fd, rtol=1e-5, atol=1e-8)
