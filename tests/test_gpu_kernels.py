"""HIP kernels vs the CPU oracle, through the C ABI (python -m pytest tests -m gpu).

Bars: bit-exact for indices/masks and for every kernel whose arithmetic is a fixed IEEE sequence shared
with the oracle (inv3x3, sampler forward/backward, interp2x, MC vertices+faces); stated f32 tolerances
for the MFMA contractions (different summation order than a sequential reference).
"""
import os
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPO = Path(__file__).resolve().parent.parent


def gpu(t):
    return t.to(DEV)


# ------------------------------------------------------------------------------------------ inv3x3
@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 10000, 35937])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_inv3x3_bit_exact(oracle, n, dtype):
    from recmv import FastMinv
    torch.manual_seed(n + 1)
    ms = torch.randn(n, 3, 3, dtype=dtype)
    if n > 8:  # adversarial rows: singular, and |det| straddling 1e-4
        ms[0] = 0
        ms[1] = torch.eye(3, dtype=dtype)
        ms[1, 0, 0] = 0.99e-4
        ms[2] = torch.eye(3, dtype=dtype)
        ms[2, 0, 0] = 1.01e-4
        ms[3] = torch.ones(3, 3, dtype=dtype)
    inv_o, chk_o = oracle.inv3x3_forward(ms)
    inv_g, chk_g = FastMinv.Fast3x3Minv(gpu(ms))
    assert chk_g.dtype == torch.bool and inv_g.shape == (n, 3, 3)
    assert torch.equal(chk_g.cpu(), chk_o)
    assert torch.equal(inv_g.cpu(), inv_o), "inverse must be bit-identical to the oracle"
    g = torch.randn(n, 3, 3, dtype=dtype)
    out_o = oracle.inv3x3_backward(g, inv_o)
    out_g = FastMinv.Fast3x3Minv_backward(gpu(g), inv_g)
    assert torch.equal(out_g.cpu(), out_o)


def test_inv3x3_identity_check_like_reference():
    from recmv import FastMinv
    torch.manual_seed(0)
    ms = torch.randn(10000, 3, 3, device=DEV)           # FastMinv/check.py:7-8
    invs, checks = FastMinv.Fast3x3Minv(ms)
    err = (invs[checks].matmul(ms[checks]) - torch.eye(3, device=DEV).view(1, 3, 3)).norm(dim=(1, 2))
    assert err.mean().item() < 1e-4


def test_inv3x3_argument_errors():
    from recmv import FastMinv
    with pytest.raises(RuntimeError):
        FastMinv.Fast3x3Minv(torch.randn(4, 3, 3, device=DEV).half())
    with pytest.raises(RuntimeError):
        FastMinv.Fast3x3Minv(torch.randn(4, 3, 6, device=DEV)[:, :, ::2])      # not contiguous
    with pytest.raises(RuntimeError):
        FastMinv.Fast3x3Minv_backward(torch.randn(4, 3, 3, device=DEV), torch.randn(4, 3, 3, device=DEV).double())


def test_fastdiff_function_gradcheck():
    from recmv.utils import FastDiff3x3MinvFunction
    torch.manual_seed(0)
    m = (torch.randn(6, 3, 3, dtype=torch.double, device=DEV) + 2 * torch.eye(3, dtype=torch.double, device=DEV)
         ).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda x: FastDiff3x3MinvFunction.apply(x)[0], (m,))


# ------------------------------------------------------------------------------------------ sampler
def _sampler_case(dtype, C=5, dims=(15, 15, 15), P=10, channels_last=False, seed=0, spread=2.2):
    g = torch.Generator().manual_seed(seed)
    inp = torch.randn(1, C, *dims, dtype=dtype, generator=g)
    if channels_last:
        inp = inp.contiguous(memory_format=torch.channels_last_3d)
    grid = (torch.rand(1, 1, 1, P, 3, dtype=dtype, generator=g) - 0.5) * spread
    return inp, grid


@pytest.mark.parametrize("dtype,cl,C", [(torch.float64, False, 5), (torch.float32, False, 5),
                                         (torch.float32, True, 24), (torch.float32, True, 8),
                                         (torch.float32, False, 24), (torch.float64, True, 4)])
def test_sampler_forward_backward_dbackward_vs_oracle(oracle, dtype, cl, C):
    from recmv import GridSamplerMine
    inp, grid = _sampler_case(dtype, C=C, dims=(9, 13, 11), P=4099, channels_last=cl, seed=C)
    out_o = oracle.gs3d_forward(inp, grid)
    out_g = GridSamplerMine.forward(gpu(inp), gpu(grid), 0, 1)
    assert torch.equal(out_g.cpu(), out_o), "forward is a fixed fma chain: bit-exact"
    go = torch.randn(out_o.shape, dtype=dtype, generator=torch.Generator().manual_seed(1))
    gi_o, gg_o = oracle.gs3d_backward(inp, grid, go)
    with GridSamplerMine.exact_order():         # the reference's channel order (the default for grad_grid-only requests is not)
        _sampler_exact_backward_checks(oracle, dtype, cl, inp, grid, go, gi_o, gg_o)


def _sampler_exact_backward_checks(oracle, dtype, cl, inp, grid, go, gi_o, gg_o):
    from recmv import GridSamplerMine
    gi_g, gg_g = GridSamplerMine.backward(gpu(inp), gpu(grid), gpu(go), 0, 1)
    assert torch.equal(gg_g.cpu(), gg_o), "grad_grid: bit-exact"
    tol = 1e-12 if dtype == torch.float64 else 2e-5                     # atomics: order differs
    torch.testing.assert_close(gi_g.cpu(), gi_o, rtol=tol, atol=tol)
    gi_none, gg2 = GridSamplerMine.backward(gpu(inp), gpu(grid), gpu(go), 0, 1, need_grad_input=False)
    assert gi_none is None and torch.equal(gg2, gg_g)
    # double backward
    ggI = torch.randn(inp.shape, dtype=dtype, generator=torch.Generator().manual_seed(2))
    if cl:
        ggI = ggI.contiguous(memory_format=torch.channels_last_3d)
    ggG = torch.randn(grid.shape, dtype=dtype, generator=torch.Generator().manual_seed(3))
    a_o, b_o, c_o = oracle.gs3d_dbackward(ggI, ggG, inp, grid, go)
    a_g, b_g, c_g = GridSamplerMine.dbackward(gpu(ggI), gpu(ggG), gpu(inp), gpu(grid), gpu(go), 0, 1)
    assert torch.equal(b_g.cpu(), b_o) and torch.equal(c_g.cpu(), c_o)
    torch.testing.assert_close(a_g.cpu(), a_o, rtol=tol, atol=tol)
    # ggI = None (the hot path: volume is a frozen buffer)
    a2_o, b2_o, c2_o = oracle.gs3d_dbackward(None, ggG, inp, grid, go, need_grad_input=False)
    a2_g, b2_g, c2_g = GridSamplerMine.dbackward(None, gpu(ggG), gpu(inp), gpu(grid), gpu(go), 0, 1,
                                                 need_grad_input=False)
    assert a2_g is None and torch.equal(b2_g.cpu(), b2_o) and torch.equal(c2_g.cpu(), c2_o)


def test_sampler_exact_order_around_the_forward_call_governs_backward_and_double_backward(oracle):
    """`with GridSamplerMine.exact_order():` around the FORWARD of the autograd Function is enough: the mode is recorded in the
    context and applied when autograd launches backward / double backward later, outside the block (the switch itself is a process
    global).  Channels-last skinning-like volume with a frozen buffer, grad_grid only: the record-coalesced default differs from the
    oracle in the last bits, the exact order is bit-equal."""
    from recmv import GridSamplerMine
    from recmv.MCAcc.grid_sampler_mine import GridSamplerMine3dFunction
    inp, grid = _sampler_case(torch.float32, C=24, dims=(9, 13, 11), P=4099, channels_last=True, seed=7)
    go = torch.randn(1, 24, 1, 1, 4099, generator=torch.Generator().manual_seed(1))
    _, gg_o = oracle.gs3d_backward(inp, grid, go)
    ggG = torch.randn(grid.shape, generator=torch.Generator().manual_seed(3))
    _, b_o, c_o = oracle.gs3d_dbackward(None, ggG, inp, grid, go, need_grad_input=False)
    vol = gpu(inp)
    results = {}
    for exact in (True, False):
        g = gpu(grid).requires_grad_(True)
        go_g = gpu(go).requires_grad_(True)
        with GridSamplerMine.exact_order(exact):
            out = GridSamplerMine3dFunction.apply(vol, g)
        assert GridSamplerMine.current_mode() == 0                      # outside the block: the default again
        (gg,) = torch.autograd.grad(out, g, go_g, create_graph=True)    # backward Function, launched outside the block
        b, c = torch.autograd.grad(gg, (g, go_g), gpu(ggG))             # double backward
        results[exact] = (gg.detach().cpu(), b.cpu(), c.cpu())
    assert all(torch.equal(x, y) for x, y in zip(results[True], (gg_o, b_o, c_o)))
    assert not all(torch.equal(x, y) for x, y in zip(results[False], (gg_o, b_o, c_o))), "the default is the other lane order"
    for x, y in zip(results[False], (gg_o, b_o, c_o)):
        assert float((x - y).abs().max()) <= 2e-6 * float(y.abs().max())


@pytest.mark.parametrize("C", [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 20])
def test_sampler_backward_record_lanes_within_tolerance(oracle, C):
    """The DEFAULT backward / double backward for grad_grid-only requests on a channels-last f32 volume (what the loop asks: the
    skinning volume is a frozen buffer): G lanes per point x V channels per lane, per-point sums finished by a lane butterfly —
    the reference's terms (GridSamplerMineKernel.cu:333-570, 575-914) in another order.  Bound: 2e-6 of the largest entry of each
    tensor (north_star: gradients within an f32 tolerance); the exact-order kernels stay bit-equal to the oracle
    (test_sampler_forward_backward_dbackward_vs_oracle).  Ragged last block, several batch items, points outside the volume, on
    the far border, and non-finite coordinates (every corner skipped: zeros); C = 20 has no lane split and falls back."""
    from recmv import GridSamplerMine
    g = torch.Generator().manual_seed(100 + C)
    N, P = 3, 4099 + 32 * 11 + 3
    inp = torch.randn(N, C, 9, 13, 11, generator=g).contiguous(memory_format=torch.channels_last_3d)
    grid = (torch.rand(N, 1, 1, P, 3, generator=g) - 0.5) * 2.6
    grid[0, 0, 0, 5] = torch.tensor([float('nan'), 0.1, 0.2])
    grid[1, 0, 0, 7] = torch.tensor([0.3, float('inf'), -0.2])
    grid[2, 0, 0, 9] = torch.tensor([1.0 - 1.0 / 11, 1.0 - 1.0 / 13, 1.0 - 1.0 / 9])     # exactly the last voxel centre
    go = torch.randn(N, C, 1, 1, P, generator=g)
    ggG = torch.randn(N, 1, 1, P, 3, generator=g)
    _, gg_o = oracle.gs3d_backward(inp, grid, go)
    _, b_o, c_o = oracle.gs3d_dbackward(None, ggG, inp, grid, go, need_grad_input=False)

    def close(name, got, want):
        scale = float(want.abs().max())
        err = float((got.cpu() - want).abs().max()) / scale
        assert err <= 2e-6, (name, C, err)
        return err

    _, gg = GridSamplerMine.backward(gpu(inp), gpu(grid), gpu(go), 0, 1, need_grad_input=False)
    _, b, c = GridSamplerMine.dbackward(None, gpu(ggG), gpu(inp), gpu(grid), gpu(go), 0, 1, need_grad_input=False)
    errs = [close('grad_grid', gg, gg_o), close('dbackward grad_grid', b, b_o), close('grad_grad_output', c, c_o)]
    print("C=%d: largest deviations relative to the largest entry: %.1e %.1e %.1e" % (C, *errs))
    for t in (gg, b, c):                                     # a NaN coordinate: the reference skips every corner -> zeros
        assert torch.isfinite(t).all()
    assert float(gg[0, 0, 0, 5].abs().max()) == 0.0 and float(c[0, :, 0, 0, 5].abs().max()) == 0.0     # (NaN; +inf clips to the border)
    # run to run reproducible (a fixed butterfly, no atomics)
    _, gg_again = GridSamplerMine.backward(gpu(inp), gpu(grid), gpu(go), 0, 1, need_grad_input=False)
    assert torch.equal(gg, gg_again)
    # a strided grad_output view and a single batch item
    go_wide = torch.randn(1, C, 1, 1, 2 * P, generator=g)
    _, gg1_o = oracle.gs3d_backward(inp[:1], grid[:1], go_wide[..., ::2].contiguous())
    _, gg1 = GridSamplerMine.backward(gpu(inp[:1]), gpu(grid[:1]), gpu(go_wide)[..., ::2], 0, 1, need_grad_input=False)
    close('grad_grid, strided grad_output', gg1, gg1_o)


@pytest.mark.parametrize("C", [4, 8, 12, 16, 24, 32, 20])
def test_sampler_forward_record_lanes_vs_oracle(oracle, C):
    """The record-coalesced forward (G lanes per point x V channels per lane, branch-free clamped gathers, for lists of >= 2048
    points) and the fall-back for widths without a lane split (C = 20) or other grid shapes: bit-exact, incl. a ragged last block,
    points outside the volume and several batch items."""
    from recmv import GridSamplerMine
    inp, grid = _sampler_case(torch.float32, C=C, dims=(9, 13, 11), P=4099 + 64 * 37 + 5, channels_last=True, seed=C, spread=2.6)
    assert torch.equal(GridSamplerMine.forward(gpu(inp), gpu(grid), 0, 1).cpu(), oracle.gs3d_forward(inp, grid))
    g = torch.Generator().manual_seed(C + 1)
    inp3 = torch.randn(3, C, 7, 9, 8, generator=g).contiguous(memory_format=torch.channels_last_3d)
    grid3 = (torch.rand(3, 2, 3, 700, 3, generator=g) - 0.5) * 2.4
    assert torch.equal(GridSamplerMine.forward(gpu(inp3), gpu(grid3), 0, 1).cpu(), oracle.gs3d_forward(inp3, grid3))


def test_sampler_equals_torch_grid_sample_on_gpu():
    from recmv import GridSamplerMine
    inp, grid = _sampler_case(torch.float32, C=24, dims=(17, 29, 21), P=20000, channels_last=True)
    ref = F.grid_sample(inp, grid, mode="bilinear", padding_mode="border", align_corners=False)
    out = GridSamplerMine.forward(gpu(inp), gpu(grid), 0, 1)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-6, atol=2e-6)


def test_sampler_multibatch_strided_grid(oracle):
    from recmv import GridSamplerMine
    g = torch.Generator().manual_seed(5)
    inp = torch.randn(2, 6, 5, 7, 9, generator=g)
    grid_full = (torch.rand(2, 2, 3, 4, 6, generator=g) - 0.5) * 2.4
    grid = grid_full[..., ::2]                                        # strided last dim
    out_o = oracle.gs3d_forward(inp, grid)
    out_g = GridSamplerMine.forward(gpu(inp), gpu(grid_full)[..., ::2], 0, 1)
    assert torch.equal(out_g.cpu(), out_o)


def test_sampler_functions_gradcheck_like_reference():
    """MCAcc/check_grid_sampler_mine.py:5-16 on the HIP kernels (f64)."""
    from recmv.MCAcc.grid_sampler_mine import GridSamplerMine3dBackwardFunction, GridSamplerMine3dFunction
    torch.manual_seed(0)
    inp = torch.randn(1, 5, 15, 15, 15, dtype=torch.double, device=DEV, requires_grad=True)
    grid = ((torch.rand(1, 1, 1, 10, 3, dtype=torch.double, device=DEV) - 0.5) * 2.2).requires_grad_(True)
    assert torch.autograd.gradcheck(GridSamplerMine3dFunction.apply, (inp, grid))
    go = torch.randn(1, 5, 1, 1, 10, dtype=torch.double, device=DEV, requires_grad=True)
    assert torch.autograd.gradcheck(GridSamplerMine3dBackwardFunction.apply, (inp, grid, go))
    # frozen volume: gradients only wrt grid / grad_output
    inp_f = inp.detach()
    assert torch.autograd.gradcheck(lambda g: GridSamplerMine3dFunction.apply(inp_f, g), (grid,))
    assert torch.autograd.gradgradcheck(lambda g: GridSamplerMine3dFunction.apply(inp_f, g), (grid,))


def test_sampler_argument_errors():
    from recmv import GridSamplerMine
    inp = torch.randn(1, 2, 3, 3, 3, device=DEV)
    grid = torch.zeros(1, 1, 1, 2, 3, device=DEV)
    with pytest.raises(RuntimeError, match="Bilinear"):
        GridSamplerMine.forward(inp, grid, 1, 1)
    with pytest.raises(RuntimeError, match="Border"):
        GridSamplerMine.forward(inp, grid, 0, 0)
    with pytest.raises(RuntimeError, match="dtype"):
        GridSamplerMine.forward(inp, grid.double(), 0, 1)
    with pytest.raises(RuntimeError, match="batch"):
        GridSamplerMine.forward(inp, grid.repeat(2, 1, 1, 1, 1), 0, 1)
    out = GridSamplerMine.forward(inp, torch.zeros(1, 1, 1, 0, 3, device=DEV), 0, 1)     # empty grid
    assert out.shape == (1, 2, 1, 1, 0)


# ------------------------------------------------------------------------------------------ interp2x
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", [(1, 1, 2, 2, 2), (1, 1, 6, 9, 5), (2, 3, 4, 3, 7), (1, 1, 33, 33, 33)])
def test_interp2x_bit_exact(oracle, dtype, shape):
    from recmv import interp2x_boundary3d
    torch.manual_seed(sum(shape))
    x = torch.randn(*shape, dtype=dtype)
    out_o, bnd_o = oracle.interp2x_forward(x, 0.1)
    out_g, bnd_g = interp2x_boundary3d.forward(gpu(x), 0.1)
    assert bnd_g.dtype == torch.bool
    assert torch.equal(out_g.cpu(), out_o) and torch.equal(bnd_g.cpu(), bnd_o)
    go = torch.randn_like(out_o)
    assert torch.equal(interp2x_boundary3d.backward(gpu(go)).cpu(), oracle.interp2x_backward(go))


# ------------------------------------------------------------------------------------------ MC
def _noise_volume(shape, seed):
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(*shape, generator=g)
    # smooth a little so the surface is not pure salt-and-pepper, keep plenty of ambiguous cases
    if min(shape) >= 3:
        v = F.avg_pool3d(v[None, None], 3, 1, 1)[0, 0]
    return v.contiguous()


@pytest.mark.parametrize("shape,seed", [((9, 9, 9), 0), ((33, 33, 33), 1), ((17, 21, 13), 2), ((5, 7, 65), 3),
                                        ((6, 5, 66), 4), ((4, 4, 130), 5), ((3, 3, 2), 6), ((40, 9, 64), 7),
                                        ((2, 2, 2), 8), ((65, 65, 65), 9)])
def test_mc_bit_exact_vs_oracle(oracle, shape, seed):
    from recmv import MCGpu
    vol = _noise_volume(shape, seed)
    args = (0.013, 0.021, 0.017, -0.5, -0.6, -0.4, 0.02)
    v_o, f_o = oracle.mc(vol, *args)
    v_g, f_g = MCGpu.mc_gpu(gpu(vol), *args)
    assert v_g.dtype == torch.float32 and f_g.dtype == torch.int64
    assert v_g.shape == v_o.shape and f_g.shape == f_o.shape
    assert torch.equal(f_g.cpu(), f_o), "triangle/vertex indexing must be bit-exact"
    assert torch.equal(v_g.cpu(), v_o), "vertex positions must be bit-exact"


def test_mc_single_pass_route_capacity_overflow_and_unaligned_base(oracle):
    """mc_gpu's second call for a grid takes the no-round-trip route (recmv_mc_run into buffers sized from the previous
    extraction); a surface that outgrows the margin repeats the emit pass with exact sizes; a volume whose base pointer
    is only 4-byte aligned (a slice) goes through the staging tile's alignment peel.  All bit-exact vs the oracle."""
    import ctypes as C
    from recmv import MCGpu, _lib as L
    args = (0.013, 0.021, 0.017, -0.5, -0.6, -0.4, 0.0)
    shape = (21, 19, 70)
    small = _noise_volume(shape, 11) + 1.2          # few sign changes
    dense = _noise_volume(shape, 12)                # many more: overflows the capacity guessed from `small`
    MCGpu._last_sizes.clear()
    for vol in (small, small, dense, dense, small):
        v_o, f_o = oracle.mc(vol, *args)
        v_g, f_g = MCGpu.mc_gpu(gpu(vol), *args)
        assert torch.equal(f_g.cpu(), f_o) and torch.equal(v_g.cpu(), v_o)
    assert oracle.mc(dense, *args)[0].shape[0] > 1.25 * oracle.mc(small, *args)[0].shape[0] + 4096
    # capacities are never exceeded: run with room for 10 vertices / 7 faces and check the guard words behind them
    lib = L.lib()
    vol = gpu(dense)
    ws = torch.empty(int(lib.recmv_mc_workspace_bytes(*shape)), dtype=torch.uint8, device=DEV)
    vbuf = torch.full((40, 3), -7.0, device=DEV)
    fbuf = torch.full((40, 3), -7, dtype=torch.int64, device=DEV)
    cdev = torch.zeros(3, dtype=torch.int32, device=DEV)
    L.check(lib.recmv_mc_run(L.ptr(vol), *shape, args[6], *args[:6], L.ptr(ws), ws.numel(), L.ptr(vbuf), 10,
                             L.ptr(fbuf), 7, L.ptr(cdev), L.stream_ptr(vol.device)), "mc_run")
    torch.cuda.synchronize()
    v_o, f_o = oracle.mc(dense, *args)
    assert cdev[:2].tolist() == [v_o.shape[0], f_o.shape[0]]
    assert torch.equal(vbuf[:10].cpu(), v_o[:10]) and torch.equal(fbuf[:7].cpu(), f_o[:7])
    assert (vbuf[10:] == -7.0).all() and (fbuf[7:] == -7).all()
    # base pointer misaligned by 1, 2, 3 floats
    for shift in (1, 2, 3):
        flat = torch.zeros(dense.numel() + 8, device=DEV)
        flat[shift:shift + dense.numel()] = gpu(dense).reshape(-1)
        view = flat[shift:shift + dense.numel()].view(*shape)
        assert view.data_ptr() % 16 == 4 * shift and view.is_contiguous()
        v_g, f_g = MCGpu.mc_gpu(view, *args)
        assert torch.equal(f_g.cpu(), f_o) and torch.equal(v_g.cpu(), v_o)


@pytest.mark.parametrize("shape", [(70, 37, 129), (37, 70, 257), (9, 140, 31), (12, 10, 515)])
def test_mc_tile_shapes_bit_exact_vs_oracle(oracle, shape):
    """Volumes that exercise every staging-tile shape of the classify pass (rows of 31 ... 515 floats, clipped tiles in
    i and j, several tiles per slab): a smooth field with a few thousand faces, bit-exact vs the oracle."""
    from recmv import MCGpu
    ax = [torch.linspace(-1, 1, n) for n in shape]
    X, Y, Z = torch.meshgrid(*ax, indexing="ij")
    vol = (torch.sqrt(X ** 2 + (0.9 * Y) ** 2 + (1.1 * Z) ** 2) - 0.83 + 0.05 * torch.sin(5 * X + 3 * Y) * torch.cos(4 * Z)
           ).float().contiguous()
    args = (0.02, 0.03, 0.01, -1.0, -1.0, -1.0, 0.0)
    v_o, f_o = oracle.mc(vol, *args)
    assert f_o.shape[0] > 1000
    for _ in range(2):                                        # two-phase route, then the single-pass route
        v_g, f_g = MCGpu.mc_gpu(gpu(vol), *args)
        assert torch.equal(f_g.cpu(), f_o) and torch.equal(v_g.cpu(), v_o)


def test_mc_sphere_and_degenerate(oracle):
    from recmv import MCGpu
    from test_mc_oracle import mesh_invariants, sphere_volume
    vol = sphere_volume(65)
    v, f = MCGpu.mc_gpu(gpu(vol), 2 / 64, 2 / 64, 2 / 64, -1.0, -1.0, -1.0, 0.0)
    assert mesh_invariants(v.cpu(), f.cpu()) == 2
    v, f = MCGpu.mc_gpu(torch.ones(5, 6, 7, device=DEV))
    assert v.shape == (0, 3) and f.shape == (0, 3)
    assert MCGpu.mc_gpu(torch.ones(0, 6, 7, device=DEV)) == []
    big = sphere_volume(9, r=1.2)                                      # surface touches the box
    v_o, f_o = oracle.mc(big)
    v, f = MCGpu.mc_gpu(gpu(big))
    assert torch.equal(f.cpu(), f_o) and (f == -1).any()
    with pytest.raises(RuntimeError):
        MCGpu.mc_gpu(torch.ones(4, 4, 4, device=DEV).double())


def test_mc_full_size_257_properties():
    """BASELINE config 3 size: 257^3.  Size-independent properties: deterministic, closed manifold, chi=2."""
    from recmv import MCGpu
    from test_mc_oracle import mesh_invariants
    n = 257
    ax = torch.linspace(-1, 1, n, device=DEV)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    vol = (torch.sqrt(X ** 2 + (Y * 0.8) ** 2 + Z ** 2) - 0.6 + 0.03 * torch.sin(9 * X) * torch.cos(7 * Z)).contiguous()
    step = 2.0 / (n - 1)
    v1, f1 = MCGpu.mc_gpu(vol, step, step, step, -1.0, -1.0, -1.0, 0.0)
    v2, f2 = MCGpu.mc_gpu(vol, step, step, step, -1.0, -1.0, -1.0, 0.0)
    assert torch.equal(v1, v2) and torch.equal(f1, f2)
    assert v1.shape[0] > 100000
    assert mesh_invariants(v1.cpu(), f1.cpu()) == 2


@pytest.mark.parametrize("shape", [(33, 41, 25), (65, 65, 65), (129, 97, 70)])
def test_mc_three_volumes_in_one_launch_set_equal_three_extractions(shape):
    """MCGpu.mc_gpu_multi / recmv_mc_run_batch — the body's and the garments' volumes of a re-mesh through ONE set of four launches
    (grid y = volume) — against one MCGpu.mc_gpu per volume (OptimGarmentNetwork.py:581-618): vertices and faces bit-identical per
    volume; a surface that outgrows its capacity guess between two calls takes the exact-size emit of its own volume; the first call
    of a grid (no guess yet) is the per-volume route."""
    from recmv import MCGpu
    gen = torch.Generator().manual_seed(sum(shape))
    ax = [torch.linspace(-1, 1, n) for n in shape]
    X, Y, Z = torch.meshgrid(*ax, indexing="ij")

    def vol(r, wob):
        return (torch.sqrt(X ** 2 + (0.8 * Y) ** 2 + Z ** 2) - r + wob * torch.sin(9 * X) * torch.cos(7 * Z)
                + 0.01 * torch.randn(shape, generator=gen)).contiguous().to(DEV)

    geom = (2. / shape[0], 2. / shape[1], 2. / shape[2], -1., -1., -1., 0.0)
    first = [vol(0.5, 0.02), vol(0.6, 0.03), vol(0.3, 0.0)]
    second = [vol(0.52, 0.02), vol(0.9, 0.05), vol(0.31, 0.0)]          # the middle surface grows by far more than the 25 % margin
    MCGpu._last_sizes.clear()
    for vols in (first, second, first):
        got = MCGpu.mc_gpu_multi(vols, *geom)
        for v3, (v, f) in zip(vols, got):
            want_v, want_f = MCGpu.mc_gpu(v3, *geom)
            assert torch.equal(v, want_v) and torch.equal(f, want_f) and v.shape[0] > 50
    assert MCGpu.mc_gpu_multi([first[0]], *geom)[0][0].shape == got[0][0].shape          # a single volume: the per-volume route


# ------------------------------------------------------------------------------------------ GEMM / PE
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 3, 39), (257, 512, 39), (300, 473, 512), (1000, 257, 512),
                                   (129, 130, 167), (4096, 512, 512), (77, 3, 512), (640, 512, 289), (6144, 512, 512),
                                   (3072, 512, 473), (20000, 512, 512), (116000, 512, 64)])   # the last two: 64x128 / 128x128 occupancy tiles
def test_gemm_nt_vs_fp64(M, N, K):
    from recmv import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) / np.sqrt(K)
    bias = torch.randn(N, generator=g)
    ref = (A.double() @ B.double().t() + bias.double())
    out = ops.gemm_nt(gpu(A), gpu(B), gpu(bias))
    # f32 MFMA == an fmaf chain: error ~1e-7 * sum|a b|  (guide: 0.75-1.5e-7)
    bound = 4e-7 * (A.abs().double() @ B.abs().double().t()) + 1e-6
    assert ((out.cpu().double() - ref).abs() <= bound).all()
    # the activations are 1-Lipschitz, so the same error bound carries through the fused epilogue
    def within(out, ref64, scale=1.0):
        assert ((out.cpu().double() - ref64).abs() <= scale * bound + 2e-7 * ref64.abs()).all()

    within(ops.gemm_nt(gpu(A), gpu(B), gpu(bias), ops.ACT_SOFTPLUS, 100.0, 0.5), F.softplus(ref, beta=100) * 0.5)
    within(ops.gemm_nt(gpu(A), gpu(B), None, ops.ACT_RELU), torch.relu(A.double() @ B.double().t()))
    within(ops.gemm_nt(gpu(A), gpu(B), gpu(bias), ops.ACT_TANH), torch.tanh(ref))


@pytest.mark.parametrize("M0,M1,N,K", [(128, 77, 512, 512), (3072, 2900, 512, 512), (3200, 3000, 473, 512), (2944, 3071, 512, 39),
                                       (40960, 40000, 512, 512), (384, 5, 1, 512), (1280, 1200, 3, 512)])
def test_gemm_nt_seg_equals_two_plain_products(M0, M1, N, K):
    """recmv_gemm_nt_seg / recmv_gemm_nt_mulgrad_seg (rows [0, M0) x B^T, rows [M0, M0 + M1) x B2^T in ONE launch; M0 a multiple
    of 128) against the two plain launches, softplus epilogue, no bias, and the activation-gradient epilogue alike.  The 64 x 64
    and 128 x 128 tiles run one fma chain over k per output element, the 64 x 32 tile (fewest rows) sums two half-K chains: where
    the combined launch and a plain one get the same kind of chain the comparison is bit for bit, otherwise within the f32 MFMA
    bound of test_gemm_nt_vs_fp64 (both sides against fp64)."""
    from recmv import _lib as L, ops
    g = torch.Generator().manual_seed(M0 + N)
    A = gpu(torch.randn(M0 + M1, K, generator=g))
    B, B2 = gpu(torch.randn(N, K, generator=g) / np.sqrt(K)), gpu(torch.randn(N, K, generator=g) / np.sqrt(K))
    b, b2 = gpu(torch.randn(N, generator=g)), gpu(torch.randn(N, generator=g))
    Y = gpu(torch.rand(M0 + M1, N, generator=g))
    lib, st = L.lib(), L.stream_ptr(A.device)
    narrow = lambda M: -(-M // 64) * -(-N // 64) < 640 and -(-M // 128) * -(-N // 128) < 512      # dispatch_nt's tile choice
    exact = {0: narrow(M0 + M1) == narrow(M0), 1: narrow(M0 + M1) == narrow(M1)}

    def same(got, want, Ar, W, part, scale=1.0):
        if exact[part]:
            assert torch.equal(got, want), (part, float((got - want).abs().max()))
        else:
            bound = scale * (4e-7 * (Ar.abs().double() @ W.abs().double().t()) + 1e-6) * 2
            assert ((got.double() - want.double()).abs() <= bound + 4e-7 * want.abs().double()).all(), part

    def seg(bias, bias2, act, p, scale):
        out = torch.empty(M0 + M1, N, device=DEV)
        L.check(lib.recmv_gemm_nt_seg(L.ptr(A), K, L.ptr(B), K, L.ptr(bias), L.ptr(B2), L.ptr(bias2), M0, L.ptr(out), N, M0 + M1, N, K,
                                      act, p, scale, st), "gemm_nt_seg")
        return out

    for bias, bias2, act, p, scale in ((b, b2, ops.ACT_SOFTPLUS, 100.0, 0.5), (None, None, ops.ACT_NONE, 0.0, 1.0)):
        got = seg(bias, bias2, act, p, scale)
        same(got[:M0], ops.gemm_nt(A[:M0], B, bias, act, p, scale), A[:M0], B, 0)
        same(got[M0:], ops.gemm_nt(A[M0:], B2, bias2, act, p, scale), A[M0:], B2, 1)
    out = torch.empty(M0 + M1, N, device=DEV)
    L.check(lib.recmv_gemm_nt_mulgrad_seg(L.ptr(A), K, L.ptr(B), L.ptr(B2), M0, K, L.ptr(out), N, M0 + M1, N, K, L.ptr(Y), N,
                                          ops.ACT_SOFTPLUS, 100.0, 1.0, 1.0, st), "mulgrad_seg")
    for part, (rows, W) in enumerate(((slice(0, M0), B), (slice(M0, None), B2))):
        ref = torch.empty(out[rows].shape, device=DEV)
        L.check(lib.recmv_gemm_nt_mulgrad(L.ptr(A[rows]), K, L.ptr(W), K, L.ptr(ref), N, ref.shape[0], N, K, L.ptr(Y[rows]), N,
                                          ops.ACT_SOFTPLUS, 100.0, 1.0, 1.0, st), "mulgrad")
        same(out[rows], ref, A[rows], W, part)
    # a split that is not a multiple of the tile height is refused
    assert lib.recmv_gemm_nt_seg(L.ptr(A), K, L.ptr(B), K, None, L.ptr(B2), None, 100, L.ptr(out), N, M0 + M1, N, K, 0, 0.0, 1.0, st) != 0


_OCC_CHILD = """
import sys, torch
sys.path.insert(0, sys.argv[1])
from recmv import _lib as L, ops
out = {}
for (M, N, K) in ((20000, 512, 512), (116000, 512, 64), (115300, 473, 100)):
    g = torch.Generator().manual_seed(M + K)
    A, B, b = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) / K ** 0.5).cuda(), torch.randn(N, generator=g).cuda()
    Y = torch.rand(M, N, generator=g).cuda() * 0.05
    out[f"softplus{M}"] = ops.gemm_nt(A, B, b, ops.ACT_SOFTPLUS, 100.0, 0.5).cpu()
    o = torch.empty(M, N, device="cuda")
    L.check(L.lib().recmv_gemm_nt_mulgrad(L.ptr(A), K, L.ptr(B), K, L.ptr(o), N, M, N, K, L.ptr(Y), N, ops.ACT_SOFTPLUS, 100.0, 1.0,
                                          1.0, L.stream_ptr(A.device)), "mulgrad")
    out[f"mulgrad{M}"] = o.cpu()
    if K % 4 == 0 and N == 512:
        Yk = torch.rand(M, K, generator=g).cuda() * 0.05
        L.check(L.lib().recmv_gemm_nt_actgrad(L.ptr(A), K, L.ptr(Yk), K, L.ptr(B), K, L.ptr(o), N, M, N, K, ops.ACT_SOFTPLUS, 100.0,
                                              1.0, 1.0, L.stream_ptr(A.device)), "actgrad")
        out[f"actgrad{M}"] = o.cpu()
torch.save(out, sys.argv[2])
"""


def test_gemm_nt_occupancy_kernels_bit_equal_to_the_kernel_they_replace(tmp_path):
    """The large f32 products run gemm_nt_occ_kernel (one 16-column K-tile in LDS, four / five workgroups per CU, 128x128 or 64x128
    tiles by launch size); RECMV_GEMM_OCC=0 keeps gemm_nt_kernel<2, ...> (two per CU).  Same products, same order along k: the
    outputs are equal bit for bit — forward epilogue, activation-gradient epilogue and operand transform, aligned and ragged
    shapes (the switch is read once per process, hence two children)."""
    import subprocess
    import sys
    res = {}
    for occ in ("1", "0"):
        f = tmp_path / f"occ{occ}.pt"
        env = dict(os.environ, RECMV_GEMM_OCC=occ)
        subprocess.run([sys.executable, "-c", _OCC_CHILD, str(REPO / "rec-mv_amd"), str(f)], env=env, check=True, timeout=600)
        res[occ] = torch.load(f)
    assert set(res["1"]) == set(res["0"]) and len(res["1"]) >= 8
    for k in res["1"]:
        assert torch.equal(res["1"][k], res["0"][k]), (k, float((res["1"][k] - res["0"][k]).abs().max()))


def test_gemm_nt_strided_views():
    from recmv import ops
    g = torch.Generator().manual_seed(0)
    buf = torch.randn(300, 512, generator=g)
    A = gpu(buf)[:, :473]                                  # lda 512, K 473 (skip-layer shape)
    B = torch.randn(64, 473, generator=g)
    out = torch.zeros(300, 128, device=DEV)
    ops.gemm_nt(A, gpu(B), None, out=out[:, 32:96])
    ref = buf[:, :473].double() @ B.double().t()
    bound = 4e-7 * (buf[:, :473].abs().double() @ B.abs().double().t()) + 1e-6
    assert ((out[:, 32:96].cpu().double() - ref).abs() <= bound).all()
    assert (out[:, :32] == 0).all() and (out[:, 96:] == 0).all()


@pytest.mark.parametrize("K,M,N", [(1, 1, 1), (33, 5, 3), (1000, 512, 39), (5000, 473, 512), (150000, 512, 512),
                                   (777, 130, 257), (64, 3, 512)])
def test_gemm_tn_vs_fp64(K, M, N):
    from recmv import ops
    g = torch.Generator().manual_seed(K + M)
    A = torch.randn(K, M, generator=g)
    B = torch.randn(K, N, generator=g)
    ref = A.double().t() @ B.double()
    out = ops.gemm_tn(gpu(A), gpu(B))
    bound = 4e-7 * (A.abs().double().t() @ B.abs().double()) + 1e-6
    assert ((out.cpu().double() - ref).abs() <= bound).all()
    assert torch.equal(out, ops.gemm_tn(gpu(A), gpu(B))), "split-K reduction order is fixed: deterministic"


def test_gemm_tn_widths_inside_wider_buffers():
    """dW of the skip layer: 473 of 512 columns on either side (row stride 512 covers the width rounded up to 4, so the
    four-workgroups-per-CU kernel runs with whole-float4 loads; what lies in the padding columns — NaN here — must not reach
    the result)."""
    from recmv import ops
    g = torch.Generator().manual_seed(5)
    K = 9000
    bufA, bufB = torch.randn(K, 512, generator=g), torch.randn(K, 512, generator=g)
    bufA[:, 473:] = float("nan")
    bufB[:, 473:] = float("nan")
    for (M, N) in ((473, 473), (473, 512), (512, 473), (6, 473)):
        A, B = gpu(bufA)[:, :M], gpu(bufB)[:, :N]
        if N == 512:
            B = gpu(torch.nan_to_num(bufB, nan=0.5))
        if M == 512:
            A = gpu(torch.nan_to_num(bufA, nan=0.25))
        ref = A.cpu().double().t() @ B.cpu().double()
        out = ops.gemm_tn(A, B)
        bound = 4e-7 * (A.cpu().abs().double().t() @ B.cpu().abs().double()) + 1e-6
        assert torch.isfinite(out).all()
        assert ((out.cpu().double() - ref).abs() <= bound).all(), (M, N)


def test_matmul_functions_gradcheck_structure():
    """First and second derivatives of the product Functions stay on the kernels and match autograd of @."""
    from recmv import ops
    g = torch.Generator().manual_seed(1)
    A = gpu(torch.randn(70, 40, generator=g)).requires_grad_(True)
    B = gpu(torch.randn(50, 40, generator=g)).requires_grad_(True)
    A2, B2 = A.detach().clone().requires_grad_(True), B.detach().clone().requires_grad_(True)

    def second_order(mm, a, b):
        y = mm(a, b)
        ga, = torch.autograd.grad((y ** 2).sum(), a, create_graph=True)
        l = (ga ** 2).sum() + y.sum()
        return torch.autograd.grad(l, [a, b])

    r = second_order(lambda a, b: a @ b.t(), A2, B2)
    o = second_order(ops.MatmulNT.apply, A, B)
    for x, y in zip(o, r):
        torch.testing.assert_close(x, y, rtol=1e-3, atol=1e-4 * float(y.abs().max()))


def test_posenc_kernel_vs_torch_path():
    from recmv import ops
    from recmv.model.Embedder import get_embedder
    from recmv.utils import annealing_weights
    x = gpu(torch.randn(1000, 3, generator=torch.Generator().manual_seed(0)))
    embed, dim = get_embedder(6)
    assert dim == 39
    for ws in (None, annealing_weights(6, 0.62), [0.0] * 12):
        hip = ops.posenc(x, 6, ws)
        tor = embed(x.clone().requires_grad_(True), ws)             # torch path
        torch.testing.assert_close(hip, tor.detach(), rtol=1e-6, atol=2e-6)
    buf = torch.full((1000, 512), 7.0, device=DEV)
    ops.posenc(x, 6, None, 0.5, out=buf[:, 473:], ld_fill=39)
    torch.testing.assert_close(buf[:, 473:], 0.5 * embed(x.clone().requires_grad_(True)).detach(), rtol=1e-6,
                               atol=2e-6)
    assert (buf[:, :473] == 7.0).all()


# ------------------------------------------------------------------------------------------ fused helpers
def test_posenc_function_first_and_second_order_vs_torch_path():
    """PosEnc / PosEncVjp / PosEncJvp / PosEncVjp2 against autograd of the plain torch encoding (f32)."""
    from recmv import ops
    from recmv.utils import annealing_weights
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(300, 3, generator=g)
    ws = tuple(annealing_weights(6, 0.62))

    def torch_pe(x):
        outs = [x]
        for b in range(6):
            outs.append(ws[2 * b] * torch.sin(x * (2.0 ** b)))
            outs.append(ws[2 * b + 1] * torch.cos(x * (2.0 ** b)))
        return torch.cat(outs, -1)

    Wt = gpu(torch.randn(39, 5, generator=g))

    def run(pe):
        x = gpu(x0).clone().requires_grad_(True)
        y = pe(x) @ Wt                                     # mix the features like a layer would
        f = torch.tanh(y).sum(1)
        gx, = torch.autograd.grad(f.sum(), x, create_graph=True)
        loss = ((gx.norm(dim=1) - 1) ** 2).mean() + f.mean()
        gx2, = torch.autograd.grad(loss, x)
        return y.detach(), gx.detach(), gx2

    a = run(lambda x: ops.PosEnc.apply(x, 6, ws))
    b = run(torch_pe)
    torch.testing.assert_close(a[0], b[0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a[1], b[1], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(a[2], b[2], rtol=2e-3, atol=2e-3 * float(b[2].abs().max()))


def test_act_grad_and_weight_norm_functions():
    from recmv import ops
    g = torch.Generator().manual_seed(1)
    for act, param, fn in ((ops.ACT_SOFTPLUS, 100.0, lambda z: torch.nn.functional.softplus(z, beta=100)),
                           (ops.ACT_TANH, 0.0, torch.tanh), (ops.ACT_RELU, 0.0, torch.relu)):
        z = gpu(torch.randn(257, 33, generator=g) * 0.05).requires_grad_(True)
        gy = gpu(torch.randn(257, 33, generator=g)).requires_grad_(True)
        y = fn(z)
        ref_gz, = torch.autograd.grad(y, z, gy, create_graph=True)
        out = ops.ActGrad.apply(gy, y, act, param)
        torch.testing.assert_close(out, ref_gz, rtol=1e-4, atol=1e-5)
        # second order: d/dz and d/dgy of sum(gz * r)
        r = gpu(torch.randn(257, 33, generator=g))
        a = torch.autograd.grad((out * r).sum(), [z, gy], allow_unused=True, retain_graph=True)
        b = torch.autograd.grad((ref_gz * r).sum(), [z, gy], allow_unused=True)
        for u, v in zip(a, b):
            if v is None or float(v.abs().max()) == 0:
                assert u is None or float(u.abs().max()) == 0
            else:
                torch.testing.assert_close(u, v, rtol=2e-3, atol=2e-3 * float(v.abs().max()) + 1e-6)
    v = gpu(torch.randn(473, 512, generator=g)).requires_grad_(True)
    gg = gpu(torch.rand(473, 1, generator=g) + 0.5).requires_grad_(True)
    W = ops.weight_norm(v, gg)
    Wr = gg * (v / v.norm(dim=1, keepdim=True))
    torch.testing.assert_close(W, Wr, rtol=1e-5, atol=1e-6)
    r = gpu(torch.randn(473, 512, generator=g))
    a = torch.autograd.grad((W * r).sum(), [v, gg])
    b = torch.autograd.grad((Wr * r).sum(), [v, gg])
    torch.testing.assert_close(a[0], b[0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a[1], b[1], rtol=1e-4, atol=1e-4)


def test_kinematic_chain_kernel_vs_python_loop():
    """Fused chain (csrc/kinematic_chain.hip) vs the 23-step loop restatement of model/Deformer.py:372-405."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    import common_setup as cs
    from recmv.model import LBSkinner
    sk = cs.build_skinner(LBSkinner).to(DEV)
    g = torch.Generator().manual_seed(3)
    poses = gpu(0.4 * torch.randn(7, 24, 3, generator=g))
    poses[0, 3] = 0.0                                     # zero rotation: the 1e-8 guard
    p1 = poses.clone().requires_grad_(True)
    p2 = poses.clone().requires_grad_(True)
    G1, A1 = sk._chain_fused(p1)
    G2, _ = sk._chain(p2)
    A2 = (G2.unsqueeze(-1) * sk.init_pose.view(1, 24, 4, 4).unsqueeze(-3)).sum(-2)
    torch.testing.assert_close(G1, G2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(A1, A2, rtol=1e-5, atol=1e-6)
    rG, rA = gpu(torch.randn(7, 24, 4, 4, generator=g)), gpu(torch.randn(7, 24, 4, 4, generator=g))
    ga, = torch.autograd.grad((G1 * rG).sum() + (A1 * rA).sum(), p1)
    gb, = torch.autograd.grad((G2 * rG).sum() + (A2 * rA).sum(), p2)
    torch.testing.assert_close(ga, gb, rtol=2e-4, atol=2e-4)


def test_singular_values_closed_form():
    from recmv.loop import singular_values_3x3
    g = torch.Generator().manual_seed(5)
    J = gpu(torch.eye(3).view(1, 3, 3) + 0.3 * torch.randn(2000, 3, 3, generator=g))
    s = singular_values_3x3(J)
    ref = torch.linalg.svdvals(J.cpu().double()).float()
    torch.testing.assert_close(s.cpu(), ref, rtol=2e-3, atol=2e-4)


@pytest.mark.gpu
def test_grid_sampler_accepts_half_like_the_reference_dispatch():
    """The reference dispatches its sampler kernels for half as well (AT_DISPATCH_FLOATING_TYPES_AND_HALF, MCAcc/cuda/
    GridSamplerMineKernel.cu:931,963,1001).  Here an f16 call computes in f32 and rounds the results once: forward, backward
    and double backward return f16 tensors that agree with the f32 path on the same (f16-representable) operands to f16 rounding."""
    from recmv import GridSamplerMine
    g = torch.Generator().manual_seed(3)
    vol = gpu(torch.randn(1, 24, 9, 11, 7, generator=g)).half()
    grid = gpu((torch.rand(1, 1, 1, 700, 3, generator=g) - 0.5) * 2.2).half()
    go = gpu(torch.randn(1, 24, 1, 1, 700, generator=g)).half()
    ggg = gpu(torch.randn(1, 1, 1, 700, 3, generator=g)).half()
    out_h = GridSamplerMine.forward(vol, grid, 0, 1)
    out_f = GridSamplerMine.forward(vol.float(), grid.float(), 0, 1)
    assert out_h.dtype == torch.float16 and torch.equal(out_h, out_f.half())
    gi_h, gg_h = GridSamplerMine.backward(vol, grid, go, 0, 1)
    gi_f, gg_f = GridSamplerMine.backward(vol.float(), grid.float(), go.float(), 0, 1)
    assert gi_h.dtype == gg_h.dtype == torch.float16
    assert torch.equal(gg_h, gg_f.half())
    torch.testing.assert_close(gi_h.float(), gi_f, rtol=2e-3, atol=2e-3)      # (atomics scatter, see below)
    r_h = GridSamplerMine.dbackward(None, ggg, vol, grid, go, 0, 1)
    r_f = GridSamplerMine.dbackward(None, ggg.float(), vol.float(), grid.float(), go.float(), 0, 1)
    for i, (a, b) in enumerate(zip(r_h, r_f)):
        assert a.dtype == torch.float16
        if i == 0:       # grad_input is a scatter with float atomics (as in the reference): two calls differ in the last f32 bits
            torch.testing.assert_close(a.float(), b, rtol=2e-3, atol=2e-3)
        else:
            assert torch.equal(a, b.half())
    with pytest.raises(RuntimeError):                      # mixed dtypes are still refused, as in the reference (:31-33)
        GridSamplerMine.forward(vol, grid.float(), 0, 1)


# ------------------------------------------------------------------------------------------ matrix mode bf16x6
@pytest.fixture
def bf16x6_mode():
    """Switch librecmv_hip.so to the 3-way-bf16-split matrix mode for one test (recmv_set_gemm_mode)."""
    from recmv import _lib as L
    prev = L.set_gemm_mode(1)
    try:
        yield
    finally:
        L.set_gemm_mode(prev)


@pytest.mark.parametrize("M,N,K", [(5, 3, 39), (300, 473, 512), (4096, 512, 512), (129, 130, 167), (20000, 512, 512),
                                   (70001, 512, 40), (33000, 257, 168), (16400, 473, 36)])
def test_gemm_nt_bf16x6_same_bound_as_f32(bf16x6_mode, M, N, K):
    """The six-product bf16 split meets the SAME error bound vs fp64 as the exact-f32 MFMA path.  The last four shapes
    take the staged-split 128x128 kernel (gemm_nt_b3_kernel): whole K-tiles, a K tail, a single short K-tile, ragged
    M and N."""
    from recmv import ops
    g = torch.Generator().manual_seed(M + 3 * N)
    A = torch.randn(M, K, generator=g) * torch.logspace(-3, 2, M).view(-1, 1)      # rows over 5 decades
    B = torch.randn(N, K, generator=g) / np.sqrt(K)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ B.double().t() + bias.double()
    out = ops.gemm_nt(gpu(A), gpu(B), gpu(bias))
    bound = 4e-7 * (A.abs().double() @ B.abs().double().t()) + 1e-6
    assert ((out.cpu().double() - ref).abs() <= bound).all()


@pytest.mark.parametrize("K,M,N", [(33, 5, 3), (5000, 473, 512), (150000, 512, 512)])
def test_gemm_tn_bf16x6_same_bound_as_f32(bf16x6_mode, K, M, N):
    from recmv import ops
    g = torch.Generator().manual_seed(K + M)
    A = torch.randn(K, M, generator=g)
    B = torch.randn(K, N, generator=g)
    ref = A.double().t() @ B.double()
    out = ops.gemm_tn(gpu(A), gpu(B))
    bound = 4e-7 * (A.abs().double().t() @ B.abs().double()) + 1e-6
    assert ((out.cpu().double() - ref).abs() <= bound).all()
    assert torch.equal(out, ops.gemm_tn(gpu(A), gpu(B)))


# ------------------------------------------------------------------------------------------ ray-path pieces of round 2
def test_act_grad_2d_vector_and_scalar_paths_agree():
    """recmv_act_grad_2d: the float4 path (aligned, widths % 4 == 0) and the element path give the same numbers, both
    equal to g * act'(y) in torch (softplus beta=100: act'(z) = 1 - exp(-beta y))."""
    import ctypes as C
    from recmv import _lib as L
    from recmv import ops
    g0 = torch.Generator().manual_seed(5)
    P, Cw = 3001, 512
    gy = gpu(torch.randn(P, Cw, generator=g0))
    y = gpu(torch.rand(P, Cw, generator=g0) * 0.05)
    want = 0.5 * gy * (-torch.expm1(-100.0 * 1.25 * y))
    outs = []
    for ld in (Cw, Cw + 1):                     # ld 513: rows lose their 16-byte alignment -> element path
        buf_g = torch.zeros(P, ld, device=gy.device)
        buf_y = torch.zeros(P, ld, device=gy.device)
        buf_o = torch.zeros(P, ld, device=gy.device)
        buf_g[:, :Cw], buf_y[:, :Cw] = gy, y
        L.check(L.lib().recmv_act_grad_2d(L.ptr(buf_g), ld, L.ptr(buf_y), ld, L.ptr(buf_o), ld, P, Cw, ops.ACT_SOFTPLUS,
                                          100.0, 1.25, 0.5, L.stream_ptr(gy.device)), "act_grad_2d")
        outs.append(buf_o[:, :Cw].clone())
    assert torch.equal(outs[0], outs[1])
    assert torch.allclose(outs[0], want, rtol=2e-6, atol=1e-7)
    # one cotangent row for every point (row stride 0)
    row = gpu(torch.randn(Cw, generator=g0))
    out = torch.empty(P, Cw, device=gy.device)
    L.check(L.lib().recmv_act_grad_2d(L.ptr(row), 0, L.ptr(y), Cw, L.ptr(out), Cw, P, Cw, ops.ACT_SOFTPLUS, 100.0, 1.0, 1.0,
                                      L.stream_ptr(gy.device)), "act_grad_2d")
    assert torch.allclose(out, row.view(1, -1) * (-torch.expm1(-100.0 * y)), rtol=2e-6, atol=1e-7)


def test_rootfind_step_equals_rootfind_update():
    """recmv_rootfind_step (step index on the device, marks for a polling host) performs exactly recmv_rootfind_update's
    arithmetic: same points, flags and counts over several steps, the last one without an update (step == times)."""
    from recmv import chains
    g0 = torch.Generator().manual_seed(11)
    P, times = 5000, 2
    p0 = torch.randn(P, 3, generator=g0)
    un0 = (torch.rand(P, generator=g0) < 0.9).to(torch.uint8)
    steps = []
    for _ in range(times + 1):
        f = torch.randn(P, generator=g0) * 1e-4
        steps.append((f, torch.randn(P, 3, generator=g0), torch.rand(P, generator=g0), torch.rand(P, generator=g0) * 0.05,
                      torch.randn(P, 3, generator=g0)))
    pa, ua = gpu(p0.clone()), gpu(un0.clone())
    pb, ub = gpu(p0.clone()), gpu(un0.clone())
    cnt_a = torch.zeros(times + 2, dtype=torch.int32, device=pa.device)
    ints = torch.zeros(2 * (times + 2) + 1, dtype=torch.int32, device=pa.device)
    counters, marks, state = ints[:times + 2], ints[times + 2:2 * (times + 2)], ints[2 * (times + 2):]
    for it, (f, gf, l2, ang, gd) in enumerate(steps):
        args = [gpu(t) for t in (f, gf, l2, ang, gd)]
        chains.rootfind_update(pa, *args, ua, cnt_a[it:it + 1], 5e-5, 0.02, 3.05, 1.0, it < times)
        chains.rootfind_step(pb, *args, ub, counters, marks, state, 5e-5, 0.02, 3.05, 1.0, times)
        assert int(state[0]) == it + 1
    assert torch.equal(pa, pb) and torch.equal(ua, ub)
    assert torch.equal(cnt_a[:times + 1], counters[:times + 1])
    assert torch.equal(marks[:times + 1], counters[:times + 1] + 1) and int(marks[times + 1]) == 0


@pytest.mark.parametrize("M,N,K", [(3100, 512, 512), (6144, 39, 512), (300, 473, 512), (20000, 512, 512), (9000, 257, 168)])
def test_gemm_nt_mulgrad_equals_product_then_activation_gradient(M, N, K):
    """recmv_gemm_nt_mulgrad: (A B^T) (.) softplus'(Y) * scale in the product's epilogue == the plain product followed by
    recmv_act_grad_2d (narrow / 64x64 / 128x128 tiles, ragged N and K)."""
    from recmv import _lib as L
    from recmv import ops
    g0 = torch.Generator().manual_seed(M + N)
    A = gpu(torch.randn(M, K, generator=g0))
    B = gpu(torch.randn(N, K, generator=g0) / np.sqrt(K))
    Y = gpu(torch.rand(M, N, generator=g0) * 0.05)
    out = torch.empty(M, N, device=A.device)
    L.check(L.lib().recmv_gemm_nt_mulgrad(L.ptr(A), K, L.ptr(B), K, L.ptr(out), N, M, N, K, L.ptr(Y), N, ops.ACT_SOFTPLUS,
                                          100.0, 1.25, 0.5, L.stream_ptr(A.device)), "gemm_nt_mulgrad")
    prod = ops.gemm_nt(A, B)
    want = 0.5 * prod * (-torch.expm1(-100.0 * 1.25 * Y))
    assert torch.allclose(out, want, rtol=3e-6, atol=1e-6 * float(prod.abs().max()))
    # ReLU: exact
    L.check(L.lib().recmv_gemm_nt_mulgrad(L.ptr(A), K, L.ptr(B), K, L.ptr(out), N, M, N, K, L.ptr(Y - 0.025), N, ops.ACT_RELU,
                                          0.0, 1.0, 1.0, L.stream_ptr(A.device)), "gemm_nt_mulgrad")
    assert torch.equal(out, prod * ((Y - 0.025) > 0).float())


# ------------------------------------------------------------------------------------------ deformation regulariser
@pytest.mark.gpu
@pytest.mark.parametrize("P,spread", [(1, 0.3), (255, 0.3), (256, 0.05), (70001, 0.3), (4099, 1e-3)])
def test_def_regu_value_and_gradient_match_the_host_svd(P, spread):
    """recmv_def_regu (Jacobi eigen-decomposition of J^T J, analytic gradient) against the reference's own route — torch.svd on
    the host, log, GMRobustError, autograd — in float64 (oracle.def_regu).  Jacobians around the identity (where a deformer
    lives: nearly repeated singular values, the closed form's worst case) and far from it; the gradient also through the
    autograd Function the loop calls."""
    from oracle import oracle as orc
    from recmv import ops
    g = torch.Generator().manual_seed(P)
    J = torch.eye(3).view(1, 3, 3) + spread * torch.randn(P, 3, 3, generator=g)
    if P > 300:
        J[7] = torch.eye(3)                                   # exact identity: x = 0, gradient 0
        J[8] = torch.diag(torch.tensor([2.0, 2.0, 0.5]))      # a repeated singular value away from 1
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
        J[9] = q                                              # a rotation: all singular values 1
    c = 0.01
    y_ref, g_ref = orc.def_regu(J, c)
    Jg = gpu(J).requires_grad_(True)
    y = ops.def_regu(Jg, c)
    w = gpu(torch.rand(P, generator=g) + 0.5)
    (y * w).sum().backward()
    # y in [0, 2): an f32 rounding of log(sigma) (1.5e-7 absolute near sigma = 1) moves x = sum log^2 by 3e-7 |log sigma| and y by
    # at most 0.5 / c^2 of that — a few 1e-6 at c = 0.01
    torch.testing.assert_close(y.detach().cpu(), y_ref.float(), rtol=2e-3, atol=2e-5)
    g_ref_w = (g_ref * w.cpu().double().view(-1, 1, 1)).float()
    err = (Jg.grad.cpu() - g_ref_w).abs()
    scale = g_ref_w.abs().amax(dim=(1, 2), keepdim=True)
    # dy/dJ ~ log(sigma) / c^2: an absolute rounding of 1e-7 in log(sigma) is 1e-3 of gradient at c = 0.01; the worst of the 630 k
    # elements of the largest case sits at 3.8e-3 with the rotations' column updates compiled to packed f32 instructions and at 5.3e-3
    # compiled to scalar ones (round 5: the kernel is built without packed instructions) — the same few roundings of
    # log(sigma) taken in another order
    assert bool((err <= 2e-3 * scale + 6e-3).all()), float((err - 2e-3 * scale).max())     # (measured worst 5.3e-3)
    if P > 300:
        assert float(y[7]) == 0.0 and float(Jg.grad[7].abs().max()) == 0.0
        assert float(y[9]) < 1e-6


@pytest.mark.gpu
def test_def_regu_matches_the_torch_closed_form_the_loop_used():
    """Same term as the ~90-launch torch form it replaces (singular_values_3x3 + GMRobustError), value of the mean and gradient,
    on Jacobians of the size the loop sees."""
    from recmv import ops, utils
    from recmv.loop import singular_values_3x3
    g = torch.Generator().manual_seed(11)
    J = gpu(torch.eye(3).view(1, 3, 3) + 0.2 * torch.randn(6000, 3, 3, generator=g))
    a = J.clone().requires_grad_(True)
    b = J.clone().requires_grad_(True)
    la = ops.def_regu(a, 0.01).mean()
    s = torch.log(singular_values_3x3(b))
    lb = utils.GMRobustError((s * s).sum(1), 0.01, True).mean()
    la.backward()
    lb.backward()
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    assert float((a.grad - b.grad).abs().max()) <= 2e-3 * float(b.grad.abs().max())


# ------------------------------------------------------------------------------------------ bf16x6, weights split once
@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(20000, 512, 512), (70001, 473, 512), (33000, 512, 168), (16400, 512, 40)])
def test_bf16x6_presplit_weights_are_bit_identical_to_the_in_loop_split(bf16x6_mode, M, N, K):
    """recmv_b3_split writes the SAME pieces the staging path of gemm_nt_b3_kernel forms in the loop (one split8), and the PRE
    variant multiplies them in the same order: registering a weight matrix changes the time, not one bit of the product —
    plain product, fused bias + softplus epilogue, the activation-gradient operand product, ragged M / N, a K tail, a weight
    registered with more rows than the product uses; forgetting the entry goes back to the in-loop split."""
    from recmv import _lib as L, ops
    g = torch.Generator().manual_seed(M + N)
    A = gpu(torch.randn(M, K, generator=g))
    W = gpu(torch.randn(N, K, generator=g) / np.sqrt(K))
    b = gpu(torch.randn(N, generator=g))
    Y = gpu(torch.randn(M, K, generator=g))
    base = [ops.gemm_nt(A, W), ops.gemm_nt(A, W, b, ops.ACT_SOFTPLUS, 100.0, 0.70710678), ops.gemm_nt(A, W[:N - 5])]
    assert getattr(W, "_recmv_b3", None) is None
    ops.presplit(W)
    assert W._recmv_b3 is not None
    again = [ops.gemm_nt(A, W), ops.gemm_nt(A, W, b, ops.ACT_SOFTPLUS, 100.0, 0.70710678), ops.gemm_nt(A, W[:N - 5])]
    for x, y in zip(base, again):
        assert torch.equal(x, y)
    L.check(L.lib().recmv_b3_forget(L.ptr(W)), "forget")
    assert torch.equal(ops.gemm_nt(A, W), base[0])


@pytest.mark.gpu
def test_bf16x6_presplit_rides_on_the_weight_norm_and_its_transpose(bf16x6_mode):
    """In the bf16x6 mode every weight version gets its planes where it is made: the weight normalisation's output and the cached
    transpose; the registration ends with the tensor (a new tensor at the same address starts unregistered)."""
    from recmv import ops
    g = torch.Generator().manual_seed(2)
    v = gpu(torch.randn(512, 512, generator=g)).requires_grad_(True)
    gg = gpu(torch.rand(512, 1, generator=g) + 0.5).requires_grad_(True)
    W = ops.weight_norm(v, gg)
    assert getattr(W, "_recmv_b3", None) is not None
    Wt = ops.transposed(W)
    assert getattr(Wt, "_recmv_b3", None) is not None
    x = gpu(torch.randn(20000, 512, generator=g))
    y = ops.gemm_nt(x, W.detach())
    os.environ["RECMV_B3_PRESPLIT"] = "0"
    try:
        W2 = ops.weight_norm(v, gg)
        assert getattr(W2, "_recmv_b3", None) is None
        assert torch.equal(ops.gemm_nt(x, W2.detach()), y)
    finally:
        del os.environ["RECMV_B3_PRESPLIT"]


# ------------------------------------------------------------------------------------------ camera (csrc/camera.hip)
def _cameras_pair(monkeypatch, H=512, W=384, seed=0):
    """The same camera twice: on the fused kernels and on the torch expressions they replace (RECMV_FUSED_CAMERA=0)."""
    from recmv.model import RectifiedPerspectiveCameras
    g = torch.Generator().manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    R = gpu((q * torch.tensor([-1., -1., 1.])).view(1, 3, 3))

    def leaves():
        return (gpu(torch.tensor([[1000. * W / 512, 980. * H / 512]])).requires_grad_(True),
                gpu(torch.tensor([[W / 2. + 3., H / 2. - 5.]])).requires_grad_(True),
                gpu(torch.tensor([[0.03, -0.02, 3.0]])).requires_grad_(True))
    fa, ppa, Ta = leaves()
    fused = RectifiedPerspectiveCameras(fa, ppa, R, Ta, image_size=[(W, H)])
    assert fused._cam16[0] is not None
    fb, ppb, Tb = leaves()
    monkeypatch.setenv("RECMV_FUSED_CAMERA", "0")
    plain = RectifiedPerspectiveCameras(fb, ppb, R, Tb, image_size=[(W, H)])
    monkeypatch.delenv("RECMV_FUSED_CAMERA")
    assert plain._cam16[0] is None
    return fused, (fa, ppa, Ta), plain, (fb, ppb, Tb)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [0, 1, 257, 70001, 270054])
def test_camera_kernels_match_the_torch_expressions(P, monkeypatch):
    """csrc/camera.hip against the element-wise expressions of RectifiedPerspectiveCameras (model/CameraMine.py:62-88, :104-142,
    :146-169, `project`): NDC / screen / pixel coordinates and rays to a few ulp (same operation order, no contraction), the
    gradients to the points to 1e-5 relative and the fixed-order sums for translation / focal length / principal point to 1e-4 of
    their scale; twice in a row bit-identical."""
    fused, la, plain, lb = _cameras_pair(monkeypatch)
    g = torch.Generator().manual_seed(P + 1)
    pts = gpu(torch.randn(P, 3, generator=g) * 0.4)
    for name in ("transform_points_ndc", "transform_points_screen", "project"):
        pa, pb = pts.clone().requires_grad_(True), pts.clone().requires_grad_(True)
        oa, ob = getattr(fused, name)(pa), getattr(plain, name)(pb)
        assert oa.shape == ob.shape
        if P == 0:
            continue
        torch.testing.assert_close(oa, ob, rtol=2e-6, atol=2e-6 * float(ob.abs().max()))
        w = gpu(torch.randn(ob.shape, generator=g))
        ga = torch.autograd.grad((oa * w).sum(), [pa, *la])
        gb = torch.autograd.grad((ob * w).sum(), [pb, *lb])
        torch.testing.assert_close(ga[0], gb[0], rtol=1e-5, atol=1e-5 * float(gb[0].abs().max()))
        for x, y in zip(ga[1:], gb[1:]):
            assert x.shape == y.shape
            torch.testing.assert_close(x, y, rtol=2e-4, atol=2e-4 * max(float(y.abs().max()), 1e-3))
        ga2 = torch.autograd.grad((getattr(fused, name)(pa) * w).sum(), [pa, *la])
        assert all(torch.equal(x, y) for x, y in zip(ga, ga2)), "the camera gradients are sums in a fixed order"
    # [N,P,3] input of transform_points_screen keeps its shape
    if P >= 257:
        q3 = pts[:256].view(2, 128, 3)
        torch.testing.assert_close(fused.transform_points_screen(q3), plain.transform_points_screen(q3), rtol=2e-6, atol=1e-4)
    # rays: integer pixels and the float [P,3] form
    col = torch.randint(0, 384, (P,), generator=g).to("cuda")
    row = torch.randint(0, 512, (P,), generator=g).to("cuda")
    ra, rb = fused.view_rays_pix(col, row), plain.view_rays_pix(col, row)
    pix = torch.stack([col, row, torch.ones_like(col)], dim=1).float()
    rc = fused.view_rays(pix)
    assert ra.shape == (P, 3)
    if P:
        torch.testing.assert_close(ra, rb, rtol=2e-6, atol=2e-7)
        assert torch.equal(ra, rc)
        w = gpu(torch.randn(P, 3, generator=g))
        ga = torch.autograd.grad((ra * w).sum(), la[:2])
        gb = torch.autograd.grad((rb * w).sum(), lb[:2])
        for x, y in zip(ga, gb):
            torch.testing.assert_close(x, y, rtol=2e-4, atol=2e-4 * max(float(y.abs().max()), 1e-6))


@pytest.mark.gpu
def test_camera_kernels_argument_errors():
    import ctypes as C
    from recmv import _lib as L
    lib = L.lib()
    x = gpu(torch.zeros(4, 3))
    cam = gpu(torch.zeros(16))
    assert lib.recmv_cam_project(L.ptr(x), 4, L.ptr(cam), 64.0, 64.0, 3, L.ptr(x), None) == -1
    assert lib.recmv_cam_project(L.ptr(x), 4, None, 64.0, 64.0, 0, L.ptr(x), None) == -1
    assert lib.recmv_cam_project_backward(L.ptr(x), L.ptr(x), 4, L.ptr(cam), 64.0, 64.0, 0, None, L.ptr(cam), L.ptr(cam), 1, None) == -1
    assert lib.recmv_cam_rays(None, None, None, 4, L.ptr(cam), L.ptr(x), None) == -1
    assert lib.recmv_cam_partial_floats(10 ** 9) == 1024 * 7 and lib.recmv_cam_partial_floats(1) == 7
