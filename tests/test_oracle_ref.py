"""oracle/_ref — the REFERENCE's own kernels (MCGpu/CudaKernels.cu:4-521, FastMinv/Matrix3x3InvKernels.cu:18-104,
MCAcc/cuda/GridSamplerMineKernel.cu:29-914, MCAcc/cuda/interp2x_boundary3d_kernel.cu:8-240), compiled for the host from the
reference tree (oracle/Makefile `ref`, oracle/ref/*.cpp + cuda_host_shim.h / torch_host_shim.h), against the C restatement
oracle/recmv_oracle.c.  This is what pins the oracle's marching cubes, 3x3 inverse, grid sampler (forward, backward, double
backward) and 2x boundary upsampler to the reference itself (no golden mesh exists in the reference).

Comparison rule for MC: the reference hands out vertex / face ids with atomics, so its order is
arbitrary; its output is put into the canonical order — vertices by ascending lattice-edge key, read from the
reference's own edge->vertex table — and must then equal the oracle's output bit for bit: vertex positions (f32
bits), face corner ids (incl. the -1 of edges without an owner voxel) and, for the serial index-order run, the face
order (voxel order, case-table triangle order).  A scrambled run (other "atomic" order) must canonicalise to the same
mesh.
"""
import numpy as np
import pytest
import torch

from oracle import ref as R

pytestmark = pytest.mark.skipif(not R.build(), reason="oracle/_ref not built and /root/reference not mounted")


def _volumes():
    g = torch.Generator().manual_seed(0)
    vols = []
    for i in range(10):                                               # the 10 noise volumes of the GPU MC test
        shape = [(9, 11, 7), (16, 16, 16), (5, 33, 12), (20, 9, 31), (2, 2, 2), (13, 13, 13), (33, 5, 7),
                 (8, 24, 16), (17, 19, 23), (3, 40, 3)][i]
        vols.append(("noise%d" % i, torch.randn(*shape, generator=g).contiguous(), 0.1 * (i % 3 - 1)))
    n = 33
    ax = torch.linspace(-1, 1, n)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    vols.append(("sphere", (torch.sqrt((X - 0.03) ** 2 + (Y + 0.02) ** 2 + (Z - 0.01) ** 2) - 0.6).float().contiguous(), 0.0))
    n = 9
    ax = torch.linspace(-1, 1, n)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    vols.append(("box_touching", (torch.sqrt(X ** 2 + Y ** 2 + Z ** 2) - 1.2).float().contiguous(), 0.0))
    q = torch.round(torch.randn(12, 12, 12, generator=g) * 2) / 2                    # many exact ties with the iso value
    vols.append(("ties", q.contiguous(), 0.5))
    return vols


SCALES = [(1.0, 1.0, 1.0, 0.0, 0.0, 0.0), (2.0 / 32, 0.0371, 0.113, -1.0, 0.37, -2.2)]


@pytest.mark.parametrize("name,vol,iso", _volumes(), ids=[v[0] for v in _volumes()])
def test_oracle_mc_equals_the_reference_kernels_after_canonicalisation(oracle, name, vol, iso):
    for sc in SCALES:
        v_o, f_o = oracle.mc(vol, *sc, iso)
        v_r, f_r, state, _ = R.mc_gpu(vol, *sc, iso, fma=True, return_edge_state=True)
        assert v_r.shape == v_o.shape and f_r.shape == f_o.shape, (name, v_r.shape, v_o.shape, f_r.shape, f_o.shape)
        cv, cf, keys = R.canonical(v_r, f_r, state)
        assert torch.equal(cv.view(torch.int32), v_o.view(torch.int32)), name          # f32 bit patterns
        assert torch.equal(cf, f_o), name                                            # ids, winding AND face order
        # another id-assignment order (what the atomics of a GPU run would give): same canonical mesh
        v_s, f_s, state_s, _ = R.mc_gpu(vol, *sc, iso, fma=True, scramble=7, return_edge_state=True)
        if v_s.shape[0] > 3 and name != "noise4":
            assert not torch.equal(v_s, v_r), "the scrambled run should hand out ids in another order"
        cvs, cfs, _ = R.canonical(v_s, f_s, state_s)
        assert torch.equal(cvs.view(torch.int32), v_o.view(torch.int32))
        assert torch.equal(R.sorted_faces(cfs), R.sorted_faces(f_o))


def test_unit_spacing_needs_no_fma_assumption(oracle):
    """With step 1 / origin 0 the scaling `v*step+min` is exact either way: the un-contracted build of the reference
    kernels gives the same bits, so the lattice-space positions (d_fGetOffset, edge origin + t*direction incl. the
    `1 - t` of the y edges) are pinned independently of how nvcc contracts the scaling."""
    g = torch.Generator().manual_seed(3)
    vol = torch.randn(14, 15, 16, generator=g)
    v_o, f_o = oracle.mc(vol)
    v_r, f_r, state, _ = R.mc_gpu(vol, fma=False, return_edge_state=True)
    cv, cf, _ = R.canonical(v_r, f_r, state)
    assert torch.equal(cv.view(torch.int32), v_o.view(torch.int32)) and torch.equal(cf, f_o)
    # general spacing: contracted (nvcc default -fmad=true) and un-contracted builds differ by at most one ulp
    sc = SCALES[1]
    a = R.mc_gpu(vol, *sc, 0.0, fma=True)[0]
    b = R.mc_gpu(vol, *sc, 0.0, fma=False)[0]
    ulp = (a.view(torch.int32) - b.view(torch.int32)).abs().max()
    assert int(ulp) <= 1


def test_reference_capacity_note_and_degenerate_inputs(oracle):
    assert R.mc_gpu(torch.ones(5, 6, 7))[0].shape == (0, 3)
    g = torch.Generator().manual_seed(1)
    dense = torch.randn(8, 8, 8, generator=g)
    *_, over = R.mc_gpu(dense, return_edge_state=True)
    assert over, "white noise exceeds the 5 % buffers the reference allocates (CudaKernels.cu:590-592)"
    n = 33
    ax = torch.linspace(-1, 1, n)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    *_, over = R.mc_gpu((torch.sqrt(X ** 2 + Y ** 2 + Z ** 2) - 0.6).float().contiguous(), return_edge_state=True)
    assert not over


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_oracle_inv3x3_equals_the_reference_kernels(oracle, dtype):
    g = torch.Generator().manual_seed(5)
    for n in (1, 10000, 35937):                                        # FastMinv/check.py:7 and the C1 grid size
        ms = torch.randn(n, 3, 3, generator=g, dtype=dtype)
        if n > 100:                                                    # adversarial rows: det = 0 and |det| around 1e-4
            ms[0] = 0
            ms[1] = torch.eye(3, dtype=dtype) * 0.0464                 # det ~ 9.99e-5  -> singular by the rule
            ms[2] = torch.eye(3, dtype=dtype) * 0.04642                # det ~ 1.0002e-4 -> regular
            ms[3, 2] = ms[3, 0] * 2                                    # exactly dependent rows
        inv_o, chk_o = oracle.inv3x3_forward(ms)
        inv_r, chk_r = R.inv3x3_forward(ms)
        assert torch.equal(chk_o, chk_r)
        assert torch.equal(inv_o.view(torch.int64 if dtype == torch.float64 else torch.int32),
                           inv_r.view(torch.int64 if dtype == torch.float64 else torch.int32))
        grads = torch.randn(n, 3, 3, generator=g, dtype=dtype)
        out_o = oracle.inv3x3_backward(grads, inv_o)
        out_r = R.inv3x3_backward(grads, inv_r)
        assert torch.equal(out_o, out_r)


# ------------------------------------------------------------------------------------------------ sampler, upsampler
def _sampler_cases(dtype):
    """(input, grid): the reference's own check shape (MCAcc/check_grid_sampler_mine.py:5-16), a channels-last skinning-like volume
    with a point list (what LBSkinner passes), several batch items with a 3-D output lattice, coordinates beyond [-1, 1]."""
    g = torch.Generator().manual_seed(11)
    cases = [(torch.randn(1, 5, 15, 15, 15, generator=g, dtype=dtype), (torch.rand(1, 1, 1, 10, 3, generator=g, dtype=dtype) - 0.5) * 2.2),
             (torch.randn(1, 24, 5, 9, 7, generator=g, dtype=dtype).contiguous(memory_format=torch.channels_last_3d),
              (torch.rand(1, 1, 1, 300, 3, generator=g, dtype=dtype) - 0.5) * 2.2),
             (torch.randn(2, 5, 7, 9, 8, generator=g, dtype=dtype), (torch.rand(2, 3, 4, 50, 3, generator=g, dtype=dtype) - 0.5) * 2.6),
             (torch.randn(1, 3, 2, 2, 2, generator=g, dtype=dtype), (torch.rand(1, 1, 1, 64, 3, generator=g, dtype=dtype) - 0.5) * 4.0)]
    edge = torch.tensor([[-1., -1., -1.], [1., 1., 1.], [0., 0., 0.], [1., -1., 0.5], [-1.0000001, 0.9999999, 0.]], dtype=dtype)
    cases.append((torch.randn(1, 4, 3, 4, 5, generator=g, dtype=dtype), edge.view(1, 1, 1, -1, 3)))
    return cases


def test_oracle_sampler_f32_equals_the_reference_kernels(oracle):
    """GridSamplerMineKernel.cu:29-914 compiled for the host, f32 (the loop's type).  Forward: bit-equal to the kernel text with
    `a*b+c` contracted (nvcc's default, the accumulation `out += value * weight` becomes an fma — the chain the oracle and the
    HIP kernel spell out).  Backward and double backward: bit-equal to the kernel text WITHOUT contraction — all outputs,
    incl. grad_input accumulated in index order; against the contracted build they differ by rounding only (which product of
    a four-factor term a compiler fuses is its own choice; nvcc's is not reproducible on the host)."""
    g = torch.Generator().manual_seed(12)
    for inp, grid in _sampler_cases(torch.float32):
        out_o = oracle.gs3d_forward(inp, grid)
        assert torch.equal(out_o, R.gs3d_forward(inp, grid, fma=True))
        torch.testing.assert_close(out_o, R.gs3d_forward(inp, grid, fma=False), rtol=1e-6, atol=1e-6)     # rounding only
        go = torch.randn(out_o.shape, generator=g)
        gi_o, gg_o = oracle.gs3d_backward(inp, grid, go)
        gi_r, gg_r = R.gs3d_backward(inp, grid, go, fma=False)
        assert torch.equal(gi_o, gi_r) and torch.equal(gg_o, gg_r)
        gi_f, gg_f = R.gs3d_backward(inp, grid, go, fma=True)
        torch.testing.assert_close(gi_o, gi_f, rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(gg_o, gg_f, rtol=1e-5, atol=1e-5 * float(gg_o.abs().max()))
        gi_s, gg_s = R.gs3d_backward(inp, grid, go, fma=False, scramble=7)             # another order of the atomicAdds
        assert torch.equal(gg_s, gg_r)
        torch.testing.assert_close(gi_s, gi_r, rtol=1e-5, atol=1e-5)
        ggI = torch.randn(inp.shape, generator=g).contiguous(memory_format=torch.channels_last_3d if inp.shape[1] == 24
                                                                  else torch.contiguous_format)
        ggG = torch.randn(grid.shape, generator=g)
        for got, want in zip(oracle.gs3d_dbackward(ggI, ggG, inp, grid, go), R.gs3d_dbackward(ggI, ggG, inp, grid, go, fma=False)):
            assert torch.equal(got, want)
        for got, want in zip(oracle.gs3d_dbackward(ggI, ggG, inp, grid, go), R.gs3d_dbackward(ggI, ggG, inp, grid, go, fma=True)):
            torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5 * max(float(want.abs().max()), 1.0))


def test_oracle_sampler_f64_agrees_with_the_reference_kernels(oracle):
    """f64 (gradcheck only).  Backward / double backward: bit-equal to the uncontracted kernel text, as in f32.  Forward: within
    an ulp or two — in f64 the unnormalisation `((x + 1) * W - 1) / 2` is contractable too (in f32 its product is rounded to
    f32 before the double-precision subtraction), and the oracle contracts the accumulation only."""
    g = torch.Generator().manual_seed(13)
    for inp, grid in _sampler_cases(torch.float64):
        out_o = oracle.gs3d_forward(inp, grid)
        torch.testing.assert_close(out_o, R.gs3d_forward(inp, grid, fma=False), rtol=1e-14, atol=1e-14)
        torch.testing.assert_close(out_o, R.gs3d_forward(inp, grid, fma=True), rtol=1e-13, atol=1e-13)
        go = torch.randn(out_o.shape, generator=g, dtype=torch.float64)
        for got, want in zip(oracle.gs3d_backward(inp, grid, go), R.gs3d_backward(inp, grid, go, fma=False)):
            assert torch.equal(got, want)
        ggI = torch.randn(inp.shape, generator=g, dtype=torch.float64)
        ggG = torch.randn(grid.shape, generator=g, dtype=torch.float64)
        for got, want in zip(oracle.gs3d_dbackward(ggI, ggG, inp, grid, go), R.gs3d_dbackward(ggI, ggG, inp, grid, go, fma=False)):
            assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_oracle_interp2x_equals_the_reference_kernels(oracle, dtype):
    """interp2x_boundary3d_kernel.cu:8-240 compiled for the host: upsampled values, boundary flags and the backward, bit for
    bit — incl. values sitting exactly on the balance value and single-voxel axes."""
    g = torch.Generator().manual_seed(14)
    for shape, balance in (((1, 1, 5, 6, 4), 0.0), ((2, 3, 3, 2, 7), 0.3), ((1, 1, 1, 4, 4), 0.5), ((1, 2, 9, 9, 9), -0.2)):
        x = torch.randn(*shape, generator=g, dtype=dtype)
        x.view(-1)[::7] = balance                                               # exact ties with the balance value
        out_o, bnd_o = oracle.interp2x_forward(x, balance)
        out_r, bnd_r = R.interp2x_forward(x, balance)
        assert torch.equal(out_o, out_r) and torch.equal(bnd_o, bnd_r)
        go = torch.randn(out_o.shape, generator=g, dtype=dtype)
        assert torch.equal(oracle.interp2x_backward(go), R.interp2x_backward(go))
