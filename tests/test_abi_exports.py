"""The C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every symbol
that include/recmv_hip.h declares.  No compute calls here."""
import ctypes

import pytest


def test_library_builds_and_exports_all_declared_symbols():
    from recmv import _lib
    path = _lib.build()
    assert path.exists()
    lib = ctypes.CDLL(str(path))
    names = _lib.exported_symbols()
    assert len(names) >= 16
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in recmv_hip.h but not exported: {missing}"
    assert _lib.lib().recmv_abi_version() == _lib.ABI_VERSION


def test_argument_errors_do_not_need_a_gpu():
    """Argument validation happens before any HIP call: error codes + messages through the C ABI."""
    from recmv import _lib
    lib = _lib.lib()
    assert lib.recmv_inv3x3_forward(None, None, None, -1, 0, None) == -1
    assert b"n=-1" in lib.recmv_last_error()
    assert lib.recmv_inv3x3_forward(None, None, None, 0, 0, None) == 0          # empty batch is a no-op
    assert lib.recmv_mc_workspace_bytes(0, 4, 4) == 0
    assert lib.recmv_mc_workspace_bytes(257, 257, 257) > 0
    d = _lib.Tensor5()
    for i in range(5):
        d.size[i] = 2
        d.stride[i] = 1
    d.size[4] = 3
    # nearest interpolation / zero padding are rejected like GridSamplerMine.cpp:58-63
    assert lib.recmv_grid_sample3d_forward(None, d, None, d, None, d, 1, 1, 0, None) == -3
    assert b"Bilinear" in lib.recmv_last_error()
    assert lib.recmv_grid_sample3d_forward(None, d, None, d, None, d, 0, 0, 0, None) == -3
    assert b"Border" in lib.recmv_last_error()


def test_product_has_no_cpu_fallback():
    """Ops refuse CPU tensors (the reference's CHECK_CUDA) instead of silently computing on the host."""
    import torch
    from recmv import FastMinv, GridSamplerMine, MCGpu, interp2x_boundary3d, ops
    with pytest.raises(RuntimeError):
        FastMinv.Fast3x3Minv(torch.randn(4, 3, 3))
    with pytest.raises(RuntimeError):
        GridSamplerMine.forward(torch.randn(1, 2, 3, 3, 3), torch.zeros(1, 1, 1, 2, 3), 0, 1)
    with pytest.raises(RuntimeError):
        MCGpu.mc_gpu(torch.randn(4, 4, 4))
    with pytest.raises(RuntimeError):
        interp2x_boundary3d.forward(torch.randn(1, 1, 3, 3, 3), 0.0)
    with pytest.raises(RuntimeError):
        ops.gemm_nt(torch.randn(4, 4), torch.randn(4, 4))
    from recmv import raster
    first, num = torch.tensor([0]), torch.tensor([4])
    with pytest.raises(RuntimeError):
        raster.rasterize_meshes(torch.zeros(4, 3, 3), first, num, (8, 8))
    with pytest.raises(RuntimeError):
        raster.rasterize_points(torch.zeros(4, 3), first, num, (8, 8), 0.1, 4)
    with pytest.raises(RuntimeError):
        raster.alpha_composite(torch.zeros(1, 2, 2, 3, dtype=torch.int32), torch.zeros(1, 2, 2, 3), torch.zeros(1, 4))


def test_rasteriser_argument_errors_through_the_c_abi():
    from recmv import _lib
    lib = _lib.lib()
    assert lib.recmv_rasterize_meshes_workspace_bytes(3, 512, 512, 1000) == 3 * 512 * 512 * 8 + 1000 * 16 + 64
    assert lib.recmv_rasterize_meshes_workspace_bytes(-1, 8, 8, 0) == -1
    assert lib.recmv_rasterize_meshes(None, None, None, 1, 0, 0, 0, 8, 0.0, 1, 0, None, None, None, None, None, 0,
                                      None) == -1                                   # H = 0
    assert b"bad sizes" in lib.recmv_last_error()
    assert lib.recmv_rasterize_meshes(None, None, None, 1, 0, 0, 8, 8, -1.0, 1, 0, None, None, None, None, None, 0,
                                      None) == -1
    assert b"blur_radius" in lib.recmv_last_error()
    assert lib.recmv_rasterize_points_workspace_bytes(1, 8, 8, 10, -0.5) == -1
    assert lib.recmv_rasterize_points(None, None, None, 1, 10, 10, 8, 8, 0.0, 4, None, None, None, None, 0, None) == -1
    assert b"radius" in lib.recmv_last_error()
    assert lib.recmv_alpha_composite_forward(None, None, None, 1, 8, 8, 0, 1, 4, 0.0, None, None) == -1
