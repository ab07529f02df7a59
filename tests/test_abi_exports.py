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


def test_the_regulariser_kernel_has_no_packed_f32_instructions():
    """Round 5 traced the run-to-run divergence of the bf16x6 matrix mode to packed-f32 VALU instructions: beside that mode's NT product
    kernels a wave executing v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 gets wrong results in lanes 48-63 (tools/erratum/valu_disturb_repro.hip,
    DESIGN.md §9).  rec-mv_amd/build.py therefore builds csrc/def_regu.hip — the kernel that was caught — without them in every build,
    and EVERY kernel without them under RECMV_NO_PACKED_F32=1 (the build for RECMV_GEMM_MODE=1).  Disassembles the gfx950 code
    objects inside the shared library."""
    import os
    import re
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin/"
    if not all(os.path.exists(llvm + t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")):
        pytest.skip("ROCm LLVM tools not present")
    from recmv import _lib
    so = str(_lib.build())
    per_kernel = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([llvm + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
        for i, st in enumerate(starts):
            part, co = os.path.join(d, "b%d.bin" % i), os.path.join(d, "b%d.co" % i)
            with open(part, "wb") as fh:
                fh.write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            r = subprocess.run([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
            if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            kern = None
            for line in subprocess.run([llvm + "llvm-objdump", "-d", co], capture_output=True, text=True).stdout.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    kern = m.group(1)
                    per_kernel.setdefault(kern, 0)
                elif kern and re.search(r"v_pk_(?:mul|fma|add)_f32", line):
                    per_kernel[kern] += 1
    regu = [k for k in per_kernel if "def_regu_kernel" in k]
    assert regu and all(per_kernel[k] == 0 for k in regu), {k: per_kernel[k] for k in regu}
    if os.environ.get("RECMV_NO_PACKED_F32") == "1":
        assert sum(per_kernel.values()) == 0, {k: v for k, v in per_kernel.items() if v}
