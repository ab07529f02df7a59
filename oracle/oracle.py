"""ctypes front-end of the CPU oracle (oracle/recmv_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product (rec-mv_amd/) never imports this module.

All functions take/return CPU torch tensors and mirror the Python-visible contract of the reference's
extension modules (FastMinv / GridSamplerMine / interp2x_boundary3d / MCGpu), SURVEY.md §8b.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liboracle.so"
_lib = None


def build(force: bool = False) -> Path:
    srcs = [_HERE / "recmv_oracle.c", _HERE / "oracle_impl.inc", _HERE / "mc_tables.inc"]
    if (not force and _LIB_PATH.exists()
            and all(_LIB_PATH.stat().st_mtime > s.stat().st_mtime for s in srcs)):
        return _LIB_PATH
    r = subprocess.run(["make", "-C", str(_HERE), "-B"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
    return _lib


def _suf(t: torch.Tensor) -> str:
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise RuntimeError("oracle: tensor must be float32 or float64")


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _i64(vals):
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])


def _cpu(t):
    assert t.device.type == "cpu", "oracle works on CPU tensors"
    return t


# ------------------------------------------------------------------------------------------- inv3x3
def inv3x3_forward(ms: torch.Tensor):
    ms = _cpu(ms).contiguous()
    n = ms.shape[0]
    invs = torch.zeros(n, 3, 3, dtype=ms.dtype)
    checks = torch.zeros(n, dtype=torch.uint8)
    getattr(lib(), "oracle_inv3x3_forward_" + _suf(ms))(_p(ms), _p(invs), _p(checks), C.c_int64(n))
    return [invs, checks.bool()]


def inv3x3_backward(grads: torch.Tensor, invs: torch.Tensor):
    grads = _cpu(grads).contiguous()
    invs = _cpu(invs).contiguous()
    assert grads.dtype == invs.dtype
    n = invs.shape[0]
    outs = torch.zeros(n, 3, 3, dtype=invs.dtype)
    getattr(lib(), "oracle_inv3x3_backward_" + _suf(invs))(_p(grads), _p(invs), _p(outs), C.c_int64(n))
    return outs


# --------------------------------------------------------------------------------------- grid sampler
def gs3d_forward(input: torch.Tensor, grid: torch.Tensor):
    _cpu(input), _cpu(grid)
    N, Cc = input.shape[0], input.shape[1]
    out = torch.empty(N, Cc, grid.shape[1], grid.shape[2], grid.shape[3], dtype=input.dtype)
    getattr(lib(), "oracle_gs3d_forward_" + _suf(input))(
        _p(input), _i64(input.shape), _i64(input.stride()), _p(grid), _i64(grid.shape), _i64(grid.stride()),
        _p(out), _i64(out.stride()))
    return out


def gs3d_backward(input, grid, grad_output, need_grad_input=True):
    _cpu(input), _cpu(grid), _cpu(grad_output)
    grad_input = torch.zeros_like(input) if need_grad_input else None
    grad_grid = torch.empty(grid.shape, dtype=grid.dtype)  # contiguous
    getattr(lib(), "oracle_gs3d_backward_" + _suf(input))(
        _p(input), _i64(input.shape), _i64(input.stride()), _p(grid), _i64(grid.shape), _i64(grid.stride()),
        _p(grad_output), _i64(grad_output.stride()), _p(grad_input),
        _i64(grad_input.stride() if grad_input is not None else input.stride()), _p(grad_grid))
    return grad_input, grad_grid


def gs3d_dbackward(ggI, ggG, input, grid, grad_output, need_grad_input=True):
    grad_input = torch.zeros_like(input) if need_grad_input else None
    grad_grid = torch.empty(grid.shape, dtype=grid.dtype)
    ggO = torch.empty(grad_output.shape, dtype=grad_output.dtype)
    getattr(lib(), "oracle_gs3d_dbackward_" + _suf(input))(
        _p(ggI), _i64(ggI.stride() if ggI is not None else input.stride()), _p(ggG), _i64(ggG.stride()),
        _p(input), _i64(input.shape), _i64(input.stride()), _p(grid), _i64(grid.shape), _i64(grid.stride()),
        _p(grad_output), _i64(grad_output.stride()), _p(grad_input),
        _i64(grad_input.stride() if grad_input is not None else input.stride()), _p(grad_grid), _p(ggO),
        _i64(ggO.stride()))
    return grad_input, grad_grid, ggO


class OracleGridSample3dBackwardFunction(torch.autograd.Function):
    """MCAcc/grid_sampler_mine.py:44-65, backed by the oracle."""

    @staticmethod
    def forward(ctx, input, grid, grad_output):
        ctx.save_for_backward(input, grid, grad_output)
        return gs3d_backward(input.detach(), grid.detach(), grad_output.detach())

    @staticmethod
    def backward(ctx, ggI, ggG):
        input, grid, grad_output = ctx.saved_tensors
        return gs3d_dbackward(ggI.contiguous(), ggG.contiguous(), input.detach(), grid.detach(),
                              grad_output.detach())


class OracleGridSample3dFunction(torch.autograd.Function):
    """MCAcc/grid_sampler_mine.py:8-42, backed by the oracle."""

    @staticmethod
    def forward(ctx, input, grid):
        ctx.save_for_backward(input, grid)
        return gs3d_forward(input.detach(), grid.detach())

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        return OracleGridSample3dBackwardFunction.apply(input, grid, grad_output)


# ------------------------------------------------------------------------------------------ interp2x
def interp2x_forward(input: torch.Tensor, balance_value: float):
    input = _cpu(input).contiguous()
    B, Cc, d, h, w = input.shape
    out = torch.empty(B, Cc, 2 * d - 1, 2 * h - 1, 2 * w - 1, dtype=input.dtype)
    bnd = torch.empty(out.shape, dtype=torch.uint8)
    getattr(lib(), "oracle_interp2x_forward_" + _suf(input))(
        _p(input), _p(out), _p(bnd), C.c_int64(B * Cc), C.c_int64(d), C.c_int64(h), C.c_int64(w),
        C.c_float(balance_value))
    return [out, bnd.bool()]


def interp2x_backward(grad_output: torch.Tensor):
    grad_output = _cpu(grad_output).contiguous()
    B, Cc, D, H, W = grad_output.shape
    gi = torch.empty(B, Cc, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2, dtype=grad_output.dtype)
    getattr(lib(), "oracle_interp2x_backward_" + _suf(grad_output))(
        _p(grad_output), _p(gi), C.c_int64(B * Cc), C.c_int64(D), C.c_int64(H), C.c_int64(W))
    return gi


# ------------------------------------------------------------------------------------ marching cubes
def mc(sdfs: torch.Tensor, xstep=1.0, ystep=1.0, zstep=1.0, xmin=0.0, ymin=0.0, zmin=0.0, fTargetValue=0.0):
    """MCGpu.mc_gpu contract (MCGpu/MCGpu.cpp:20-56), canonical deterministic order."""
    sdfs = _cpu(sdfs).contiguous()
    assert sdfs.dtype == torch.float32 and sdfs.dim() == 3
    nx, ny, nz = sdfs.shape
    counts = (C.c_int64 * 2)(0, 0)
    f = lib().oracle_mc
    args = [_p(sdfs), C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_float(fTargetValue), C.c_float(xstep),
            C.c_float(ystep), C.c_float(zstep), C.c_float(xmin), C.c_float(ymin), C.c_float(zmin)]
    rc = f(*args, C.c_void_p(0), C.c_void_p(0), counts)
    if rc != 0:
        return []
    verts = torch.zeros(counts[0], 3, dtype=torch.float32)
    faces = torch.zeros(counts[1], 3, dtype=torch.int64)
    rc = f(*args, _p(verts), _p(faces), counts)
    assert rc == 0
    return [verts, faces]


# ------------------------------------------------------------------------------------ mesh rasteriser
def rasterize_meshes(face_verts, mesh_first_face, mesh_num_faces, image_size, blur_radius=0.0,
                     perspective_correct=True, cull_backfaces=False, scan=False):
    """pytorch3d `rasterize_meshes(..., faces_per_pixel=1)` contract on packed NDC face vertices [F,3,3]:
    returns (pix_to_face [N,H,W,1] int64, zbuf [N,H,W,1], bary_coords [N,H,W,1,3], dists [N,H,W,1]); -1 = empty.
    scan=True runs the face-ordered CPU port (same answer; the variant bench.py's cpu_baseline times)."""
    fv = _cpu(face_verts).contiguous().float()
    first = _cpu(mesh_first_face).contiguous().long()
    num = _cpu(mesh_num_faces).contiguous().long()
    H, W = image_size
    N = first.numel()
    p2f = torch.empty(N, H, W, 1, dtype=torch.int64)
    zbuf = torch.empty(N, H, W, 1, dtype=torch.float32)
    bary = torch.empty(N, H, W, 1, 3, dtype=torch.float32)
    dists = torch.empty(N, H, W, 1, dtype=torch.float32)
    fn = lib().oracle_rasterize_meshes_scan if scan else lib().oracle_rasterize_meshes
    rc = fn(_p(fv), _p(first), _p(num), C.c_int64(N), C.c_int64(H), C.c_int64(W),
                                       C.c_float(blur_radius), C.c_int(int(perspective_correct)),
                                       C.c_int(int(cull_backfaces)), _p(p2f), _p(zbuf), _p(bary), _p(dists))
    assert rc == 0
    return p2f, zbuf, bary, dists


# ------------------------------------------------------------------------------------ point rasteriser
def rasterize_points(points, cloud_first, cloud_num, image_size, radius, points_per_pixel, scan=False):
    """pytorch3d `rasterize_points` contract on packed NDC points [P,3]: (idx [N,H,W,K] int32, zbuf, dists)."""
    pts = _cpu(points).contiguous().float()
    first, num = _cpu(cloud_first).contiguous().long(), _cpu(cloud_num).contiguous().long()
    H, W = image_size
    N, K = first.numel(), int(points_per_pixel)
    idx = torch.empty(N, H, W, K, dtype=torch.int32)
    zbuf = torch.empty(N, H, W, K, dtype=torch.float32)
    dists = torch.empty(N, H, W, K, dtype=torch.float32)
    fn = lib().oracle_rasterize_points_scan if scan else lib().oracle_rasterize_points
    rc = fn(_p(pts), _p(first), _p(num), C.c_int64(N), C.c_int64(H), C.c_int64(W), C.c_float(radius), C.c_int(K),
            _p(idx), _p(zbuf), _p(dists))
    assert rc == 0
    return idx, zbuf, dists


def rasterize_points_backward(points, idx, grad_dists, grad_zbuf=None):
    pts = _cpu(points).contiguous().float()
    N, H, W, K = idx.shape
    out = torch.empty_like(pts)
    gd = grad_dists.contiguous() if grad_dists is not None else None
    gz = grad_zbuf.contiguous() if grad_zbuf is not None else None
    rc = lib().oracle_rasterize_points_backward(_p(pts), _p(idx.contiguous()), _p(gd), _p(gz), C.c_int64(N),
                                                C.c_int64(pts.shape[0]), C.c_int64(H), C.c_int64(W), C.c_int(K),
                                                _p(out))
    assert rc == 0
    return out


def alpha_composite_forward(idx, alphas, features):
    """idx / alphas [N,H,W,K], features [C,P] -> images [N,C,H,W]."""
    N, H, W, K = idx.shape
    Cc, P = features.shape
    images = torch.empty(N, Cc, H, W, dtype=torch.float32)
    rc = lib().oracle_alpha_composite_forward(_p(idx.contiguous()), _p(alphas.contiguous()), _p(features.contiguous()),
                                              C.c_int64(N), C.c_int64(H), C.c_int64(W), C.c_int(K), C.c_int64(Cc),
                                              C.c_int64(P), _p(images))
    assert rc == 0
    return images


def alpha_composite_backward(idx, alphas, features, grad_images, need_grad_features=True):
    N, H, W, K = idx.shape
    Cc, P = features.shape
    ga = torch.empty(N, H, W, K, dtype=torch.float32)
    gf = torch.empty(Cc, P, dtype=torch.float32) if need_grad_features else None
    rc = lib().oracle_alpha_composite_backward(_p(idx.contiguous()), _p(alphas.contiguous()),
                                               _p(features.contiguous()), _p(grad_images.contiguous()), C.c_int64(N),
                                               C.c_int64(H), C.c_int64(W), C.c_int(K), C.c_int64(Cc), C.c_int64(P),
                                               _p(ga), _p(gf))
    assert rc == 0
    return ga, gf


# ------------------------------------------------------------------------------- deformation regulariser
def def_regu(J: torch.Tensor, c: float):
    """(y [P], dy/dJ [P,3,3]) of the reference's deformation regulariser, the way the reference computes it
    (OptimGarmentNetwork.py:1143-1155): host torch.svd of the Jacobians, log of the singular values, sum of squares through
    utils.GMRobustError(x, c, True) = 2 x / c^2 / (x / c^2 + 4) (utils/utils.py:87-91); the gradient by autograd through the SVD.
    Evaluated in float64 whatever the input precision (the checker of recmv_def_regu)."""
    Jd = _cpu(J).double().detach().requires_grad_(True)
    _, s, _ = torch.svd(Jd)
    s = torch.log(s)
    x = (s * s).sum(1)
    y = 2. * x / (c * c) / (x / (c * c) + 4)
    g, = torch.autograd.grad(y.sum(), Jd)
    return y.detach(), g
