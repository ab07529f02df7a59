"""ctypes front-end of oracle/_ref: the REFERENCE's own kernels compiled for the host (oracle/Makefile `ref`).

TEST INFRASTRUCTURE: imported only by tests/ and by __graft_entry__.build() (which builds it — building the checker is
not using it).  The product (rec-mv_amd/) never imports this module.

`oracle/_ref/*.so` are built in the development container from the sources under /root/reference (read-only) and
travel to the GPU box as built files; `available()` tells whether they are there.  Functions mirror the contracts of
`MCGpu.mc_gpu` (MCGpu/MCGpu.cpp:20-56), `FastMinv.Fast3x3Minv(_backward)` (FastMinv/M3x3Inv.cpp:12-59),
`GridSamplerMine.forward / backward / dbackward` (MCAcc/cuda/GridSamplerMineKernel.cu:917-1022) and
`interp2x_boundary3d.forward / backward` (MCAcc/cuda/interp2x_boundary3d_kernel.cu:243-304) on CPU tensors, plus
`canonical()` = the canonical MC ordering of SURVEY.md §8a-E applied to a reference run.
"""
from __future__ import annotations

import ctypes as C
import math
import subprocess
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
_REF = _HERE / "_ref"
REFERENCE = Path("/root/reference")
_LIBS = {"mc_fma": "libref_mc_fma.so", "mc_nofma": "libref_mc_nofma.so", "minv": "libref_minv.so",
         "gs3d_fma": "libref_gs3d_fma.so", "gs3d_nofma": "libref_gs3d_nofma.so", "interp2x": "libref_interp2x.so"}
_loaded = {}


def available() -> bool:
    return all((_REF / n).exists() for n in _LIBS.values())


def build(force: bool = False) -> bool:
    """Build oracle/_ref from the reference tree when it is mounted (development container); keep prebuilt files
    otherwise (GPU box).  Returns `available()`."""
    if REFERENCE.exists():
        cmd = ["make", "-C", str(_HERE), "ref"] + (["-B"] if force else [])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle/_ref build failed:\n" + r.stdout + r.stderr)
    return available()


def _lib(name):
    if name not in _loaded:
        if not (_REF / _LIBS[name]).exists():
            build()
        _loaded[name] = C.CDLL(str(_REF / _LIBS[name]))
    return _loaded[name]


def _p(t):
    return C.c_void_p(t.data_ptr())


def _coprime_stride(n: int, seed: int) -> int:
    s = (seed * 2654435761 + 12345) % max(n, 2)
    s = max(s, 2)
    while math.gcd(s, n) != 1:
        s += 1
    return s


def mc_gpu(sdfs: torch.Tensor, xstep=1.0, ystep=1.0, zstep=1.0, xmin=0.0, ymin=0.0, zmin=0.0, fTargetValue=0.0,
           fma: bool = True, scramble: int = 0, return_edge_state: bool = False):
    """The reference's MC kernels (CudaKernels.cu:316-521) run serially on the host.  `scramble` != 0 visits the voxels
    in a scrambled order (the order in which the reference's atomics hand out vertex / face ids on a GPU is arbitrary);
    `fma` picks the build whose `v*step+min` is contracted like nvcc's default (-fmad=true)."""
    sdfs = sdfs.contiguous()
    assert sdfs.dtype == torch.float32 and sdfs.dim() == 3 and sdfs.device.type == "cpu"
    L = _lib("mc_fma" if fma else "mc_nofma")
    L.ref_mc_run.restype = C.c_void_p
    nx, ny, nz = sdfs.shape
    n = nx * ny * nz
    L.ref_mc_set_loop_stride(C.c_long(_coprime_stride(n, scramble) if scramble else 1))
    nv, nf, over = C.c_int(0), C.c_int(0), C.c_int(0)
    ctx = L.ref_mc_run(_p(sdfs), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_float(fTargetValue), C.byref(nv),
                       C.byref(nf), C.byref(over))
    L.ref_mc_set_loop_stride(C.c_long(1))
    if not ctx:
        return []
    verts = torch.zeros(nv.value, 3, dtype=torch.float32)
    faces = torch.zeros(nf.value, 3, dtype=torch.int64)
    state = torch.zeros(n * 3, dtype=torch.int32) if return_edge_state else None
    L.ref_mc_fetch(C.c_void_p(ctx), C.c_float(xstep), C.c_float(ystep), C.c_float(zstep), C.c_float(xmin),
                   C.c_float(ymin), C.c_float(zmin), _p(verts), _p(faces),
                   _p(state) if state is not None else C.c_void_p(0))
    L.ref_mc_free(C.c_void_p(ctx))
    out = [verts, faces]
    if return_edge_state:
        out += [state, bool(over.value)]
    return out


def canonical(verts, faces, edge_state):
    """SURVEY.md §8a-E canonical order applied to a reference run: vertices by ascending lattice-edge key
    ((x*NY+y)*NZ+z)*3+dir — read off the reference's own edge->vertex table, no geometry involved —, face corners
    renumbered accordingly.  Faces keep the order of the run (= voxel order, triangle order for the serial index-order
    run); `sorted_faces()` gives an order-free view for scrambled runs."""
    keys = torch.nonzero(edge_state >= 0, as_tuple=True)[0]            # ascending edge keys that carry a vertex
    old_ids = edge_state[keys].long()
    assert old_ids.numel() == verts.shape[0] and torch.equal(torch.sort(old_ids).values, torch.arange(verts.shape[0]))
    new_of_old = torch.empty(verts.shape[0], dtype=torch.int64)
    new_of_old[old_ids] = torch.arange(verts.shape[0])
    cverts = verts[old_ids]
    cfaces = torch.where(faces >= 0, new_of_old[faces.clamp(min=0)], faces)
    return cverts, cfaces, keys


def sorted_faces(faces):
    """Rows sorted lexicographically (corner order inside a row untouched)."""
    if faces.numel() == 0:
        return faces
    order = torch.arange(faces.shape[0])
    for col in (2, 1, 0):
        order = order[torch.sort(faces[order, col], stable=True).indices]
    return faces[order]


def inv3x3_forward(ms: torch.Tensor):
    ms = ms.contiguous()
    n = ms.shape[0]
    invs = torch.empty(n, 3, 3, dtype=ms.dtype)          # the reference allocates with empty_like (M3x3Inv.cpp:19)
    checks = torch.zeros(n, dtype=torch.bool)
    fn = {torch.float32: "ref_M3x3Inv_float", torch.float64: "ref_M3x3Inv_double"}[ms.dtype]
    getattr(_lib("minv"), fn)(_p(ms), _p(invs), _p(checks), C.c_int(n))
    return [invs, checks]


def inv3x3_backward(grads: torch.Tensor, invs: torch.Tensor):
    grads, invs = grads.contiguous(), invs.contiguous()
    outs = torch.empty_like(grads)
    fn = {torch.float32: "ref_M3x3Inv_backward_float", torch.float64: "ref_M3x3Inv_backward_double"}[grads.dtype]
    getattr(_lib("minv"), fn)(_p(grads), _p(invs), _p(outs), C.c_int(grads.shape[0]))
    return outs


# ------------------------------------------------------------------------------------ 3-D grid sampler
def _i32(values):
    return (C.c_int * 5)(*[int(v) for v in values])


def _desc(t):
    return _p(t), _i32(t.shape), _i32(t.stride())


def _gs3d(fma):
    return _lib("gs3d_fma" if fma else "gs3d_nofma")


def _suffix(t):
    return {torch.float32: "float", torch.float64: "double"}[t.dtype]


def gs3d_forward(input, grid, interp=0, pad=1, fma=True):
    """The reference's grid_sampler_3d_kernel (GridSamplerMineKernel.cu:160-328) run serially; tensors keep their strides."""
    out = torch.empty(input.shape[0], input.shape[1], grid.shape[1], grid.shape[2], grid.shape[3], dtype=input.dtype)
    getattr(_gs3d(fma), "ref_gs3d_forward_" + _suffix(input))(*_desc(input), *_desc(grid), *_desc(out), C.c_int(interp), C.c_int(pad))
    return out


def gs3d_backward(input, grid, grad_output, interp=0, pad=1, fma=True, scramble=0):
    """grid_sampler_3d_backward_kernel (:331-570): (grad_input zeros + scattered adds, grad_grid).  `scramble` != 0 visits the
    output locations in a scrambled order — the order in which a GPU's atomicAdd reaches grad_input is arbitrary."""
    gi, gg = torch.zeros_like(input), torch.empty(grid.shape, dtype=grid.dtype)
    L = _gs3d(fma)
    n = grid.shape[0] * grid.shape[1] * grid.shape[2] * grid.shape[3]
    L.ref_gs3d_set_loop_stride(C.c_long(_coprime_stride(n, scramble) if scramble else 1))
    getattr(L, "ref_gs3d_backward_" + _suffix(input))(*_desc(grad_output), *_desc(input), *_desc(grid), *_desc(gi), *_desc(gg),
                                                     C.c_int(interp), C.c_int(pad))
    L.ref_gs3d_set_loop_stride(C.c_long(1))
    return gi, gg


def gs3d_dbackward(ggI, ggG, input, grid, grad_output, interp=0, pad=1, fma=True):
    """grid_sampler_3d_backward_backward_kernel (:573-914): (grad_input, grad_grid, grad_grad_output)."""
    gi, gg = torch.zeros_like(input), torch.empty(grid.shape, dtype=grid.dtype)
    ggo = torch.zeros_like(grad_output)
    getattr(_gs3d(fma), "ref_gs3d_dbackward_" + _suffix(input))(*_desc(ggI), *_desc(ggG), *_desc(grad_output), *_desc(input),
                                                              *_desc(grid), *_desc(gi), *_desc(gg), *_desc(ggo),
                                                              C.c_int(interp), C.c_int(pad))
    return gi, gg, ggo


# ------------------------------------------------------------------------------------ 2x boundary upsampler
def interp2x_forward(input, balance_value):
    """interp2x_boundary3d_cuda_forward_kernel (interp2x_boundary3d_kernel.cu:10-151): [output, is_boundary]."""
    input = input.contiguous()
    B, Cc, d, h, w = input.shape
    out = torch.empty(B, Cc, 2 * d - 1, 2 * h - 1, 2 * w - 1, dtype=input.dtype)
    bnd = torch.zeros(out.shape, dtype=torch.bool)
    getattr(_lib("interp2x"), "ref_interp2x_forward_" + _suffix(input))(_p(input), _i32(input.shape), _p(out), _p(bnd),
                                                                      _i32(out.shape), C.c_float(balance_value))
    return [out, bnd]


def interp2x_backward(grad_output):
    """interp2x_boundary3d_cuda_backward_kernel (:154-239)."""
    grad_output = grad_output.contiguous()
    B, Cc, D, H, W = grad_output.shape
    gi = torch.empty(B, Cc, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2, dtype=grad_output.dtype)
    getattr(_lib("interp2x"), "ref_interp2x_backward_" + _suffix(grad_output))(_p(grad_output), _i32(grad_output.shape), _p(gi),
                                                                             _i32(gi.shape))
    return gi
