"""Drop-in for the reference's `MCGpu` extension (MCGpu/MCGpu.cpp:14-60).

Deterministic: vertices sorted by edge key, faces by (voxel, triangle) — see csrc/marching_cubes.hip.
"""
import ctypes as C

import torch

from . import _lib as L

_workspaces = {}   # per-device persistent scratch, grown monotonically (like MCGpu::init, CudaKernels.cu:572-604)


def mc_init(device_id):
    """Pre-create the per-device scratch holder (MCGpu.cpp:14-17). Optional."""
    _workspaces.setdefault(int(device_id), None)


def _workspace(device, nbytes):
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)       # per stream: concurrent extractions must not share
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


_last_sizes = {}    # (device, nx, ny, nz) -> (V, F) of the previous extraction: capacity guess for the next one
_pinned = {}        # device -> pinned int32[3] the counters are copied into


def _pinned_counts(dev_id):
    buf = _pinned.get(dev_id)
    if buf is None:
        buf = _pinned[dev_id] = torch.zeros(3, dtype=torch.int32).pin_memory()
    return buf


def mc_gpu(sdfs, xstep=1.0, ystep=1.0, zstep=1.0, xmin=0.0, ymin=0.0, zmin=0.0, fTargetValue=0.0):
    """sdfs [NX,NY,NZ] CUDA contiguous f32 -> [vertices [V,3] f32, faces [F,3] i64]; [] on bad sizes.

    The reference runs kernel -> D2H counter read -> kernel (CudaKernels.cu:620-634).  Here the whole extraction is
    enqueued at once into buffers sized from the previous extraction of the same grid (+25 %: between two re-meshes the
    surface moves little) and the counters are read once, after the last kernel; only if a buffer turns out too small
    is the emit pass repeated with exact sizes (the classification stays in the workspace).  The first call for a
    grid has no guess and takes the two-phase route (count -> allocate -> emit)."""
    L.require_cuda(sdfs, "sdfs")
    L.require_contiguous(sdfs, "sdfs")
    if sdfs.dtype != torch.float32:
        # reference: .data<float>() throws on a non-float tensor (MCGpu.cpp:49)
        raise RuntimeError("expected scalar type Float but found " + str(sdfs.dtype))
    dev_id = sdfs.get_device()
    if dev_id < 0 or dev_id >= 8:        # MCGpu.cpp:43-45
        return []
    if sdfs.dim() != 3 or min(sdfs.shape) <= 0:   # MCGpu::init returns false (CudaKernels.cu:574-575)
        return []
    nx, ny, nz = sdfs.shape
    lib = L.lib()
    with L.device_guard(sdfs.device):
        nbytes = int(lib.recmv_mc_workspace_bytes(nx, ny, nz))
        if nbytes <= 0:
            raise RuntimeError(f"mc_gpu: volume {tuple(sdfs.shape)} is outside the supported range "
                               "(at most 2^28 lattice points)")
        ws = _workspace(sdfs.device, max(nbytes, 256))
        st = L.stream_ptr(sdfs.device)
        geom = (float(fTargetValue), float(xstep), float(ystep), float(zstep), float(xmin), float(ymin), float(zmin))
        key = (dev_id, nx, ny, nz)
        guess = _last_sizes.get(key)
        if guess is None:
            counts = (C.c_int32 * 3)(0, 0, 0)
            L.check(lib.recmv_mc_count(L.ptr(sdfs), nx, ny, nz, geom[0], L.ptr(ws), ws.numel(),
                                       C.cast(counts, C.c_void_p), st), "mc_gpu/count")
            V, F, A = int(counts[0]), int(counts[1]), int(counts[2])
            vertices = torch.empty((V, 3), dtype=torch.float32, device=sdfs.device)
            faces = torch.empty((F, 3), dtype=torch.int64, device=sdfs.device)
            if A > 0:
                L.check(lib.recmv_mc_emit(L.ptr(sdfs), nx, ny, nz, *geom, L.ptr(ws), ws.numel(), A, L.ptr(vertices),
                                          V, L.ptr(faces), F, st), "mc_gpu/emit")
        else:
            vcap, fcap = (int(g * 1.25) + 4096 for g in guess)
            vbuf = torch.empty((vcap, 3), dtype=torch.float32, device=sdfs.device)
            fbuf = torch.empty((fcap, 3), dtype=torch.int64, device=sdfs.device)
            counts_dev = torch.empty(3, dtype=torch.int32, device=sdfs.device)
            L.check(lib.recmv_mc_run(L.ptr(sdfs), nx, ny, nz, *geom, L.ptr(ws), ws.numel(), L.ptr(vbuf), vcap,
                                     L.ptr(fbuf), fcap, L.ptr(counts_dev), st), "mc_gpu/run")
            host = _pinned_counts(dev_id)
            host.copy_(counts_dev, non_blocking=True)
            torch.cuda.current_stream(sdfs.device).synchronize()
            V, F, A = (int(x) for x in host.tolist())
            if V <= vcap and F <= fcap:
                vertices, faces = vbuf[:V], fbuf[:F]
            else:                                   # the surface grew by more than the margin: emit again, exact sizes
                vertices = torch.empty((V, 3), dtype=torch.float32, device=sdfs.device)
                faces = torch.empty((F, 3), dtype=torch.int64, device=sdfs.device)
                L.check(lib.recmv_mc_emit(L.ptr(sdfs), nx, ny, nz, *geom, L.ptr(ws), ws.numel(), A, L.ptr(vertices),
                                          V, L.ptr(faces), F, st), "mc_gpu/emit")
        _last_sizes[key] = (V, F)
    return [vertices, faces]
