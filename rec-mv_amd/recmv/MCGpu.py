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


def mc_gpu(sdfs, xstep=1.0, ystep=1.0, zstep=1.0, xmin=0.0, ymin=0.0, zmin=0.0, fTargetValue=0.0):
    """sdfs [NX,NY,NZ] CUDA contiguous f32 -> [vertices [V,3] f32, faces [F,3] i64]; [] on bad sizes."""
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
    with torch.cuda.device(sdfs.device):
        nbytes = lib.recmv_mc_workspace_bytes(nx, ny, nz)
        ws = _workspace(sdfs.device, max(int(nbytes), 256))
        counts = (C.c_int32 * 3)(0, 0, 0)
        st = L.stream_ptr(sdfs.device)
        L.check(lib.recmv_mc_count(L.ptr(sdfs), nx, ny, nz, float(fTargetValue), L.ptr(ws), ws.numel(),
                                   C.cast(counts, C.c_void_p), st), "mc_gpu/count")
        V, F = int(counts[0]), int(counts[1])
        vertices = torch.empty((V, 3), dtype=torch.float32, device=sdfs.device)
        faces = torch.empty((F, 3), dtype=torch.int64, device=sdfs.device)
        if V > 0 or F > 0:
            L.check(lib.recmv_mc_emit(L.ptr(sdfs), nx, ny, nz, float(fTargetValue), float(xstep), float(ystep),
                                      float(zstep), float(xmin), float(ymin), float(zmin), L.ptr(ws), ws.numel(),
                                      int(counts[2]), L.ptr(vertices), L.ptr(faces), st), "mc_gpu/emit")
    return [vertices, faces]
