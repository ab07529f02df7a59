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


def mc_gpu_multi(volumes, xstep=1.0, ystep=1.0, zstep=1.0, xmin=0.0, ymin=0.0, zmin=0.0, fTargetValue=0.0):
    """`mc_gpu` for several volumes of ONE lattice size — the body net and the garment nets of a re-mesh
    (OptimGarmentNetwork.py:581-618 calls MCGpu.mc_gpu once per net) — in ONE set of four launches (recmv_mc_run_batch: grid y =
    volume) and ONE counter read-back.  Per volume the result is exactly `mc_gpu`'s.  Volumes without a size guess yet (first
    extraction of a grid) or more than four of them take the per-volume route."""
    vols = list(volumes)
    ok = (1 < len(vols) <= 4 and all(isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32 and v.dim() == 3
                                     and v.is_contiguous() and v.shape == vols[0].shape and v.device == vols[0].device for v in vols)
          and min(vols[0].shape) > 0)
    dev = vols[0].device if ok else None
    dev_id = vols[0].get_device() if ok else -1
    nx, ny, nz = (vols[0].shape if ok else (0, 0, 0))
    keys = [(dev_id, nx, ny, nz, i) for i in range(len(vols))]
    if not ok or not (0 <= dev_id < 8) or any(k not in _last_sizes for k in keys):
        out = []
        for i, v in enumerate(vols):
            r = mc_gpu(v, xstep, ystep, zstep, xmin, ymin, zmin, fTargetValue)
            if ok and r:
                _last_sizes[keys[i]] = (int(r[0].shape[0]), int(r[1].shape[0]))
            out.append(r)
        return out
    lib = L.lib()
    n = len(vols)
    with L.device_guard(dev):
        nbytes = int(lib.recmv_mc_workspace_bytes(nx, ny, nz))
        st = torch.cuda.current_stream(dev)
        wss = []
        for i in range(n):
            key = (dev_id, st.cuda_stream, 'multi', i)
            ws = _workspaces.get(key)
            if ws is None or ws.numel() < nbytes:
                ws = _workspaces[key] = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
            wss.append(ws)
        caps = [tuple(int(g * 1.25) + 4096 for g in _last_sizes[k]) for k in keys]
        vbufs = [torch.empty((c[0], 3), dtype=torch.float32, device=dev) for c in caps]
        fbufs = [torch.empty((c[1], 3), dtype=torch.int64, device=dev) for c in caps]
        counts_dev = torch.empty((n, 3), dtype=torch.int32, device=dev)
        PtrArr, I64Arr = C.c_void_p * n, C.c_int64 * n
        geom = (float(fTargetValue), float(xstep), float(ystep), float(zstep), float(xmin), float(ymin), float(zmin))
        L.check(lib.recmv_mc_run_batch(n, PtrArr(*[v.data_ptr() for v in vols]), nx, ny, nz, *geom,
                                       PtrArr(*[w.data_ptr() for w in wss]), nbytes, PtrArr(*[b.data_ptr() for b in vbufs]),
                                       I64Arr(*[c[0] for c in caps]), PtrArr(*[b.data_ptr() for b in fbufs]),
                                       I64Arr(*[c[1] for c in caps]), PtrArr(*[counts_dev[i].data_ptr() for i in range(n)]),
                                       L.stream_ptr(dev)), "mc_gpu_multi/run")
        key_p = ('multi', dev_id)
        host = _pinned.get(key_p)
        if host is None or host.numel() < 3 * n:
            host = _pinned[key_p] = torch.zeros(12, dtype=torch.int32).pin_memory()
        host[:3 * n].copy_(counts_dev.view(-1), non_blocking=True)
        st.synchronize()
        sizes = host[:3 * n].view(n, 3).tolist()
        out = []
        for i, (V, F, A) in enumerate(sizes):
            if V <= caps[i][0] and F <= caps[i][1]:
                vertices, faces = vbufs[i][:V], fbufs[i][:F]
            else:                                   # the surface outgrew the margin: emit this volume again, exact sizes
                vertices = torch.empty((V, 3), dtype=torch.float32, device=dev)
                faces = torch.empty((F, 3), dtype=torch.int64, device=dev)
                L.check(lib.recmv_mc_emit(L.ptr(vols[i]), nx, ny, nz, *geom, L.ptr(wss[i]), wss[i].numel(), A, L.ptr(vertices), V,
                                          L.ptr(faces), F, L.stream_ptr(dev)), "mc_gpu_multi/emit")
            _last_sizes[keys[i]] = (V, F)
            out.append([vertices, faces])
    return out
