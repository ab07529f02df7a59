"""ctypes binding of librecmv_hip.so (the C ABI declared in include/recmv_hip.h).

There is NO fallback: if the library is absent (and cannot be built because hipcc is missing) importing
any recmv op raises.  The product path never routes through a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent.parent          # rec-mv_amd/
LIB_PATH = Path(os.environ["RECMV_LIB_PATH"]) if os.environ.get("RECMV_LIB_PATH") else _PKG / "lib" / "librecmv_hip.so"   # (override: A/B builds of tools/)

RECMV_OK = 0
ABI_VERSION = 9          # include/recmv_hip.h; bumped when a signature changes (v9: recmv_lbs_jet_* added; v8: recmv_cam_* added; v7: recmv_get_sampler_mode, recmv_set_jet_fill added; v5: second weight set + split_row in recmv_mlp; v6: recmv_def_regu, recmv_b3_*, recmv_mlp_rows_*, recmv_mc_run_batch added)
F32, F64 = 0, 1
ACT_NONE, ACT_RELU, ACT_SOFTPLUS, ACT_TANH = 0, 1, 2, 3


class Tensor5(C.Structure):
    _fields_ = [("size", C.c_int64 * 5), ("stride", C.c_int64 * 5)]


MLP_MAX_LAYERS = 12


class Mlp(C.Structure):
    """recmv_mlp of include/recmv_hip.h."""
    _fields_ = [("n_layers", C.c_int32), ("multires", C.c_int32), ("cond_dim", C.c_int32), ("skip_layer", C.c_int32),
                ("hidden_act", C.c_int32), ("residual", C.c_int32), ("act_param", C.c_float),
                ("dims", C.c_int32 * (MLP_MAX_LAYERS + 1)), ("rows", C.c_int32 * MLP_MAX_LAYERS),
                ("W", C.c_void_p * MLP_MAX_LAYERS), ("Wt", C.c_void_p * MLP_MAX_LAYERS),
                ("bias", C.c_void_p * MLP_MAX_LAYERS), ("pe_weights", C.c_float * 32),
                ("W2", C.c_void_p * MLP_MAX_LAYERS), ("Wt2", C.c_void_p * MLP_MAX_LAYERS),
                ("bias2", C.c_void_p * MLP_MAX_LAYERS), ("split_row", C.c_int64)]


class LbsGrid(C.Structure):
    """recmv_lbs_grid of include/recmv_hip.h."""
    _fields_ = [("volume", C.c_void_p), ("D", C.c_int64), ("H", C.c_int64), ("W", C.c_int64),
                ("center", C.c_float * 3), ("scale", C.c_float * 3)]


_lib = None


def _load_build_module():
    spec = importlib.util.spec_from_file_location("recmv_build", _PKG / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


last_build_compiled = None     # names of the sources the last build() call of this process compiled (None: build() not called)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile librecmv_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    global last_build_compiled
    mod = _load_build_module()
    path = mod.build(force=force, verbose=verbose)
    last_build_compiled = sorted(mod.COMPILED)
    return path


def _declare(lib):
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
    T5 = C.POINTER(Tensor5)
    sigs = {
        "recmv_abi_version": (C.c_int, []),
        "recmv_no_packed_f32": (C.c_int, []),
        "recmv_last_error": (C.c_char_p, []),
        "recmv_inv3x3_forward": (C.c_int, [vp, vp, vp, i64, i32, vp]),
        "recmv_inv3x3_backward": (C.c_int, [vp, vp, vp, i64, i32, vp]),
        "recmv_def_regu": (C.c_int, [vp, i64, C.c_float, vp, vp, vp]),
        "recmv_cam_partial_floats": (i64, [i64]),
        "recmv_cam_project": (C.c_int, [vp, i64, vp, C.c_float, C.c_float, i32, vp, vp]),
        "recmv_cam_project_backward": (C.c_int, [vp, vp, i64, vp, C.c_float, C.c_float, i32, vp, vp, vp, i64, vp]),
        "recmv_cam_rays": (C.c_int, [vp, vp, vp, i64, vp, vp, vp]),
        "recmv_cam_rays_backward": (C.c_int, [vp, vp, vp, vp, i64, vp, vp, vp, i64, vp]),
        "recmv_grid_sample3d_forward": (C.c_int, [vp, T5, vp, T5, vp, T5, i32, i32, i32, vp]),
        "recmv_grid_sample3d_backward": (C.c_int, [vp, T5, vp, T5, vp, T5, vp, T5, vp, i32, i32, i32, vp]),
        "recmv_grid_sample3d_dbackward": (C.c_int, [vp, T5, vp, T5, vp, T5, vp, T5, vp, T5, vp, T5, vp, vp, T5,
                                                    i32, i32, i32, vp]),
        "recmv_interp2x_boundary3d_forward": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, f32, i32, vp]),
        "recmv_interp2x_boundary3d_backward": (C.c_int, [vp, vp, i64, i64, i64, i64, i32, vp]),
        "recmv_mc_workspace_bytes": (i64, [i64, i64, i64]),
        "recmv_mc_count": (C.c_int, [vp, i64, i64, i64, f32, vp, i64, vp, vp]),
        "recmv_mc_emit": (C.c_int, [vp, i64, i64, i64, f32, f32, f32, f32, f32, f32, f32, vp, i64, i64, vp, i64, vp, i64,
                                    vp]),
        "recmv_seg3d_select": (C.c_int, [vp, vp, i64, i64, i64, vp, vp, i64, vp, vp]),
        "recmv_seg3d_points": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, vp, vp]),
        "recmv_seg3d_apply": (C.c_int, [vp, vp, i64, f32, vp, vp, vp, vp]),
        "recmv_seg3d_expand": (C.c_int, [vp, vp, i64, i64, i64, i64, vp, vp, i64, vp, vp]),
        "recmv_mc_run": (C.c_int, [vp, i64, i64, i64, f32, f32, f32, f32, f32, f32, f32, vp, i64, vp, i64, vp, i64, vp,
                                   vp]),
        "recmv_gemm_nt": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, i32, f32, f32, vp]),
        "recmv_gemm_nt_actgrad": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, f32, f32, f32, vp]),
        "recmv_gemm_nt_mulgrad": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, i64, vp, i64, i32, f32, f32, f32, vp]),
        "recmv_gemm_nt_seg": (C.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, i64, i64, i64, i32, f32, f32, vp]),
        "recmv_gemm_nt_mulgrad_seg": (C.c_int, [vp, i64, vp, vp, i64, i64, vp, i64, i64, i64, i64, vp, i64, i32, f32, f32, f32, vp]),
        "recmv_set_gemm_mode": (C.c_int, [i32]),
        "recmv_get_gemm_mode": (C.c_int, []),
        "recmv_set_b3_families": (C.c_int, [i32]),
        "recmv_b3_planes_bytes": (i64, [i64, i64]),
        "recmv_b3_split": (C.c_int, [vp, i64, i64, i64, vp, i64, vp]),
        "recmv_b3_forget": (C.c_int, [vp]),
        "recmv_set_sampler_mode": (C.c_int, [i32]),
        "recmv_get_sampler_mode": (C.c_int, []),
        "recmv_set_jet_fill": (C.c_int, [i32]),
        "recmv_gemm_tn_workspace_bytes": (i64, [i64, i64, i64]),
        "recmv_gemm_tn": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, i64, vp, i64, vp]),
        "recmv_posenc_forward": (C.c_int, [vp, i64, vp, i64, i64, i64, i32, vp, f32, vp]),
        "recmv_act_grad": (C.c_int, [vp, vp, vp, i64, i32, f32, vp]),
        "recmv_act_grad2": (C.c_int, [vp, vp, vp, vp, i64, i32, f32, vp]),
        "recmv_weight_norm_forward": (C.c_int, [vp, vp, vp, vp, i64, i64, vp]),
        "recmv_weight_norm_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, vp]),
        "recmv_posenc_vjp": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, i32, vp, vp]),
        "recmv_posenc_jvp": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i32, vp, vp]),
        "recmv_kinematic_chain_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, vp]),
        "recmv_kinematic_chain_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, vp]),
        "recmv_mlp_workspace_bytes": (i64, [C.POINTER(Mlp), i64, i32]),
        "recmv_mlp_forward": (C.c_int, [C.POINTER(Mlp), vp, vp, i64, vp, i64, i32, vp, i64, vp, i64, i32, vp]),
        "recmv_mlp_vjp_input": (C.c_int, [C.POINTER(Mlp), vp, i64, i32, vp, i64, vp, vp, i64, vp]),
        "recmv_profile_busy": (C.c_int, [vp]),
        "recmv_profile_bytes": (C.c_int, [vp, i32]),
        "recmv_profile_large": (C.c_int, [vp, i32]),
        "recmv_mc_run_batch": (C.c_int, [i32, vp, i64, i64, i64, f32, f32, f32, f32, f32, f32, f32, vp, i64, vp, vp, vp, vp, vp, vp]),
        "recmv_mlp_rows_supported": (C.c_int, [C.POINTER(Mlp)]),
        "recmv_set_mlp_rows_tile": (C.c_int, [i32]),
        "recmv_mlp_pack_bytes": (i64, [C.POINTER(Mlp)]),
        "recmv_mlp_pack": (C.c_int, [C.POINTER(Mlp), vp, i64, vp]),
        "recmv_mlp_rows_workspace_bytes": (i64, [C.POINTER(Mlp), i64]),
        "recmv_mlp_rows_forward": (C.c_int, [C.POINTER(Mlp), vp, vp, vp, i64, vp, i64, i32, vp, i64, vp, i64, i32, vp]),
        "recmv_mlp_rows_vjp_input": (C.c_int, [C.POINTER(Mlp), vp, vp, i64, i32, vp, i64, vp, vp, i64, vp]),
        "recmv_act_grad_2d": (C.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, i32, f32, f32, f32, vp]),
        "recmv_add_scaled_2d": (C.c_int, [vp, i64, vp, i64, f32, vp, i64, i64, i64, vp]),
        "recmv_colsum_workspace_bytes": (i64, [i64, i64]),
        "recmv_colsum": (C.c_int, [vp, i64, i64, i64, vp, vp, i64, vp]),
        "recmv_linear_backward_workspace_bytes": (i64, [i64, i64, i64]),
        "recmv_linear_backward": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, f32, vp, i64, vp,
                                            vp, vp, i64, vp]),
        "recmv_mlp_jet_workspace_bytes": (i64, [C.POINTER(Mlp), i64]),
        "recmv_mlp_jet_forward": (C.c_int, [C.POINTER(Mlp), vp, vp, i64, vp, vp, i64, i32, vp, i64, vp, vp, i64, vp]),
        "recmv_mlp_jet_backward": (C.c_int, [C.POINTER(Mlp), vp, vp, i64, i32, vp, i64, vp, vp, vp, vp, vp, vp, i64,
                                             vp]),
        "recmv_gather_rows": (C.c_int, [vp, i64, vp, vp, i64, i64, i64, i64, vp]),
        "recmv_profile_begin": (C.c_int, [C.c_double]),
        "recmv_profile_end": (C.c_int, [vp, i32]),
        "recmv_lbs_forward": (C.c_int, [vp, vp, i64, vp, vp, i64, C.POINTER(LbsGrid), vp, vp, vp, vp, vp, vp, vp]),
        "recmv_lbs_vjp_input": (C.c_int, [vp, vp, i64, vp, i64, C.POINTER(LbsGrid), vp, vp, vp]),
        "recmv_lbs_vjp_params_stage": (C.c_int, [vp, vp, i64, i64, C.POINTER(LbsGrid), vp, vp, vp, vp, vp]),
        "recmv_lbs_jet_forward": (C.c_int, [vp, vp, i64, vp, vp, i64, C.POINTER(LbsGrid), vp, vp, vp]),
        "recmv_lbs_jet_backward_stage": (C.c_int, [vp, vp, i64, vp, i64, C.POINTER(LbsGrid), vp, vp, vp, vp, vp, vp, vp]),
        "recmv_rootfind_update": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, vp]),
        "recmv_rootfind_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, vp]),
        "recmv_rasterize_meshes_workspace_bytes": (i64, [i64, i64, i64, i64]),
        "recmv_rasterize_points_workspace_bytes": (i64, [i64, i64, i64, i64, f32]),
        "recmv_rasterize_points": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, i64, f32, i32, vp, vp, vp, vp, i64, vp]),
        "recmv_rasterize_points_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, f32, i32, vp, vp]),
        "recmv_alpha_composite_forward": (C.c_int, [vp, vp, vp, i64, i64, i64, i32, i64, i64, f32, vp, vp]),
        "recmv_alpha_composite_backward": (C.c_int, [vp, vp, vp, vp, i64, i64, i64, i32, i64, i64, f32, vp, vp, vp]),
        "recmv_rasterize_meshes": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, i64, f32, i32, i32, vp, vp, vp, vp, vp,
                                             i64, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return sigs


def lib():
    """The loaded library.  Builds it on first use if the .so is missing and hipcc exists."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        if not LIB_PATH.exists():
            raise ImportError(f"librecmv_hip.so not found at {LIB_PATH} and could not be built; "
                              "the recmv ops have no CPU fallback")
        l = C.CDLL(str(LIB_PATH))
        _declare(l)
        if l.recmv_abi_version() != ABI_VERSION:
            raise ImportError(f"librecmv_hip.so has ABI version {l.recmv_abi_version()}, this package needs "
                              f"{ABI_VERSION}: rebuild with `python rec-mv_amd/build.py --force`")
        if os.environ.get("RECMV_GEMM_MODE"):
            _set_gemm_mode(l, int(os.environ["RECMV_GEMM_MODE"]))
        if os.environ.get("RECMV_SAMPLER_EXACT"):
            l.recmv_set_sampler_mode(int(os.environ["RECMV_SAMPLER_EXACT"]))
        _lib = l
    return _lib


def _set_gemm_mode(l, mode):
    prev = l.recmv_set_gemm_mode(int(mode))
    if int(mode) == 1 and not l.recmv_no_packed_f32():
        import warnings
        warnings.warn("matrix mode 1 (bf16x6, EXPERIMENTAL, not part of the product's default path or of the bench line) with a library "
                      "that contains packed-f32 instructions: beside this mode's product kernels such instructions were caught computing "
                      "wrong values in lanes 48-63 (tools/erratum/README.md).  Rebuild with RECMV_NO_PACKED_F32=1 python "
                      "rec-mv_amd/build.py — and note that torch's own element-wise kernels keep such instructions either way.")
    return prev


def set_gemm_mode(mode):
    """Select the matrix mode of librecmv_hip.so (0 = f32-input MFMA, the product's arithmetic; 1 = the experimental bf16x6 split) —
    the ONE way python code switches it (tests, tools): mode 1 on a library with packed-f32 instructions warns, however it was asked
    for.  Returns the previous mode."""
    return _set_gemm_mode(lib(), mode)


def exported_symbols():
    """Names declared in include/recmv_hip.h (parsed), for the export test."""
    import re
    hdr = (_PKG.parent / "include" / "recmv_hip.h").read_text()
    return sorted(set(re.findall(r"\b(recmv_[a-z0-9_]+)\s*\(", hdr)))


def check(rc: int, what: str = ""):
    if rc != RECMV_OK:
        msg = lib().recmv_last_error().decode()
        raise RuntimeError(f"{what or 'recmv'}: {msg} (code {rc})")


def raw_stream(device) -> int:
    """The current HIP stream of `device` as a raw handle (no Stream object: this runs once per launch)."""
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice() if idx is None else idx)


def stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(raw_stream(device))


class device_guard:
    """`with torch.cuda.device(dev)` for the launch wrappers, free when `dev` already is the current device (the usual
    case: one process per GPU) — the torch context manager costs two device-index resolutions per launch."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = device.index
        self.prev = -1

    def __enter__(self):
        idx = self.idx
        if idx is not None:
            cur = torch._C._cuda_getDevice()
            if cur != idx:
                self.prev = cur
                torch._C._cuda_setDevice(idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch._C._cuda_setDevice(self.prev)
            self.prev = -1
        return False


def ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---- values built once and read from several streams -----------------------------------------------------------------------------
# The loop runs on four streams (main, ray pipeline, curve branch, second garment) and keeps lazily built per-weight-version state
# on the modules (normalised weights, their transposes, posed skeletons, packed weights).  A value built on one stream is only
# valid on another after that stream has waited for the producer: `publish()` records the event behind the producer's launches and
# `acquire()` makes the CURRENT stream wait for it when it is a different one (a hit on the producing stream costs one integer
# compare).  RECMV_CACHE_EVENTS=0 drops the waits (A/B for tools/erratum/loop_repro_inproc.py — the round-4 state of half of the caches).
def publish(device):
    """Token for a value whose producing launches have just been enqueued on the current stream of `device` (None on the host)."""
    if device is None:
        return None
    device = torch.device(device)
    if device.type != "cuda":
        return None
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    return (ev, raw_stream(device), device)


def acquire(token):
    """Make the current stream wait for the producer of a published value (no-op on the producing stream / on the host)."""
    if token is None:
        return
    ev, sid, device = token
    if raw_stream(device) != sid and os.environ.get("RECMV_CACHE_EVENTS", "1") != "0":
        torch.cuda.current_stream(device).wait_event(ev)


# ---- side streams, optionally confined to a subset of the CUs -------------------------------------------------------------------
# RECMV_SIDE_CUS=k (0 < k < 8): the loop's side streams (ray pipeline, curve branch, second garment) may only use k of every 8 CUs
# (hipExtStreamCreateWithCUMask); the main stream's large products keep the whole device.  Same results (a stream's CU set changes
# where waves run, not what they compute); an A/B knob of tools/, unset by default (measured in DESIGN.md §4).
_hip = None
_masked = []


def make_stream(device):
    """A side stream of the loop (ray pipeline, curve branch, second garment's render terms): chains of short dependent launches
    that run beside the main stream's large products.  RECMV_SIDE_PRIORITY=1 creates them with high priority; measured on the frozen
    bench scene (round 6, two runs each, profiles/r06_stream_priority_ab.txt) it changes nothing — 9.84 / 9.88 it/s with, 9.89 / 9.91
    without, the root finder's 45 ms beside the mask loss unchanged: a queue's priority does not reach the workgroup slots the large
    products already hold — so the default stays normal priority."""
    k = int(os.environ.get("RECMV_SIDE_CUS", "0") or 0)
    if not (0 < k < 8):
        if os.environ.get("RECMV_SIDE_PRIORITY", "0") == "1":
            return torch.cuda.Stream(device=device, priority=-1)
        return torch.cuda.Stream(device=device)
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    n_cu = torch.cuda.get_device_properties(idx).multi_processor_count
    words = (n_cu + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(n_cu):
        if i % 8 < k:
            mask[i // 32] |= 1 << (i % 32)
    h = C.c_void_p()
    with torch.cuda.device(idx):
        rc = _hip.hipExtStreamCreateWithCUMask(C.byref(h), C.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    _masked.append(h)
    return torch.cuda.ExternalStream(h.value, device=dev)


# ---- RECMV_POISON=1: a detector for reads of memory nobody has written yet -----------------------------------------------------
# Every workspace / output buffer the wrappers allocate with torch.empty goes through `scratch()`.  With RECMV_POISON=1 it is filled
# with a signalling pattern (NaN for floats, 0x7f bytes for raw workspaces) on the allocating stream before the kernels that are
# supposed to write it are enqueued: a kernel that reads a column nobody wrote, a fill that arrives after its reader, or a block
# handed to another stream without a wait then shows as NaN in the loop's results EVERY run instead of as last-bit differences in
# one run of four (whether a fresh block holds zeros or an earlier tensor's values depends on the allocator's history).
def poison_on() -> bool:
    return os.environ.get("RECMV_POISON", "0") == "1"


def scratch(shape, dtype, device):
    t = torch.empty(shape, dtype=dtype, device=device)
    if poison_on() and t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(0xff)                                  # 0xffffffff words are NaN when read as floats
        else:
            t.fill_(-1)
    return t


def scratch_like(x):
    return scratch(tuple(x.shape), x.dtype, x.device) if poison_on() else torch.empty_like(x)


def desc5(t) -> Tensor5:
    d = Tensor5()
    for i in range(5):
        d.size[i] = t.shape[i]
        d.stride[i] = t.stride(i)
    return d


def dtype_code(t) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise RuntimeError(f"recmv: dtype {t.dtype} not supported (float32/float64 only)")


def require_cuda(t, name):
    # the reference asserts CUDA tensors (CHECK_CUDA) -> RuntimeError
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"{name} must be a CUDA tensor")


def require_contiguous(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
