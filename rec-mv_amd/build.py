"""Build librecmv_hip.so (gfx950) from rec-mv_amd/csrc/*.hip with hipcc.

In-tree build: objects go to rec-mv_amd/build/, the library to rec-mv_amd/lib/librecmv_hip.so (both
git-ignored, both travel to the GPU box with the gpurun snapshot).  hipcc cross-compiles gfx950 without
a GPU, so this runs in the CPU-only container too.

    python rec-mv_amd/build.py [--force] [--verbose]
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
# RECMV_BUILD_TAG=<tag>: an experiment build beside the product's (objects in build_<tag>/, lib/librecmv_hip_<tag>.so; load it with
# RECMV_LIB_PATH) — e.g. RECMV_BUILD_TAG=libm RECMV_HIPCC_EXTRA=-DRECMV_LIBM_SOFTPLUS for tools/trajectory_seeds.py
_TAG = os.environ.get("RECMV_BUILD_TAG", "")
OBJ = HERE / ("build_" + _TAG if _TAG else "build")
LIBDIR = HERE / "lib"
LIB = LIBDIR / ("librecmv_hip_%s.so" % _TAG if _TAG else "librecmv_hip.so")
INCLUDE = HERE.parent / "include"

ARCH = "gfx950"
COMMON_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-Wall",
    "-Wno-unused-function",
    "-Wno-unused-variable",
    f"-I{INCLUDE}",
]
# No packed-f32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).  Round 5 traced the run-to-run divergence of the optional
# bf16x6 matrix mode to exactly these: a wave executing them while waves of that mode's NT product kernels run beside it on the device
# gets WRONG results in lanes 48-63, a few launches in a hundred (gfx950, ROCm 7.2; tools/erratum/valu_disturb_repro.hip: the same column updates
# fail as v_pk_* and never fail as scalar v_mul / v_fma).  The default (f32) mode runs no such product kernel and is
# bit-reproducible with packed instructions (several hundred counted repetitions), and removing them costs the f32 MFMA kernels 2-3 %
# and the sampler ~10 % — so by default only the kernel that was caught (the deformation regulariser: long packed-f32 chains, per
# thread) is built without them, and RECMV_NO_PACKED_F32=1 builds EVERY kernel without them: the build to use with
# RECMV_GEMM_MODE=1 (with it the bf16x6 loop is identical in 60 of 60 repetitions, profiles/erratum/r05_race_12_no_packed_f32.txt).
# The switch is read by the device pass only (the host pass prints "not a recognized feature ... ignoring").
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DRECMV_NO_PACKED_F32=1"]
ALWAYS_NO_PACKED = {"def_regu.hip"}

# extern "C" entry points are exported explicitly through this macro-free rule: default visibility for
# extern "C" symbols only is obtained by the version script below.
VERSION_SCRIPT = HERE / "csrc" / "exports.map"


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: librecmv_hip.so cannot be built")
    return exe


def _newest_header_mtime() -> float:
    m = 0.0
    for p in list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list(INCLUDE.glob("*.h")):
        m = max(m, p.stat().st_mtime)
    return m


COMPILED = []          # sources this process actually compiled (build() reports them: "does it build" is observable)


def _compile(src: Path, force: bool, verbose: bool, hdr_mtime: float) -> Path:
    obj = OBJ / (src.stem + ".o")
    extra = os.environ.get("RECMV_HIPCC_EXTRA", "").split()      # e.g. -DRECMV_ROWS_TIMING for tools/mlp_rows_clock.py
    if os.environ.get("RECMV_NO_PACKED_F32") == "1" or src.name in ALWAYS_NO_PACKED:
        extra = extra + NO_PACKED_F32
    if src.name in os.environ.get("RECMV_NOCONTRACT_FILES", "").split(","):       # bisecting builds of tools/trajectory_seeds.py
        extra = extra + ["-ffp-contract=off"]
    # an object is reused only if it is newer than its source and the headers AND was compiled with these very flags (a stamp beside
    # it): switching RECMV_NO_PACKED_F32 / RECMV_HIPCC_EXTRA without --force must never link objects of two flag sets into one library
    flags = " ".join([*COMMON_FLAGS, *extra])
    stamp = obj.with_suffix(".flags")
    if (not force and obj.exists() and obj.stat().st_mtime > src.stat().st_mtime
            and obj.stat().st_mtime > hdr_mtime and stamp.exists() and stamp.read_text() == flags):
        return obj
    tmp = obj.with_suffix(f".o.tmp{os.getpid()}")
    cmd = [hipcc(), *COMMON_FLAGS, *extra, "-c", str(src), "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        tmp.unlink(missing_ok=True)
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, obj)
    stamp.write_text(flags)
    COMPILED.append(src.name)
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    """One builder at a time (every rank of a torchrun job may call this at once): an exclusive file lock around the
    whole build, objects and the library written to temporary names and renamed atomically, so that a concurrent
    reader never sees a truncated file."""
    import fcntl
    OBJ.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    with open(OBJ / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> Path:
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError(f"no .hip sources under {CSRC}")
    hdr_mtime = _newest_header_mtime()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose, hdr_mtime), srcs))
    newest_obj = max(o.stat().st_mtime for o in objs)
    if force or not LIB.exists() or LIB.stat().st_mtime < newest_obj:
        tmp = LIB.with_suffix(f".so.tmp{os.getpid()}")
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(tmp), *map(str, objs),
               f"-Wl,--version-script={VERSION_SCRIPT}"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            tmp.unlink(missing_ok=True)
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    lib = build(a.force, a.verbose)
    print(lib)
    sys.exit(0)
