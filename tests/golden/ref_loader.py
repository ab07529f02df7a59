"""Import the REAL reference Python modules (from /root/reference) in this CPU-only container.

Used only by tests/golden/make_golden.py to GENERATE fixtures (the GPU box has no /root/reference).
The reference cannot be imported as shipped: its modules import CUDA extensions and third-party
packages that are absent here.  This loader makes the import succeed without touching
the reference tree:

  * third-party packages that are absent (pytorch3d, openmesh, trimesh, pyhocon, wandb, cv2, ...)
    are replaced by inert dummy modules (any attribute = an empty class);
  * the reference's own CUDA extensions are backed by the CPU oracle (oracle/recmv_oracle.c):
        FastMinv.Fast3x3Minv / Fast3x3Minv_backward      -> oracle.inv3x3_*
        GridSamplerMine.forward / backward / dbackward    -> oracle.gs3d_*
        interp2x_boundary3d.forward / backward            -> oracle.interp2x_*
        MCGpu.mc_gpu                                      -> oracle.mc
  * torch_scatter.scatter -> Tensor.scatter_reduce ; smpl_pytorch.util.batch_rodrigues -> the standard
    HMR form (un-vendored in the reference: parity unpinned, see recmv/model/Deformer.py).

Everything else that runs is the reference's own Python code.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys
import types
from pathlib import Path

import torch

REF = Path("/root/reference")
REPO = Path(__file__).resolve().parent.parent.parent

DUMMY_TOPLEVEL = {"pytorch3d", "openmesh", "trimesh", "pyhocon", "wandb", "cv2", "skimage", "mcubes",
                  "vtkplotter", "matplotlib", "PIL", "tqdm", "imageio", "open3d", "chumpy", "sklearn_dummy",
                  "torchvision", "ot", "sksparse", "filterpy", "mmcv", "yacs", "tensorboardX", "kornia", "pymeshlab", "igl", "psbody", "cvxpy", "numba"}


class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _make_dummy(name)

    def __call__(cls, *a, **k):
        return object.__new__(cls)


def _make_dummy(name):
    return _DummyMeta(name, (), {"__init__": lambda self, *a, **k: None})


class _DummyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        d = _make_dummy(name)
        setattr(self, name, d)
        return d


class _DummyFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in DUMMY_TOPLEVEL:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _DummyModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _batch_rodrigues(theta):
    from recmv.model.Deformer import batch_rodrigues  # same restatement on both sides (documented)
    return batch_rodrigues(theta)


def _scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if dim < 0:
        dim += src.dim()
    red = {"sum": "sum", "add": "sum", "mean": "mean", "min": "amin", "max": "amax"}[reduce]
    if out is None:
        size = list(src.shape)
        size[dim] = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
        out = torch.zeros(size, dtype=src.dtype, device=src.device)
        return out.scatter_reduce(dim, index.expand_as(src) if index.dim() == src.dim() else index, src, red,
                                  include_self=False)
    return out.scatter_reduce(dim, index, src, red, include_self=True)


def install():
    """Make `import model...`, `import utils...`, `import MCAcc...` resolve to the reference."""
    if getattr(install, "_done", False):
        return
    for p in (str(REPO / "rec-mv_amd"), str(REPO)):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import oracle as orc
    sys.meta_path.insert(0, _DummyFinder())

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("FastMinv", Fast3x3Minv=orc.inv3x3_forward, Fast3x3Minv_backward=orc.inv3x3_backward)
    mod("GridSamplerMine",
        forward=lambda i, g, a, b: orc.gs3d_forward(i, g),
        backward=lambda i, g, go, a, b: orc.gs3d_backward(i, g, go),
        dbackward=lambda gI, gG, i, g, go, a, b: orc.gs3d_dbackward(gI.contiguous(), gG.contiguous(), i, g, go))
    mod("interp2x_boundary3d", forward=orc.interp2x_forward, backward=orc.interp2x_backward)
    mod("MCGpu", mc_gpu=orc.mc, mc_init=lambda d: None)
    mod("torch_scatter", scatter=_scatter)
    sp = mod("smpl_pytorch")
    sp.__path__ = []
    mod("smpl_pytorch.util", batch_rodrigues=_batch_rodrigues)
    mod("smpl_pytorch.SMPL", SMPL=_make_dummy("SMPL"), getSMPL=lambda *a, **k: None)
    # the reference root must come first so `model`, `utils`, `MCAcc`, `engineer` are ITS packages
    sys.path.insert(0, str(REF))
    install._done = True


def ref_module(name):
    install()
    return importlib.import_module(name)
