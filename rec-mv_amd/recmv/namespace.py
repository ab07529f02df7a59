"""The reference's top-level import names, bound to this package.

REC-MV's scripts import by bare top-level names (`from model.network import getOptNet`, `import utils`,
`from MCAcc import Seg3dLossless`, `from engineer.core.fl_optimizer import fl_proj_loss`, `import FastMinv`, ...:
train.py:1-20, engineer/networks/OptimGarmentNetwork.py:1-40).  `install()` registers those names in `sys.modules` so
that a script written against the reference runs on this package unchanged:

    import recmv.namespace; recmv.namespace.install()
    from model.network import getOptNet
    from engineer.core.fl_optimizer import fl_proj_loss
    from engineer.core.beta_optimizer import smpl_beta_optimizer
    from MCAcc import Seg3dLossless, create_grid3D, GridSamplerMine3dFunction
    import FastMinv, MCGpu, GridSamplerMine, interp2x_boundary3d, utils
    from dataset.dataset import getDatasetAndLoader

It refuses to shadow a module that is already imported under one of these names (e.g. the reference itself).
"""
import importlib
import sys

ALIASES = {
    "FastMinv": "recmv.FastMinv", "MCGpu": "recmv.MCGpu", "GridSamplerMine": "recmv.GridSamplerMine",
    "interp2x_boundary3d": "recmv.interp2x_boundary3d",
    "MCAcc": "recmv.MCAcc", "MCAcc.seg3d_lossless": "recmv.MCAcc.seg3d_lossless",
    "model": "recmv.model", "model.network": "recmv.model.network", "model.Deformer": "recmv.model.Deformer",
    "model.Embedder": "recmv.model.Embedder", "model.RenderNet": "recmv.model.RenderNet",
    "model.CameraMine": "recmv.model.CameraMine",
    "utils": "recmv.utils", "utils.utils": "recmv.utils.utils", "utils.FindSurfacePs": "recmv.utils.FindSurfacePs",
    "utils.constant": "recmv.utils.constant",
    "dataset": "recmv.dataset", "dataset.dataset": "recmv.dataset.dataset",
    "engineer.utils": "recmv.engineer.utils", "engineer.utils.featureline_utils": "recmv.engineer.utils.featureline_utils",
    "engineer.utils.polygons": "recmv.engineer.utils.polygons",
    "engineer.utils.matrix_transform": "recmv.engineer.utils.matrix_transform",
    "engineer": "recmv.engineer", "engineer.core": "recmv.engineer.core",
    "engineer.core.fl_optimizer": "recmv.engineer.core.fl_optimizer",
    "engineer.core.beta_optimizer": "recmv.engineer.core.beta_optimizer",
    "engineer.visualizer": "recmv.engineer.visualizer",
    "engineer.visualizer.wandb_visualizer": "recmv.engineer.visualizer.wandb_visualizer",
    "engineer.networks": "recmv.engineer.networks",
    "engineer.networks.OptimGarmentNetwork": "recmv.engineer.networks.OptimGarmentNetwork",
    "engineer.networks.OptimGarmentNetwork_Large_Pose": "recmv.engineer.networks.OptimGarmentNetwork_Large_Pose",
}


def install():
    for alias, target in ALIASES.items():
        mod = importlib.import_module(target)
        have = sys.modules.get(alias)
        if have is not None and have is not mod:
            raise ImportError(f"recmv.namespace.install: a different module is already imported as '{alias}' "
                              f"({getattr(have, '__file__', have)})")
        sys.modules[alias] = mod
    return sorted(ALIASES)
