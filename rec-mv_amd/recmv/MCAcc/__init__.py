"""recmv.MCAcc — same public names as the reference's MCAcc package (MCAcc/__init__.py:1-3)."""
from .seg3d_lossless import Seg3dLossless
from .utils import create_grid3D
from .grid_sampler_mine import GridSamplerMine3dFunction
