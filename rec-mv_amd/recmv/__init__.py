"""recmv — MI355X-native implementation of the REC-MV per-frame optimisation hot path.

Sub-modules keep the reference's names so call sites read the same:
  recmv.FastMinv, recmv.MCGpu, recmv.GridSamplerMine, recmv.interp2x_boundary3d   (extension modules)
  recmv.MCAcc   (Seg3dLossless, create_grid3D, GridSamplerMine3dFunction)
  recmv.model   (Embedder, ImplicitNetwork, MLPTranslator, LBSkinner, CompositeDeformer, RenderNet, cameras)
  recmv.utils   (FastDiff3x3MinvFunction, compute_Jacobian, cardinal rays, root finder, ...)
Everything computes through librecmv_hip.so (hand-written gfx950 kernels behind a C ABI); there is no CPU
fallback in this package.
"""
__version__ = "0.1.0"
