"""recmv.model — the reference's `model` package surface for the hot path (model/__init__.py:1-3)."""
from .Embedder import get_embedder, Embedder
from .network import ImplicitNetwork, getTmpSdf, getOptNet
from .Deformer import (MLPTranslator, LBSkinner, CompositeDeformer, Inverse_Fl_Body, getTranslatorNet, batch_rodrigues,
                       compute_lbswField, initialLBSkinner, smooth_weights)
from .RenderNet import RenderingNetwork_view_norm, getRenderNet
from .CameraMine import RectifiedPerspectiveCameras
