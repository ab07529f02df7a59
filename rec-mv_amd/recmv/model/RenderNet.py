"""IDR-style colour network (model/RenderNet.py:10-103 of the reference) on the fused MFMA kernels."""
import warnings

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..utils.utils import annealing_weights
from .Embedder import get_embedder


class RenderingNetwork_view_norm(nn.Module):
    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, multires_n=0, multires_v=0):
        super().__init__()
        self.mode = mode
        dims = [d_in + feature_vector_size] + list(dims) + [d_out]
        self.embedv_fn = None
        self.multires_v = multires_v
        if multires_v > 0:
            embedv_fn, input_ch = get_embedder(multires_v)
            self.embedv_fn = embedv_fn
            dims[0] += (input_ch - 3)
        self.embedn_fn = None
        self.multires_n = multires_n
        if multires_n > 0:
            embedn_fn, input_ch = get_embedder(multires_n)
            self.embedn_fn = embedn_fn
            dims[0] += (input_ch - 3)
        self.num_layers = len(dims)
        self.weight_norm = weight_norm
        for l in range(0, self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if weight_norm:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    lin = nn.utils.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)
        self.relu = nn.ReLU()
        self.tanh = nn.Tanh()

    def _weight(self, l):
        lin = getattr(self, "lin" + str(l))
        if self.weight_norm:
            v, g = lin.weight_v, lin.weight_g
            # computed once per parameter version and shared by every pass / stream of the optimiser step (ops.weight_norm_shared)
            return ops.weight_norm_shared(self, l, v, g), lin.bias
        return lin.weight, lin.bias

    @staticmethod
    def _embed(fn, x, multires, ratio):
        if ratio is None:
            return fn(x)
        if ratio <= 0:
            return fn(x, [0. for _ in range(multires * 2)])
        return fn(x, annealing_weights(multires, ratio))

    def forward(self, points, normals, view_dirs, feature_vectors, ratio):
        ratio = ratio['renderRatio']
        if self.embedv_fn is not None:
            view_dirs = self._embed(self.embedv_fn, view_dirs, self.multires_v, ratio)
        if self.embedn_fn is not None:
            normals = self._embed(self.embedn_fn, normals, self.multires_n, ratio)
        if self.mode == 'idr':
            x = torch.cat([points, view_dirs, normals, feature_vectors], dim=-1)
        elif self.mode == 'no_view_dir':
            x = torch.cat([points, normals, feature_vectors], dim=-1)
        elif self.mode == 'no_normal':
            x = torch.cat([points, view_dirs, feature_vectors], dim=-1)
        for l in range(0, self.num_layers - 1):
            W, b = self._weight(l)
            last = l == self.num_layers - 2
            # the final tanh (RenderNet.py:94) is fused into the last layer's epilogue
            x = ops.linear_act(x, W, b, ops.ACT_TANH if last else ops.ACT_RELU, 0.0)
        return x


def getRenderNet(device, conf):
    if conf.get_string('type') == 'RenderingNetwork_view_norm':
        return RenderingNetwork_view_norm(conf.get_int('condlen'), d_in=9, d_out=3, dims=[512, 512, 512, 512],
                                          mode='idr', weight_norm=True, multires_v=conf.get_int('multires_v'),
                                          multires_n=conf.get_int('multires_n')).to(device)
    return globals()[conf.get_string('type')](conf.get_int('condlen'), d_in=9, d_out=3, dims=[512, 512, 512, 512],
                                              mode='idr', weight_norm=True,
                                              multires=conf.get_int('multires')).to(device)
