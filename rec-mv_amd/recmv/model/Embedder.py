"""Positional encoding (model/Embedder.py:4-65 of the reference).

gamma(x) = [x, w0*sin(2^0 x), w1*cos(2^0 x), ..., w_{2L-2}*sin(2^{L-1} x), w_{2L-1}*cos(2^{L-1} x)]
Two paths with identical values: a differentiable torch path (any order of autograd — the SDF normal,
eikonal and deformer-Jacobian terms differentiate through it) and the fused HIP kernel
(`recmv_posenc_forward`) used whenever no gradient is required.
"""
import torch

from .. import ops


class Embedder:
    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs["input_dims"]
        self.include_input = kwargs["include_input"]
        self.num_freqs = kwargs["num_freqs"]
        max_freq = kwargs["max_freq_log2"]
        if kwargs["log_sampling"]:
            self.freq_bands = 2.0 ** torch.linspace(0.0, max_freq, self.num_freqs)     # Embedder.py:26-27
        else:
            self.freq_bands = torch.linspace(2.0 ** 0.0, 2.0 ** max_freq, self.num_freqs)
        self.periodic_fns = kwargs["periodic_fns"]
        self.out_dim = (d if self.include_input else 0) + d * self.num_freqs * len(self.periodic_fns)
        self._hip_ok = (d == 3 and self.include_input and kwargs["log_sampling"]
                        and list(self.periodic_fns) == [torch.sin, torch.cos])

    def embed(self, inputs, ws=None):
        if self._hip_ok and inputs.is_cuda and inputs.dtype == torch.float32 and inputs.shape[-1] == 3:
            flat = inputs.reshape(-1, 3)
            wl = None if ws is None else tuple(float(w) for w in ws)
            if torch.is_grad_enabled() and inputs.requires_grad:
                out = ops.PosEnc.apply(flat, self.num_freqs, wl)
            else:
                out = ops.posenc(flat, self.num_freqs, wl)
            return out.view(*inputs.shape[:-1], self.out_dim)
        outs = [inputs] if self.include_input else []
        i = 0
        for freq in self.freq_bands.tolist():
            for fn in self.periodic_fns:
                v = fn(inputs * freq)
                if ws is not None:
                    v = ws[i] * v                                                      # Embedder.py:34-35
                outs.append(v)
                i += 1
        return torch.cat(outs, -1)


def get_embedder(multires):
    embed_kwargs = {
        "include_input": True,
        "input_dims": 3,
        "max_freq_log2": multires - 1,
        "num_freqs": multires,
        "log_sampling": True,
        "periodic_fns": [torch.sin, torch.cos],
    }
    eo = Embedder(**embed_kwargs)

    def embed(x, ws=None, eo=eo):
        return eo.embed(x, ws)

    return embed, eo.out_dim
