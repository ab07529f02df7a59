"""SDF network (model/network.py:27-141 of the reference): ImplicitNetwork + getTmpSdf.

Same constructor, parameter names (`lin{l}.weight_g / weight_v / bias`, weight_norm dim 0), geometric
initialisation, skip connection and `.rendcond` side effect as the reference; the arithmetic runs on
the fused MFMA kernels of librecmv_hip.so (one kernel per layer: GEMM + bias + softplus(beta=100)
[+ 1/sqrt(2) skip scale]) instead of nn.Linear/cuBLAS + separate activation kernels.
"""
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from ..utils.utils import annealing_weights
from .Embedder import get_embedder

_SQRT2 = float(np.sqrt(2))


def _wn(lin):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return nn.utils.weight_norm(lin)


class ImplicitNetwork(nn.Module):
    def __init__(self, feature_vector_size, d_in, d_out, dims, geometric_init=True, bias=1.0, skip_in=(),
                 weight_norm=True, multires=0):
        super().__init__()
        dims = [d_in] + list(dims) + [d_out + feature_vector_size]
        self.d_out = d_out
        self.embed_fn = None
        self.multires = multires
        if multires > 0:
            embed_fn, input_ch = get_embedder(multires)
            self.embed_fn = embed_fn
            dims[0] = input_ch
        self.dims = dims
        self.num_layers = len(dims)
        self.skip_in = skip_in
        self.weight_norm = weight_norm
        for l in range(0, self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if l + 1 in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:                                                      # network.py:66-80
                if l == self.num_layers - 2:
                    torch.nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    torch.nn.init.constant_(lin.bias, -bias)
                elif multires > 0 and l == 0:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                    torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif multires > 0 and l in self.skip_in:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            if weight_norm:
                lin = _wn(lin)
            setattr(self, "lin" + str(l), lin)
        self.softplus = nn.Softplus(beta=100)
        self.rendcond = None

    # -- weights ---------------------------------------------------------------------------------
    def _weight(self, l):
        lin = getattr(self, "lin" + str(l))
        if self.weight_norm:
            v, g = lin.weight_v, lin.weight_g
            # computed once per parameter version and shared by every pass / stream of the optimiser step (ops.weight_norm_shared)
            return ops.weight_norm_shared(self, l, v, g), lin.bias
        return lin.weight, lin.bias

    def _pe_weights(self, ratio):
        ratio = ratio if type(ratio) == float or type(ratio) == int or ratio is None else ratio["sdfRatio"]
        if ratio is None:
            return None
        if ratio <= 0:
            return [0.0 for _ in range(self.multires * 2)]
        return annealing_weights(self.multires, ratio)

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, input, ratio=None, jet=False, features=True):
        """Extensions over the reference's signature (both optional):
        `jet=True`: also carry d f / d x forward through the layers (csrc/mlp_jet.hip); a following
        `gradient(input, output)` then returns it without autograd's double backward, and the whole term is
        differentiable to first order in the parameters and the input.
        `features=False`: evaluate only the first `d_out` rows of the last layer (the SDF value) — for the terms that
        never read `rendcond` (eikonal, |SDF(vertices)|); the unused rows get a zero gradient either way."""
        ws = self._pe_weights(ratio) if self.embed_fn is not None else None
        needs_grad = torch.is_grad_enabled() and (input.requires_grad or
                                                  any(p.requires_grad for p in self.parameters()))
        fast = input.is_cuda and input.dtype == torch.float32 and input.dim() == 2 and self.embed_fn is not None
        self.__dict__['_jet'] = None
        J = None
        if not needs_grad and fast:
            x = self._forward_inference(input, ws)
        elif jet and fast and len(self.skip_in) <= 1:
            x, J = self._forward_jet(input, ws, None if features else self.d_out)
        else:
            x = self._forward_autograd(input, ws, None if features else self.d_out)
        if x.shape[-1] > self.d_out:
            self.rendcond = x[:, self.d_out:]
            x = x[:, 0:self.d_out]
        else:
            self.rendcond = None
        if J is not None:
            self.__dict__['_jet'] = (input, x, J)
        return x

    def _forward_jet(self, input, ws, n_last=None):
        from ..chains import mlp_jet
        nl = self.num_layers - 1
        Ws, bs = [], []
        for l in range(nl):
            W, b = self._weight(l)
            if n_last is not None and l == nl - 1:
                W, b = W[:n_last].contiguous(), b[:n_last]
            Ws.append(W)
            bs.append(b)
        dims = list(self.dims)
        if n_last is not None:
            dims[-1] = n_last
        return mlp_jet(input, None, None, Ws, bs, dims, self.multires, ws, 0,
                       self.skip_in[0] if len(self.skip_in) else -1, ops.ACT_SOFTPLUS, 100.0, False, self.d_out)

    def _forward_autograd(self, input, ws, n_last=None):
        if self.embed_fn is not None:
            input = self.embed_fn(input, ws)
        x = input
        for l in range(0, self.num_layers - 1):
            if l in self.skip_in:
                x = torch.cat([x, input], 1) / np.sqrt(2)                           # network.py:105-106
            W, b = self._weight(l)
            last = l == self.num_layers - 2
            if last and n_last is not None:
                W, b = W[:n_last], b[:n_last]
            pad = (-x.shape[1]) % 4
            if pad and x.is_cuda:
                # K = 39 -> 40 with a zero column on both operands: 16-byte aligned rows for the MFMA kernel's
                # vector loader (same product; the padded column's gradient is sliced away by autograd)
                x, W = F.pad(x, (0, pad)), F.pad(W, (0, pad))
            x = ops.linear_act(x, W, b, ops.ACT_NONE if last else ops.ACT_SOFTPLUS, 100.0)
        return x

    # -- graph-free passes: one C call per pass (csrc/mlp_chain.hip) ----------------------------------
    @torch.no_grad()
    def chain(self, ws, need_t=False):
        """MlpChain over the (cached) weight-normed weights for the annealing weights `ws`."""
        from ..chains import MlpChain
        nl = self.num_layers - 1
        Ws, bs = [], []
        for l in range(nl):
            W, b = self._weight(l)
            Ws.append(W)
            bs.append(b.detach())
        hit = self.__dict__.get('_chain_cache')
        wl = None if ws is None else tuple(float(w) for w in ws)
        if hit is not None and hit[0] == wl and len(hit[1]) == nl and all(a is b for a, b in zip(hit[1], Ws)) \
                and (hit[3] or not need_t):
            if hit[3]:
                for l in range(nl):                 # (a hit on another stream than the transposes' producer waits for them)
                    self._weight_t(l, Ws[l])
            return hit[2]
        assert len(self.skip_in) <= 1, "one skip connection (the reference uses skip_in=[4])"
        Wts = [self._weight_t(l, Ws[l]) for l in range(nl)] if need_t else None
        ch = MlpChain(Ws, bs, Wts, list(self.dims), [W.shape[0] for W in Ws], self.multires, cond_dim=0,
                      skip_layer=(self.skip_in[0] if len(self.skip_in) else -1), hidden_act=ops.ACT_SOFTPLUS,
                      act_param=100.0, residual=False, pe_weights=wl)
        self.__dict__['_chain_cache'] = (wl, Ws, ch, need_t)
        return ch

    @torch.no_grad()
    def pair_chain(self, other, ws):
        """MlpChain that evaluates rows [0, split) with THIS net and rows [split, P) with `other` (same architecture, same
        annealing weights) in one launch per layer — the two garments' SDF nets in the surface root finder.  Cached on both nets'
        normalised weights."""
        from ..chains import MlpChain
        a, b = self.chain(ws, need_t=True), other.chain(ws, need_t=True)
        assert list(self.dims) == list(other.dims) and self.skip_in == other.skip_in and self.multires == other.multires
        hit = self.__dict__.get('_pair_cache')
        if hit is not None and hit[0] is a and hit[1] is b:
            return hit[2]
        nl = self.num_layers - 1
        wl = None if ws is None else tuple(float(w) for w in ws)
        ch = MlpChain(a._keep[0], a._keep[1], a._keep[2], list(self.dims), [W.shape[0] for W in a._keep[0]], self.multires,
                      cond_dim=0, skip_layer=(self.skip_in[0] if len(self.skip_in) else -1), hidden_act=ops.ACT_SOFTPLUS,
                      act_param=100.0, residual=False, pe_weights=wl, second=(b._keep[0], b._keep[1], b._keep[2]))
        assert len(a._keep[0]) == nl
        self.__dict__['_pair_cache'] = (a, b, ch)
        return ch

    @torch.no_grad()
    def _forward_inference(self, input, ws, chunk=1 << 19):
        """No-grad path: posenc + one fused kernel per layer, the whole chain enqueued by one C call per chunk."""
        ch = self.chain(ws)
        x = input.contiguous()
        P = x.shape[0]
        if P <= chunk:
            return ch.forward(x)
        out = torch.empty((P, self.dims[-1]), dtype=torch.float32, device=x.device)
        for s in range(0, P, chunk):
            ch.forward(x[s:s + chunk], out=out[s:s + chunk])
        return out

    @torch.no_grad()
    def value_and_grad(self, x, ratio=None):
        """(f(x) [P,1], grad_x f [P,3]) without building an autograd graph: the explicit layer chain forward
        (one fused kernel per layer) and the chain backward to the INPUT only (no parameter gradients), two C calls.
        Used by the surface root finder, which needs exactly this pair at every step
        (utils/FindSurfacePs.py:316-333).  Values match forward()/gradient() to f32 rounding."""
        ch = self.chain(self._pe_weights(ratio), need_t=True)
        x = x.detach().contiguous()
        f = ch.forward(x, n_out=self.d_out, keep=True)
        grad = ch.vjp_input(x, None if self.d_out == 1 else torch.ones_like(f))
        return f, grad

    def _weight_t(self, l, W):
        """W^T (contiguous) for the input-gradient products; cached with the normalised weights."""
        hit = self.__dict__.get('_wn_cache', {}).get(l)
        if hit is None or hit[1] is not W:
            return W.t().contiguous()
        if hit[2] is None:
            L.acquire(hit[3])
            hit[2] = W.t().contiguous()
            hit[4] = L.publish(W.device)
        else:
            L.acquire(hit[4])
        return hit[2]

    def gradient(self, x, y=None):
        jet = self.__dict__.get('_jet')
        if jet is not None and y is not None and jet[0] is x and jet[1] is y and self.d_out == 1:
            return jet[2].reshape(-1, 3)           # d f / d x carried by the forward pass (forward(..., jet=True))
        x.requires_grad_(True)
        if y is None:
            y = self.forward(x)
        d_output = torch.ones_like(y, requires_grad=False, device=y.device)
        gradients = torch.autograd.grad(outputs=y, inputs=x, grad_outputs=d_output, create_graph=True,
                                        retain_graph=True, only_inputs=True)[0]
        return gradients.view(-1, 3)


def getTmpSdf(device, multires, bias=0.6, feature_vector_size=256):
    """network.py:135-141 — SDF net whose geometric init is a sphere of radius `bias`."""
    net = ImplicitNetwork(feature_vector_size=feature_vector_size, d_in=3, d_out=1,
                          dims=[512, 512, 512, 512, 512, 512, 512, 512], geometric_init=True, bias=bias,
                          skip_in=[4], weight_norm=True, multires=multires)
    return net.to(device)


def getOptNet(dataset, save_folder, N, bmins, bmaxs, resolutions, device, conf, use_initial_sdf=True,
              use_initial_skinner=True, visualizer=None, opt_large=False, **hotloop_kwargs):
    """model/network.py:182-361 of the reference: build the optimisation object train.py drives — body SDF + one SDF per
    garment (:188-199), offset MLP + skinner (:223-283), cameras, the Seg3dLossless engine (:293-305), the silhouette /
    point renderers, the colour net (:323) — and return `(optNet, sdf_initialized)`.

    `dataset` is the caller's dataset object (per-frame learnable tensors + camera: `get_grad_parameters`,
    `get_camera_parameters`, `learnable_weights`, `get_batchframe_data`, `poses`, `trans`, `conds`, `camera_params`,
    `frame_num`); `None` builds the synthetic frames.  `bmins` / `bmaxs` the canonical box, `resolutions` the pyramid of
    the current stage (rows (W,H,D)), `N` the batch size (kept for signature compatibility: the stage config sets it).
    `sdf_initialized` (:201-221): -1 when `<dataset.root>/<save_folder>/initial_sdf_idr_<multires>_<pose type>.pth` (and its
    `initial_sdf_<garment>_idr_...` companions) exist — they are loaded into the body / garment nets, with the body mesh
    `...ply` beside them when there is one; otherwise the number of pre-fit epochs the caller still has to run
    (`optNet.initializeTmpSDF`, `train.initial_iters`, 1200 when that is <= 0).  Synthetic frames (no dataset) start from the
    geometric initialisation: -1.
    `<dataset.root>/<save_folder>/initial_skinner_<pose type>.pth` (:223-236), when present and `use_initial_skinner`, supplies
    the skinning volume, its box, the rest skeleton, the SMPL template and the fitted shape.  Without that file a SMPL model
    (`smpl=<model>` or an importable smpl_pytorch) lets the first run build it as the reference does (:250-276:
    `smpl_beta_optimizer`, `initialLBSkinner` at `skin_resolution`, default 129 x 225 x 65) and write it; without either the
    rig is synthetic.
    `opt_large=True` returns the large-pose variant (OptimGarmentNetwork_LargePose, :337-340)."""
    import os
    import os.path as osp
    from ..engineer.networks import OptimGarmentNetwork, OptimGarmentNetwork_LargePose
    cls = OptimGarmentNetwork_LargePose if opt_large else OptimGarmentNetwork
    bbox = None if bmins is None else (tuple(float(v) for v in bmins), tuple(float(v) for v in bmaxs))
    kw = dict(resolutions=[tuple(int(v) for v in r) for r in resolutions] if resolutions is not None else None, bbox=bbox,
              dataset=dataset)
    if dataset is not None:
        kw.update(n_frames=int(getattr(dataset, 'frame_num', len(dataset))), H=int(dataset.H), W=int(dataset.W))
    kw.update(hotloop_kwargs)
    root = getattr(dataset, 'root', None)
    pose_type = conf.get_int('train.skinner_pose_type') if 'train.skinner_pose_type' in conf else 0
    if root is not None and save_folder is not None and use_initial_skinner and 'skinner_state' not in kw:
        # :223-236 — the skinner the reference baked on its first run (`initial_skinner_<pose type>.pth`): its volume is
        # replaced by the FITE-diffused weights beside the capture when they are there, the fitted SMPL shape goes back to
        # the dataset.  (Building that file — SMPL shape fit, template posing, weight diffusion — needs the SMPL model.)
        skinner_file = osp.join(root, save_folder, 'initial_skinner_%d.pth' % pose_type)
        if osp.isfile(skinner_file):
            data = torch.load(skinner_file, map_location='cpu', weights_only=False)     # (holds numpy arrays: parents)
            dataset.shape = data['betas']
            fite = osp.join(root, 'diffused_skinning_weights.npy')
            if osp.isfile(fite):
                data = dict(data, ws=torch.from_numpy(np.load(fite)).float()[None])
            kw['skinner_state'] = data
        else:
            # :250-276 — first run on a capture: fit the SMPL shape and one shared translation to the 2-D joints, bake the
            # skinner in the A-pose of `train.skinner_pose_type`, keep everything in initial_skinner_<pose type>.pth.
            # Needs the SMPL model (`smpl=` or smpl_pytorch); without it the rig stays synthetic.
            smpl = kw.pop('smpl', None)
            if smpl is None:
                try:
                    from .Deformer import getSMPL
                    smpl = getSMPL(dataset.gender)
                except ImportError:
                    smpl = None
            if smpl is not None:
                from ..engineer.core.beta_optimizer import smpl_beta_optimizer
                from ..utils import smpl_tmp_Apose
                from .Deformer import initialLBSkinner
                init_pose = torch.from_numpy(smpl_tmp_Apose(pose_type)).view(1, 24, 3).to(device)
                if getattr(dataset, 'gt_joints2d', None) is not None:
                    betas, extra_trans = smpl_beta_optimizer(dataset.gender, init_pose, dataset, device, smpl=smpl)
                    extra_trans = extra_trans.detach().cpu()
                else:
                    betas, extra_trans = dataset.shape.detach().clone(), None
                dataset.shape = betas.detach().cpu()
                skinner, body_vs, body_fs = initialLBSkinner(dataset.gender, dataset.shape.to(device), init_pose,
                                                             kw.pop('skin_resolution', (128 + 1, 224 + 1, 64 + 1)), bmins, bmaxs,
                                                             extra_trans, smpl=smpl)
                data = {'ws': skinner.ws.contiguous().cpu(), 'bmins': skinner.b_min.cpu(), 'bmaxs': skinner.b_max.cpu(),
                        'Js': skinner.Js.cpu(), 'parents': skinner.parents, 'init_pose': skinner.init_pose.cpu(),
                        'tmpBodyVs': body_vs.detach().cpu(), 'tmpBodyFs': body_fs.cpu(), 'betas': dataset.shape,
                        'extra_trans': extra_trans, 'bbox_center': skinner.bbox_center.cpu(),
                        'bbox_extend': skinner.bbox_extend.cpu()}
                if int(kw.get('rank', 0)) == 0:         # (run the first start-up on one process: the shape fit shuffles frames)
                    os.makedirs(osp.dirname(skinner_file), exist_ok=True)
                    torch.save(data, skinner_file)
                kw['skinner_state'] = data
    kw.pop('smpl', None)
    kw.pop('skin_resolution', None)
    optNet = cls(conf, device, **kw)
    optNet.visualizer = visualizer
    sdf_initialized = -1
    if root is not None and save_folder is not None:
        sdf_initialized = conf.get_int('train.initial_iters') if 'train.initial_iters' in conf else 0
        stem = 'initial_sdf_idr_%d_%d' % (conf.get_int('sdf_net.multires'), pose_type)
        sdf_file = osp.join(root, save_folder, stem + '.pth')
        if osp.isfile(sdf_file) and use_initial_sdf:
            optNet.sdf.load_state_dict(torch.load(sdf_file, map_location='cpu'))
            for name, net in zip(optNet.garment_names, optNet.garment_nets):
                garment_file = sdf_file.replace('sdf', 'sdf_{}'.format(name))      # :213-216
                assert osp.isfile(garment_file), garment_file
                net.load_state_dict(torch.load(garment_file, map_location='cpu'))
            mesh_file = osp.join(root, save_folder, stem + '.ply')
            if osp.isfile(mesh_file):
                from ..utils import read_ply
                optNet.load_init_sdf_vertices(*read_ply(mesh_file))
            sdf_initialized = -1
        elif sdf_initialized <= 0:
            sdf_initialized = 1200
    return optNet, sdf_initialized
