"""Deformer stack of the hot path (model/Deformer.py of the reference):

  CompositeDeformer  :22-34    sequential application (offset MLP, then LBS)
  Inverse_Fl_Body    :36-122   registered feature line -> its place on the canonical body (undo translation and scale)
  MLPTranslator      :141-206  PE(p) (+) per-frame cond -> 4x512 ReLU MLP -> offset, returns p + offset
  LBSkinner          :216-445  SMPL linear-blend skinning with weights sampled from a 3-D grid
  smooth_weights, compute_lbswField, initialLBSkinner  :533-626  start-up: bake SMPL's per-vertex blend weights into the volume

All dense work goes through librecmv_hip.so: the MLP layers are fused MFMA kernels (ops.linear_act), the
skinning-weight lookup is the double-differentiable HIP sampler (MCAcc.GridSamplerMine3dFunction) on a
channels-last copy of the weight volume (one 96-byte record per corner instead of 24 strided planes),
and the per-point 24->16 blend is an MFMA product against all frames at once followed by a gather, which
removes the reference's per-batch-id Python loop with its `.any().item()` host syncs (:438-443).
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from ..MCAcc.grid_sampler_mine import GridSamplerMine3dFunction
from ..utils.utils import annealing_weights, quat2mat
from .Embedder import get_embedder


def batch_rodrigues(theta):
    """Axis-angle [N,3] -> rotation matrices [N,3,3].

    `smpl_pytorch.util.batch_rodrigues` is NOT vendored in the reference tree (model/Deformer.py:12); this
    is the standard HMR/SMPL-pytorch form it is known by: angle = ||theta + 1e-8||, quaternion
    (cos(a/2), sin(a/2) * theta/angle), quat -> matrix.  PARITY UNPINNED."""
    l1norm = torch.norm(theta + 1e-8, p=2, dim=1)
    angle = torch.unsqueeze(l1norm, -1)
    normalized = torch.div(theta, angle)
    angle = angle * 0.5
    v_cos = torch.cos(angle)
    v_sin = torch.sin(angle)
    quat = torch.cat([v_cos, v_sin * normalized], dim=1)
    return quat2mat(quat)


class KinematicChain(torch.autograd.Function):
    """poses [B,24,3] -> (G [B,24,4,4], A = G . init_pose) in ONE kernel (csrc/kinematic_chain.hip) instead of the
    reference's 23-step Python loop of 4x4 products (model/Deformer.py:384-396)."""

    @staticmethod
    def forward(ctx, poses, js_host, parents_host, init_pose):
        from .. import _lib as L
        poses_c = poses.detach().contiguous().float()
        B = poses_c.shape[0]
        G = torch.empty((B, 24, 4, 4), dtype=torch.float32, device=poses.device)
        A = torch.empty_like(G) if init_pose is not None else None
        ip = init_pose.detach().contiguous() if init_pose is not None else None
        with L.device_guard(poses.device):
            L.check(L.lib().recmv_kinematic_chain_forward(L.ptr(poses_c), js_host, parents_host, L.ptr(ip), L.ptr(G),
                                                          L.ptr(A), B, L.stream_ptr(poses.device)), "kinematic_chain")
        ctx.save_for_backward(poses_c, ip)
        ctx.consts = (js_host, parents_host)
        if A is None:
            return G, G.new_zeros(())
        return G, A

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gG, gA):
        from .. import _lib as L
        poses_c, ip = ctx.saved_tensors
        js_host, parents_host = ctx.consts
        gG = gG.contiguous() if gG is not None else None
        gA = gA.contiguous() if (gA is not None and ip is not None) else None
        gp = torch.empty_like(poses_c)
        with L.device_guard(poses_c.device):
            L.check(L.lib().recmv_kinematic_chain_backward(L.ptr(poses_c), js_host, parents_host, L.ptr(ip), L.ptr(gG),
                                                           L.ptr(gA), L.ptr(gp), poses_c.shape[0],
                                                           L.stream_ptr(poses_c.device)), "kinematic_chain_backward")
        return gp, None, None, None


def _mm4(a, b):
    """Batched small matmul [...,n,k] @ [...,k,m] by broadcasting (no BLAS on the path)."""
    return (a.unsqueeze(-1) * b.unsqueeze(-3)).sum(-2)


class CompositeDeformer(nn.Module):
    def __init__(self, deformers):
        super().__init__()
        self.N = len(deformers)
        self.defs = nn.ModuleList(deformers)

    def forward(self, ps, conds, batch_inds=None, **kwargs):
        """`jet=True` (extension): the offset MLP carries its Jacobian forward (csrc/mlp_jet.hip);
        utils.compute_Jacobian(ps, output, ...) then needs autograd only for the skinning stage."""
        assert (self.N == len(conds))
        out = ps
        jac = None
        for i, (cond, deformer) in enumerate(zip(conds, self.defs)):
            prev = out
            out = deformer(out, cond, batch_inds, **kwargs)
            if kwargs.get('jet', False):
                if i == 0:
                    jac = getattr(out, '_recmv_jac', None)
                elif jac is not None:
                    out._recmv_jac_lazy = (ps, prev, jac[1]) if i == 1 else None
        return out

    @torch.no_grad()
    def value_and_vjp(self, ps, conds, batch_inds, cotangent_fn, **kwargs):
        """d = deformer(ps) and J_d(ps)^T g with g = cotangent_fn(d), without an autograd graph and without
        parameter gradients: explicit forward of each stage, explicit backward to the INPUT.  This is what the
        surface root finder needs per step (utils/FindSurfacePs.py:319-333)."""
        assert (self.N == len(conds))
        out, saved = ps, []
        for cond, deformer in zip(conds, self.defs):
            out, sv = deformer.forward_explicit(out, cond, batch_inds, **kwargs)
            saved.append(sv)
        g = cotangent_fn(out)
        for deformer, sv in zip(reversed(self.defs), reversed(saved)):
            g = deformer.backward_input(sv, g)
        return out, g

    @torch.no_grad()
    def prepare_explicit(self, conds, **kwargs):
        """Build (on the current stream) the cached state the graph-free passes share between callers: the offset
        MLP's chain descriptor with its transposed weights and the posed skeleton."""
        for cond, deformer in zip(conds, self.defs):
            if hasattr(deformer, 'prepare_explicit'):
                deformer.prepare_explicit(cond, **kwargs)

    @torch.no_grad()
    def ray_energy_and_vjp(self, ps, conds, batch_inds, cam, rays, **kwargs):
        """For the root finder: E2 = |(d-c) x v| / |d-c| of d = deformer(ps), its angle (degrees) and
        J_d(ps)^T dE2/dd — four C calls (offset-MLP chain, fused skinning + energy, skinning VJP, MLP chain VJP)."""
        assert self.N == 2, "offset MLP followed by the LBS skinner (model/network.py:282-283)"
        mlp, lbs = self.defs[0], self.defs[1]
        mid, sv0 = mlp.forward_explicit(ps, conds[0], batch_inds, **kwargs)
        d, sv1 = lbs.forward_explicit(mid, conds[1], batch_inds, cam=cam, rays=rays, **kwargs)
        loss2, angle, g_d = sv1[3], sv1[4], sv1[5]
        g = lbs.backward_input(sv1, g_d)
        g = mlp.backward_input(sv0, g)
        return d, loss2, angle, g


class Inverse_Fl_Body(nn.Module):
    """model/Deformer.py:36-122 — takes a feature line that `align_fl` has registered (translated by `rigid_t`, scaled by
    `rigid_scale` about the template line's centroid) back to where its template lies on the canonical body:
    ((v - t) - centre) / scale + centre.  Built from the UNregistered template lines; `set_rigid_center` keeps the
    registered centroids."""

    def __init__(self, cano_fl_meshes, fl_names, rigid_t_list, rigid_scale_list):
        super().__init__()
        self.fl_names = fl_names
        self.v_dirs_dict, self.init_scale_dict, self.verts_dict, self.center_dict = {}, {}, {}, {}
        self.rigid_t_dict, self.rigid_scale_dict, self.rigid_center_dict = {}, {}, {}
        for name, mesh, rigid_t, rigid_s in zip(fl_names, cano_fl_meshes, rigid_t_list, rigid_scale_list):
            v = mesh.verts_packed() if hasattr(mesh, 'verts_packed') else mesh
            center = v.mean(0, keepdim=True)
            v_dirs = (v - center) / ((v - center).norm(dim=1, keepdim=True) + 1e-6)
            self.v_dirs_dict[name] = v_dirs
            self.init_scale_dict[name] = ((v - center) * v_dirs).sum(dim=-1, keepdim=True)
            self.verts_dict[name], self.center_dict[name] = v, center
            self.rigid_t_dict[name], self.rigid_scale_dict[name] = rigid_t, rigid_s

    def set_rigid_center(self, rigid_center_list, fl_names):
        assert len(rigid_center_list) == len(fl_names)
        self.rigid_center_dict = dict(zip(fl_names, rigid_center_list))

    def forward(self, rigid_cano_fl_verts, fl_names):
        out = []
        for verts, name in zip(rigid_cano_fl_verts, fl_names):
            center = self.center_dict[name]
            out.append(((verts - self.rigid_t_dict[name]) - center) / self.rigid_scale_dict[name] + center)
        return out


class MLPTranslator(nn.Module):
    def __init__(self, feature_vector_size, multires, weight_norm=False):
        super().__init__()
        dims = [3 + feature_vector_size, 512, 512, 512, 512, 3]
        self.feature_vector_size = feature_vector_size
        self.embed_fn = None
        self.multires = multires
        if multires > 0:
            embed_fn, input_ch = get_embedder(multires)
            self.embed_fn = embed_fn
            dims[0] = input_ch + feature_vector_size
        self.num_layers = len(dims)
        for l in range(0, self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if weight_norm:
                print('MLPTranslator:weight norm can influence weight initialization, can not produce small '
                      'weights as initialization. Now do not use weight_norm')
            if l == self.num_layers - 2:                                     # zero-translation init :163-166
                torch.nn.init.normal_(lin.weight, mean=0., std=0.001)
                torch.nn.init.constant_(lin.bias, 0.)
            setattr(self, "lin" + str(l), lin)
        self.relu = nn.ReLU()
        self.offset = {}

    def forward(self, ps, conds, batch_inds=None, **kwargs):
        ratio = kwargs['ratio']['deformerRatio']
        offset_type = kwargs.get('offset_type', None)
        if (kwargs.get('jet', False) and ps.is_cuda and ps.dtype == torch.float32 and self.embed_fn is not None
                and torch.is_grad_enabled()):
            return self._forward_jet(ps, conds, batch_inds, ratio, offset_type)
        if self.embed_fn is not None:
            if ratio is None:
                ps = self.embed_fn(ps)
            elif ratio <= 0:
                ps = self.embed_fn(ps, [0. for _ in range(self.multires * 2)])
            else:
                ps = self.embed_fn(ps, annealing_weights(self.multires, ratio))
        if batch_inds is not None:
            # `cond_index` (extension): row of the code table per point where it is not the frame index — the rows of several
            # garments in one block take their codes from the garments' tables stacked (row = frame + N * garment)
            cidx = kwargs.get('cond_index')
            x = torch.cat([ps, ops.gather_rows(conds, batch_inds if cidx is None else cidx)], dim=1)       # deterministic backward
        else:
            x = torch.cat([ps, conds.view(-1, 1, self.feature_vector_size).expand(
                -1, ps.shape[1], self.feature_vector_size)], dim=-1).view(-1, ps.shape[-1] + self.feature_vector_size)
        for l in range(0, self.num_layers - 1):
            lin = getattr(self, "lin" + str(l))
            last = l == self.num_layers - 2
            W = lin.weight
            pad = (-x.shape[1]) % 4
            if pad and x.is_cuda:      # K = 167 -> 168: 16-byte aligned rows for the MFMA kernel's vector loader
                x, W = F.pad(x, (0, pad)), F.pad(W, (0, pad))
            x = ops.linear_act(x, W, lin.bias, ops.ACT_NONE if last else ops.ACT_RELU, 0.0)
        if batch_inds is not None:
            self.offset[offset_type] = x
            return ps[..., :3] + x
        else:
            self.offset[offset_type] = x.view(ps.shape[0], ps.shape[1], 3)
            return ps[..., :3] + x.view(ps.shape[0], ps.shape[1], 3)


def _translator_forward_jet(self, ps, conds, batch_inds, ratio, offset_type):
    """forward() with the Jacobian d out / d ps carried along (one C call): out._recmv_jac = (ps, I + J_offset)."""
    from ..chains import mlp_jet
    pf = self.__dict__.pop('_jet_prefetch', None)
    if pf is not None and pf[0] is ps and pf[2] is conds and pf[3] == ratio and pf[4] is batch_inds:
        # these very points went through the net a moment ago as the second row block of jet_two_blocks()
        out = pf[1]
        self.offset[offset_type] = out - ps[..., :3]
        return out
    ws = None if ratio is None else ([0.] * (self.multires * 2) if ratio <= 0 else
                                     annealing_weights(self.multires, ratio))
    flat = ps.reshape(-1, 3)
    blocks = 0
    if batch_inds is not None:
        cidx = batch_inds
        cond2d = conds
    else:
        cond2d = conds.reshape(-1, self.feature_vector_size)
        cidx = torch.arange(cond2d.shape[0], device=ps.device).repeat_interleave(ps.shape[1])
        blocks = cond2d.shape[0]              # frame-major blocks of ps.shape[1] points each
    nl = self.num_layers - 1
    lins = [getattr(self, "lin" + str(l)) for l in range(nl)]
    Ws = [lin.weight for lin in lins]
    bs = [lin.bias for lin in lins]
    dims = [Ws[0].shape[1]] + [W.shape[0] for W in Ws]
    y, J = mlp_jet(flat, cond2d, cidx.contiguous(), Ws, bs, dims, self.multires, ws, self.feature_vector_size, -1,
                   ops.ACT_RELU, 0.0, True, 3, cond_blocks=blocks)
    out = y.view(ps.shape)
    self.offset[offset_type] = out - ps[..., :3]
    out._recmv_jac = (ps, J + torch.eye(3, device=ps.device, dtype=J.dtype).view(1, 3, 3))
    return out


def _translator_jet_two_blocks(self, pts, conds, ps, batch_inds, ratio, offset_type):
    """ONE jet pass over two row blocks that go through this net with the same parameters and the same code table: `pts` [N, n, 3]
    (frame-major blocks: row block i takes code i — forward(pts, conds, jet=True)) and `ps` [m, 3] with `batch_inds` (forward(ps,
    conds, batch_inds, jet=True)).  Returns the first block's output (its Jacobian attached for utils.compute_Jacobian(pts, out));
    the second block's output is parked and served to the next jet call on `ps` (CompositeDeformer.forward on the converged rays)."""
    from ..chains import mlp_jet
    ws = None if ratio is None else ([0.] * (self.multires * 2) if ratio <= 0 else
                                     annealing_weights(self.multires, ratio))
    cond2d = conds.reshape(-1, self.feature_vector_size)
    n1 = pts.shape[0] * pts.shape[1]
    cidx = torch.cat([torch.arange(cond2d.shape[0], device=pts.device).repeat_interleave(pts.shape[1]), batch_inds.view(-1)])
    flat = torch.cat([pts.reshape(-1, 3), ps.reshape(-1, 3)], dim=0)
    nl = self.num_layers - 1
    lins = [getattr(self, "lin" + str(l)) for l in range(nl)]
    Ws = [lin.weight for lin in lins]
    bs = [lin.bias for lin in lins]
    dims = [Ws[0].shape[1]] + [W.shape[0] for W in Ws]
    y, J = mlp_jet(flat, cond2d, cidx.contiguous(), Ws, bs, dims, self.multires, ws, self.feature_vector_size, -1,
                   ops.ACT_RELU, 0.0, True, 3, cond_blocks=0)
    J = J + torch.eye(3, device=pts.device, dtype=J.dtype).view(1, 3, 3)
    out1 = y[:n1].view(pts.shape)
    out1._recmv_jac = (pts, J[:n1])
    out2 = y[n1:].view(ps.shape)
    out2._recmv_jac = (ps, J[n1:])
    self.offset[offset_type] = out1 - pts[..., :3]
    self.__dict__['_jet_prefetch'] = (ps, out2, conds, ratio, batch_inds)
    return out1


def _translator_chain(self, ratio):
    """MlpChain of the offset MLP for the annealing state `ratio` (cached on the parameters' versions)."""
    from ..chains import MlpChain
    ws = None if ratio is None else ([0.] * (self.multires * 2) if ratio <= 0 else
                                     annealing_weights(self.multires, ratio))
    wl = None if ws is None else tuple(float(w) for w in ws)
    nl = self.num_layers - 1
    lins = [getattr(self, "lin" + str(l)) for l in range(nl)]
    key = (wl,) + tuple((lin.weight._version, lin.weight.data_ptr(), lin.bias._version) for lin in lins)
    hit = self.__dict__.get('_chain_cache')
    if hit is not None and hit[0] == key:
        L.acquire(hit[2])
        return hit[1]
    Ws = [lin.weight.detach().contiguous() for lin in lins]
    bs = [lin.bias.detach() for lin in lins]
    Wts = [_cached_t(self, l, lins[l].weight) for l in range(nl)]
    dims = [Ws[0].shape[1]] + [W.shape[0] for W in Ws]
    ch = MlpChain(Ws, bs, Wts, dims, [W.shape[0] for W in Ws], self.multires, cond_dim=self.feature_vector_size,
                  skip_layer=-1, hidden_act=ops.ACT_RELU, act_param=0.0, residual=True, pe_weights=wl)
    self.__dict__['_chain_cache'] = (key, ch, L.publish(Ws[0].device))     # behind the transposes' launches
    return ch


def _translator_explicit(self, ps, conds, batch_inds, **kwargs):
    """Explicit forward of MLPTranslator for 2-D `ps` with `batch_inds` (the ray path): one C call.
    (The `.offset` side effect of forward() is not reproduced on this path; the autograd passes that follow in
    the iteration overwrite it anyway, model/Deformer.py:201-205.)"""
    ch = _translator_chain(self, kwargs['ratio']['deformerRatio'])
    ps = ps.detach().contiguous()
    slot = kwargs.get('offset_type', None)
    cond_index = kwargs.get('cond_index')            # (rows of several garments in one block: code row = frame + N * garment)
    out = ch.forward(ps, cond=conds.detach(), cond_index=(batch_inds if cond_index is None else cond_index).contiguous(), n_out=3,
                     keep=True, slot=slot)
    return out, (ch, ps, slot)


def _translator_backward_input(self, saved, g_out):
    ch, ps, slot = saved
    return ch.vjp_input(ps, g_out.contiguous(), slot=slot)


def _cached_t(module, l, W):
    cache = module.__dict__.setdefault('_wt_cache', {})
    key = (W._version, W.data_ptr())
    hit = cache.get(l)
    if hit is None or hit[0] != key:
        Wt = W.detach().t().contiguous()
        hit = (key, Wt, L.publish(Wt.device))         # read from several streams: the event behind the transpose travels with it
        cache[l] = hit
    else:
        L.acquire(hit[2])
    return hit[1]


MLPTranslator.prepare_explicit = lambda self, cond, **kwargs: _translator_chain(self, kwargs['ratio']['deformerRatio'])
MLPTranslator._forward_jet = _translator_forward_jet
MLPTranslator.jet_two_blocks = _translator_jet_two_blocks
MLPTranslator.forward_explicit = _translator_explicit
MLPTranslator.backward_input = _translator_backward_input


def getTranslatorNet(device, conf):
    if 'type' in conf:
        return globals()[conf.get_string('type')](conf.get_int('condlen'), multires=conf.get_int('multires')).to(device)
    return MLPTranslator(conf.get_int('condlen'), multires=conf.get_int('multires')).to(device)


class LBSkinner(nn.Module):
    """SMPL-skeleton LBS grid deformer (model/Deformer.py:216-445)."""

    def __init__(self, ws, bmins, bmaxs, Js, parents, init_pose=None, align_corners=False, extra_trans=None,
                 bbox_extend=None, bbox_center=None):
        super().__init__()

        def as_row(v):
            if type(v) is list:
                return torch.tensor(v, dtype=torch.float).view(1, 3)
            if type(v) is np.ndarray:
                return torch.from_numpy(v.astype(np.float32)).view(1, 3)
            return v.view(1, 3)

        self.register_buffer('b_min', as_row(bmins))
        self.register_buffer('b_max', as_row(bmaxs))
        ws = torch.from_numpy(ws.astype(np.float32)) if type(ws) is np.ndarray else ws.to(torch.float)
        # Same logical [1,24,D,H,W] tensor (and state-dict key) as the reference, stored channels-last so
        # a trilinear corner is one contiguous 96-byte record for the HIP sampler.
        self.register_buffer('ws', ws.contiguous(memory_format=torch.channels_last_3d))
        if extra_trans is None:
            extra_trans = torch.full([1, 3], 0.).float()
        self.register_buffer('extra_trans', extra_trans.to(torch.float))
        self.register_buffer('bbox_extend', bbox_extend.to(torch.float))
        self.register_buffer('bbox_center', bbox_center.to(torch.float))
        self.align_corners = align_corners
        assert (align_corners == False)
        self.register_buffer('Js', Js.view(24, 3))
        self.parents = parents
        if init_pose is None:
            self.register_buffer('init_pose', None)
        else:
            if type(init_pose) == np.ndarray:
                init_pose = torch.from_numpy(init_pose.astype(np.float32))
            if init_pose.numel() == 24 * 3:
                init_pose = batch_rodrigues(init_pose.view(-1, 3)).view(24, 3, 3)
                self.init_pose_inverse(init_pose, self.Js)
            else:
                self.register_buffer('init_pose', init_pose.view(24, 4, 4))

    def bbox_size(self):
        margin = torch.tensor([0.15, 0.15, 0.20]).to(self.b_min)
        return self.b_min - margin, self.b_max + margin

    def init_pose_inverse(self, init_pose, Js):
        """World <- rest-pose inverse transforms of the 24 joints (model/Deformer.py:282-306)."""
        resultsR = [init_pose[0]]
        resultsT = [Js[0]]
        for i in range(1, self.parents.shape[0]):
            p = int(self.parents[i])
            j_here = Js[i] - Js[p]
            resultsR.append(resultsR[p].matmul(init_pose[i]))
            resultsT.append(resultsR[p].matmul(j_here.view(-1, 1)).view(-1) + resultsT[p])
        invs = []
        for R, T in zip(resultsR, resultsT):
            inv = torch.zeros(4, 4)
            inv[3, 3] = 1.
            inv[:3, :3] = R.transpose(0, 1)
            inv[:3, 3] = (-T.view(1, -1).matmul(R)).view(-1)
            invs.append(inv)
        self.register_buffer('init_pose', torch.stack(invs, dim=0))

    def _host_consts(self):
        if getattr(self, "_consts", None) is None:
            import ctypes as C
            js = self.Js.detach().cpu().view(-1).tolist()
            par = [int(v) for v in np.asarray(self.parents).reshape(-1)]
            par[0] = -1
            self._consts = ((C.c_float * 72)(*js), (C.c_int32 * 24)(*par))
        return self._consts

    def _chain_fused(self, poses):
        """(results, A) through the fused kernel; A is None when the skinner has no init_pose."""
        js_host, parents_host = self._host_consts()
        G, A = KinematicChain.apply(poses.reshape(-1, 24, 3), js_host, parents_host, self.init_pose)
        return G, (A if self.init_pose is not None else None)

    def _chain(self, poses):
        """Kinematic chain: global 4x4 of every joint, [B,24,4,4] (model/Deformer.py:372-396)."""
        batch_size = poses.shape[0]
        Rs = batch_rodrigues(poses.view(-1, 3)).view(batch_size, 24, 3, 3)
        Js = self.Js.view(1, 24, 3, 1).expand(batch_size, 24, 3, 1)

        def make_A(R, t):
            R_homo = F.pad(R, [0, 0, 0, 1, 0, 0])
            t_homo = torch.cat([t, torch.ones(R.shape[0], 1, 1).to(R.device)], dim=1)
            return torch.cat([R_homo, t_homo], 2)

        results = [make_A(Rs[:, 0], Js[:, 0])]
        for i in range(1, self.parents.shape[0]):
            p = int(self.parents[i])
            A_here = make_A(Rs[:, i], Js[:, i] - Js[:, p])
            results.append(_mm4(results[p], A_here))
        return torch.stack(results, dim=1), Js

    def posedSkeleton(self, conds):
        poses, trans = conds
        assert (poses.shape[0] == trans.shape[0])
        if poses.is_cuda and poses.dtype == torch.float32:
            results, _ = self._chain_fused(poses)
        else:
            results, _ = self._chain(poses)
        return results[:, :, :3, 3]

    def inv_transform_v(self, v, scale_grid, transl):
        v = v - transl[None, None]
        v = v / scale_grid
        v = v * 2
        return v

    def skinning_weights(self, tps):
        """Sampled blend weights [P,24] at canonical points (model/Deformer.py:411-421)."""
        nps = self.inv_transform_v(tps, self.bbox_extend, self.bbox_center).view(-1, 3)
        return GridSamplerMine3dFunction.apply(self.ws, nps.reshape(1, 1, 1, -1, 3)).view(-1, nps.shape[0]).transpose(0, 1)

    def forward(self, ps, conds, batch_inds=None, **kwargs):
        if type(ps) == list:
            tps, ps = ps
        else:
            tps = ps
        poses, trans = conds
        trans = trans + self.extra_trans
        batch_size = poses.shape[0]
        assert (batch_size == trans.shape[0])
        if poses.is_cuda and poses.dtype == torch.float32 and self.init_pose is not None:
            _, A = self._chain_fused(poses)
            results = Js = None
        else:
            results, Js = self._chain(poses)
            A = None
        if A is not None:
            pass
        elif self.init_pose is None:
            Js_w0 = torch.cat([Js, torch.zeros(batch_size, 24, 1, 1).to(poses.device)], dim=2)
            init_bone = _mm4(results, Js_w0)
            init_bone = F.pad(init_bone, [3, 0, 0, 0, 0, 0, 0, 0])
            A = results - init_bone
        else:
            A = _mm4(results, self.init_pose.view(1, 24, 4, 4).expand(batch_size, 24, 4, 4))
        if batch_inds is None:
            batch_size2, pnum, _ = ps.shape
            assert (batch_size == batch_size2)
            flat = ps.reshape(-1, 3)
            binds = torch.arange(batch_size, device=ps.device).repeat_interleave(pnum)
        else:
            flat = ps.reshape(-1, 3)
            assert (batch_inds.numel() == flat.shape[0])
            binds = batch_inds
        fused = (tps is ps and flat.is_cuda and flat.dtype == torch.float32 and not kwargs.get('jet', False)
                 and torch.is_grad_enabled() and A.shape[1:] == (24, 4, 4))
        jet = (kwargs.get('jet', False) and tps is ps and flat.is_cuda and flat.dtype == torch.float32 and torch.is_grad_enabled()
               and A.shape[1:] == (24, 4, 4) and not self.ws.requires_grad and os.environ.get('RECMV_LBS_JET', '1') != '0')
        J = None
        if fused:
            # first-order path: one kernel forward, fused VJP kernels backward (csrc/lbs_fused.hip)
            from ..chains import LbsFused
            v = LbsFused.apply(flat, A, trans, binds.contiguous(), self._lbs_grid(), self._blend_classic)
        elif jet:
            # value + Jacobian d v / d ps in one kernel, once-differentiable (csrc/lbs_fused.hip: lbs_jet_*): utils.compute_Jacobian(ps,
            # v) finds it on the output instead of differentiating the composition below three times with create_graph
            from ..chains import LbsJet
            v, J = LbsJet.apply(flat, A, trans, binds.contiguous(), self._lbs_grid())
        else:
            v = self._blend_classic(flat, A, trans, binds, tps=tps)
        if batch_inds is None:
            v = v.view(batch_size, pnum, 3)
        if J is not None:
            v._recmv_jac = (ps, J)
        return v

    def _blend_classic(self, flat, A, trans, binds, tps=None):
        """v = T(p) [p;1] + trans with T = sum_j w_j A_j as a composition of differentiable ops (any order)."""
        batch_size = A.shape[0]
        ps_ws = self.skinning_weights(flat if tps is None else tps)             # [P,24]
        # T[p] = sum_j w[p,j] * A[b_p, j]  — one MFMA product against every frame, then gather by frame
        Ball = A.reshape(batch_size, 24, 16).permute(0, 2, 1).reshape(batch_size * 16, 24)
        Tall = ops.MatmulNT.apply(ps_ws, Ball).view(-1, batch_size, 16)
        T = Tall.gather(1, binds.view(-1, 1, 1).expand(-1, 1, 16)).view(-1, 4, 4)
        v = (T[:, :3, :3] * flat.unsqueeze(-2)).sum(-1) + T[:, :3, 3]
        return v + ops.gather_rows(trans, binds)

    # -- graph-free passes on ray points: fused kernels (csrc/lbs_fused.hip) ----------------------------
    def _lbs_grid(self):
        from ..chains import lbs_grid
        hit = self.__dict__.get('_grid_cache')
        if hit is None or hit[0] != self.ws.data_ptr():
            center = self.bbox_center.detach().cpu().view(-1).tolist()
            # (a skinner baked by the reference normalises with ONE extent — a cube, model/Deformer.py:609 —, the
            # synthetic rig with one per axis)
            scale = (2.0 / self.bbox_extend.detach().cpu().view(-1).expand(3)).tolist()
            hit = (self.ws.data_ptr(), lbs_grid(self.ws, center, scale))
            self.__dict__['_grid_cache'] = hit
        return hit[1]

    @torch.no_grad()
    def _posed(self, poses, trans):
        """(A [B,24,4,4], trans + extra_trans) for this pose tensor; cached while the same tensor is passed again
        (the root finder evaluates the deformer up to 21 times with the same per-frame parameters)."""
        hit = self.__dict__.get('_posed_cache')
        if hit is not None and hit[0] is poses and hit[1] == poses._version and hit[2] is trans \
                and hit[3] == trans._version:
            L.acquire(hit[6])                        # (the garments' root finders read it on their own streams)
            return hit[4], hit[5]
        _, A = self._chain_fused(poses.detach())
        A = A.contiguous()
        t = (trans.detach() + self.extra_trans).contiguous()
        self.__dict__['_posed_cache'] = (poses, poses._version, trans, trans._version, A, t, L.publish(A.device))
        return A, t

    @torch.no_grad()
    def prepare_explicit(self, conds, **kwargs):
        self._posed(conds[0], conds[1])
        self._lbs_grid()

    @torch.no_grad()
    def forward_explicit(self, ps, conds, batch_inds, cam=None, rays=None, **kwargs):
        """Explicit (graph-free) forward for 2-D `ps` with `batch_inds`: one fused kernel.  With `rays` and `cam` it
        also returns the ray energy, its angle and its gradient wrt the deformed point."""
        from .. import chains
        poses, trans = conds
        A, t = self._posed(poses, trans)
        ps = ps.detach().contiguous()
        frame = batch_inds.contiguous()
        d, loss2, angle, g_d = chains.lbs_forward(ps, frame, A, t, self._lbs_grid(), cam, rays)
        return d, (ps, frame, A, loss2, angle, g_d)

    @torch.no_grad()
    def backward_input(self, saved, g_v):
        from .. import chains
        ps, frame, A = saved[0], saved[1], saved[2]
        return chains.lbs_vjp_input(ps, frame, A, self._lbs_grid(), g_v.contiguous())


# ------------------------------------------------------------------------------------------ start-up: baking the skinner
def getSMPL(gender):
    """The SMPL body model of the reference (`smpl_pytorch.SMPL.getSMPL`, un-vendored; its pkl files are not redistributable).
    Used only by the start-up steps that build a skinner from scratch; pass `smpl=` to them to supply any object with the same
    surface (`__call__(betas, poses, True) -> (verts, _, _)`, `skeleton(betas, True) -> (Js, _)`, `weight`, `parents`, `faces`,
    `joint_regressor`)."""
    try:
        from smpl_pytorch.SMPL import getSMPL as ref_get
    except ImportError as e:
        raise ImportError("the SMPL model (smpl_pytorch + its model files) is not part of this package: pass smpl=<model> or "
                          "start from an initial_skinner_<pose type>.pth baked by the reference") from e
    return ref_get(gender)


def smooth_weights(weights, times=3):
    """model/Deformer.py:533-544 — `times` passes of: pull every interior voxel 30 % towards the mean of its six neighbours,
    renormalise the channels of every voxel to sum 1."""
    for _ in range(times):
        c = weights[:, :, 1:-1, 1:-1, 1:-1]
        mean = (weights[:, :, 2:, 1:-1, 1:-1] + weights[:, :, :-2, 1:-1, 1:-1] + weights[:, :, 1:-1, 2:, 1:-1]
                + weights[:, :, 1:-1, :-2, 1:-1] + weights[:, :, 1:-1, 1:-1, 2:] + weights[:, :, 1:-1, 1:-1, :-2]) / 6.0
        weights[:, :, 1:-1, 1:-1, 1:-1] = (c - mean) * 0.7 + mean
        weights = weights / weights.sum(1, keepdim=True)
    return weights


def compute_lbswField(bmins, bmaxs, resolutions, smpl_verts, smpl_ws, align_corners=False, mean_neighbor=5, smooth_times=30):
    """model/Deformer.py:546-591 — blend weights of every voxel centre of a (W, H, D) grid over [bmins, bmaxs]: inverse-distance
    mean of the `mean_neighbor` nearest template vertices' weights (distances clamped to [1e-4, 1]), then `smooth_weights`.
    Returns [1, J, D, H, W]."""
    device = smpl_verts.device
    bmins = torch.as_tensor(bmins).float().to(device).view(1, -1)
    bmaxs = torch.as_tensor(bmaxs).float().to(device).view(1, -1)
    W, H, D = resolutions
    res = torch.tensor(resolutions).float().to(device).view(1, -1)
    gridD, gridH, gridW = torch.meshgrid(torch.arange(D, device=device), torch.arange(H, device=device),
                                         torch.arange(W, device=device), indexing='ij')
    coords = torch.stack([gridW, gridH, gridD]).view(3, -1).t().float()
    unit = coords / (res - 1) if align_corners else coords / res + (1.0 / res) / 2
    centres = unit * (bmaxs - bmins) + bmins
    rows = []
    for chunk in torch.split(centres, 50000):
        dists, indices = (chunk[:, None, :] - smpl_verts[None, :, :]).norm(dim=-1).topk(mean_neighbor, dim=-1, largest=False)
        w = 1. / torch.clamp(dists, 0.0001, 1.)
        w = w / w.sum(-1, keepdim=True)
        rows.append((smpl_ws[indices.view(-1)] * w.view(-1, 1)).reshape(w.shape[0], mean_neighbor, -1).sum(1))
    field = torch.cat(rows, dim=0).transpose(0, 1).reshape(1, -1, D, H, W)
    return smooth_weights(field, smooth_times)


def initialLBSkinner(gender, shape, pose, resolution, bmins=None, bmaxs=None, extra_trans=None, smpl=None):
    """model/Deformer.py:594-626 — the skinner of a capture from scratch: SMPL posed into the A-pose `pose` with the fitted
    `shape`, its per-vertex blend weights baked into a `resolution` = (W, H, D) volume over the template's bounding box
    (30 nearest vertices, 30 smoothing passes); normalisation box = 1.1 x the largest extent around the box centre.
    Returns (LBSkinner, template vertices [V,3], template faces [F,3]).  (`bmins` / `bmaxs` are accepted and unused: the
    reference only runs with the adaptive box — with a given box it stops on an undefined name.)"""
    smpl = (smpl if smpl is not None else getSMPL(gender)).to(shape.device)
    Js, _ = smpl.skeleton(shape.view(1, -1), True)
    verts, _, _ = smpl(shape.view(1, -1), pose.view(1, 24, 3), True)
    verts = verts.view(-1, 3)
    lo, hi = verts.min(0).values, verts.max(0).values
    ws = compute_lbswField(lo.tolist(), hi, resolution, verts, smpl.weight.view(verts.shape[0], 24), align_corners=False,
                           mean_neighbor=30, smooth_times=30)
    skinner = LBSkinner(ws, lo, hi, Js, smpl.parents, init_pose=pose, align_corners=False, extra_trans=extra_trans,
                        bbox_extend=(hi - lo).max() * 1.1, bbox_center=(lo + hi) / 2)
    return skinner, verts, torch.as_tensor(smpl.faces, dtype=torch.long, device=verts.device)
