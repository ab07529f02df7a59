"""Deformer stack of the hot path (model/Deformer.py of the reference):

  CompositeDeformer  :22-34    sequential application (offset MLP, then LBS)
  MLPTranslator      :141-206  PE(p) (+) per-frame cond -> 4x512 ReLU MLP -> offset, returns p + offset
  LBSkinner          :216-445  SMPL linear-blend skinning with weights sampled from a 3-D grid

All dense work goes through librecmv_hip.so: the MLP layers are fused MFMA kernels (ops.linear_act), the
skinning-weight lookup is the double-differentiable HIP sampler (MCAcc.GridSamplerMine3dFunction) on a
channels-last copy of the weight volume (one 96-byte record per corner instead of 24 strided planes),
and the per-point 24->16 blend is an MFMA product against all frames at once followed by a gather, which
removes the reference's per-batch-id Python loop with its `.any().item()` host syncs (:438-443).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

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
        with torch.cuda.device(poses.device):
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
        with torch.cuda.device(poses_c.device):
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
        assert (self.N == len(conds))
        out = ps
        for cond, deformer in zip(conds, self.defs):
            out = deformer(out, cond, batch_inds, **kwargs)
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
        if self.embed_fn is not None:
            if ratio is None:
                ps = self.embed_fn(ps)
            elif ratio <= 0:
                ps = self.embed_fn(ps, [0. for _ in range(self.multires * 2)])
            else:
                ps = self.embed_fn(ps, annealing_weights(self.multires, ratio))
        if batch_inds is not None:
            x = torch.cat([ps, conds.index_select(0, batch_inds)], dim=1)
        else:
            x = torch.cat([ps, conds.view(-1, 1, self.feature_vector_size).expand(
                -1, ps.shape[1], self.feature_vector_size)], dim=-1).view(-1, ps.shape[-1] + self.feature_vector_size)
        for l in range(0, self.num_layers - 1):
            lin = getattr(self, "lin" + str(l))
            last = l == self.num_layers - 2
            x = ops.linear_act(x, lin.weight, lin.bias, ops.ACT_NONE if last else ops.ACT_RELU, 0.0)
        if batch_inds is not None:
            self.offset[offset_type] = x
            return ps[..., :3] + x
        else:
            self.offset[offset_type] = x.view(ps.shape[0], ps.shape[1], 3)
            return ps[..., :3] + x.view(ps.shape[0], ps.shape[1], 3)


def _translator_explicit(self, ps, conds, batch_inds, **kwargs):
    """Explicit forward of MLPTranslator for 2-D `ps` with `batch_inds` (the ray path)."""
    ratio = kwargs['ratio']['deformerRatio']
    ws = None if ratio is None else ([0.] * (self.multires * 2) if ratio <= 0 else
                                     annealing_weights(self.multires, ratio))
    wl = None if ws is None else tuple(float(w) for w in ws)
    P, dev = ps.shape[0], ps.device
    ps = ps.detach().contiguous()
    d_pe = 3 + 6 * self.multires
    d_in = d_pe + self.feature_vector_size
    d_inp = (d_in + 3) // 4 * 4
    x = torch.zeros((P, d_inp), dtype=torch.float32, device=dev)
    ops.posenc(ps, self.multires, wl, 1.0, out=x[:, :d_pe], ld_fill=d_pe)
    x[:, d_pe:d_in] = conds.detach().index_select(0, batch_inds)
    h = x[:, :d_in]
    acts = []
    nl = self.num_layers - 1
    for l in range(nl):
        lin = getattr(self, "lin" + str(l))
        h = ops.gemm_nt(h, lin.weight.detach(), lin.bias.detach(), ops.ACT_NONE if l == nl - 1 else ops.ACT_RELU)
        acts.append(h)
    off = acts[-1]
    self.offset[kwargs.get('offset_type', None)] = off
    return ps + off, (ps, acts, wl, d_pe)


def _translator_backward_input(self, saved, g_out):
    ps, acts, wl, d_pe = saved
    nl = self.num_layers - 1
    g = g_out.contiguous()
    for l in range(nl - 1, -1, -1):
        lin = getattr(self, "lin" + str(l))
        if l < nl - 1:
            g = ops.act_grad(g, acts[l], ops.ACT_RELU, 0.0)
        Wt = _cached_t(self, l, lin.weight)
        g = ops.gemm_nt(g, Wt)
    g_pe = g[:, :d_pe].contiguous()
    g_pe[:, :3] += g_out                              # the `ps[..., :3] + offset` residual path
    return ops._pe_vjp(ps, g_pe, None, self.multires, wl)


def _cached_t(module, l, W):
    cache = module.__dict__.setdefault('_wt_cache', {})
    key = (W._version, W.data_ptr())
    hit = cache.get(l)
    if hit is None or hit[0] != key:
        hit = (key, W.detach().t().contiguous())
        cache[l] = hit
    return hit[1]


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
        ps_ws = self.skinning_weights(tps)                                     # [P,24]

        if batch_inds is None:
            batch_size2, pnum, _ = ps.shape
            assert (batch_size == batch_size2)
            flat = ps.reshape(-1, 3)
            binds = torch.arange(batch_size, device=ps.device).repeat_interleave(pnum)
        else:
            flat = ps.reshape(-1, 3)
            assert (batch_inds.numel() == flat.shape[0])
            binds = batch_inds
        # T[p] = sum_j w[p,j] * A[b_p, j]  — one MFMA product against every frame, then gather by frame
        Ball = A.reshape(batch_size, 24, 16).permute(0, 2, 1).reshape(batch_size * 16, 24)
        Tall = ops.MatmulNT.apply(ps_ws, Ball).view(-1, batch_size, 16)
        T = Tall.gather(1, binds.view(-1, 1, 1).expand(-1, 1, 16)).view(-1, 4, 4)
        v = (T[:, :3, :3] * flat.unsqueeze(-2)).sum(-1) + T[:, :3, 3]
        v = v + trans.index_select(0, binds)
        if batch_inds is None:
            return v.view(batch_size, pnum, 3)
        return v

    @torch.no_grad()
    def forward_explicit(self, ps, conds, batch_inds, **kwargs):
        """Explicit (graph-free) forward for 2-D `ps` with `batch_inds`; see CompositeDeformer.value_and_vjp."""
        from .. import GridSamplerMine
        poses, trans = conds
        trans = trans.detach() + self.extra_trans
        B = poses.shape[0]
        _, A = self._chain_fused(poses.detach())
        ps = ps.detach().contiguous()
        P = ps.shape[0]
        scale = 2.0 / self.bbox_extend.view(1, 3)
        nps = ((ps - self.bbox_center.view(1, 3)) * scale).contiguous()
        grid = nps.view(1, 1, 1, P, 3)
        w = GridSamplerMine.forward(self.ws, grid, 0, 1).view(24, P).t().contiguous()       # [P,24]
        Ball = A.reshape(B, 24, 16).permute(0, 2, 1).reshape(B * 16, 24).contiguous()
        Tall = ops.gemm_nt(w, Ball).view(P, B, 16)
        T = Tall.gather(1, batch_inds.view(-1, 1, 1).expand(-1, 1, 16)).view(P, 4, 4)
        v = (T[:, :3, :3] * ps.unsqueeze(-2)).sum(-1) + T[:, :3, 3] + trans.index_select(0, batch_inds)
        return v, (ps, grid, T, A, batch_inds, scale, B)

    @torch.no_grad()
    def backward_input(self, saved, g_v):
        from .. import GridSamplerMine
        ps, grid, T, A, batch_inds, scale, B = saved
        P = ps.shape[0]
        g_p = (T[:, :3, :3] * g_v.unsqueeze(-1)).sum(-2)                                    # T33^T g
        # g_w[p,j] = sum_{i<3,k} g_v[p,i] * A[b_p,j,i,k] * [ps;1][k]
        ph = torch.cat([ps, torch.ones(P, 1, device=ps.device)], dim=1)
        gT = torch.zeros((P, 4, 4), dtype=torch.float32, device=ps.device)
        gT[:, :3, :] = g_v.unsqueeze(-1) * ph.unsqueeze(-2)
        gTall = torch.zeros((P, B, 16), dtype=torch.float32, device=ps.device)
        gTall.scatter_(1, batch_inds.view(-1, 1, 1).expand(-1, 1, 16), gT.view(P, 1, 16))
        BallT = A.reshape(B, 24, 16).permute(1, 0, 2).reshape(24, B * 16).contiguous()
        g_w = ops.gemm_nt(gTall.view(P, B * 16), BallT)                                     # [P,24]
        go = g_w.t().contiguous().view(1, 24, 1, 1, P)
        _, g_grid = GridSamplerMine.backward(self.ws, grid, go, 0, 1, need_grad_input=False)
        return g_p + g_grid.view(P, 3) * scale
