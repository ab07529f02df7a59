"""CPU port of the hot loop, for bench.py's `cpu_baseline` leg and for CPU tests of the host logic.

TEST / BASELINE INFRASTRUCTURE.  The product (rec-mv_amd/recmv) has no CPU path: its ops refuse CPU tensors.
`install()` swaps, from the OUTSIDE, the handful of entry points through which recmv reaches
librecmv_hip.so for plain-torch / C-oracle equivalents, so the very same loop code (recmv/loop.py) runs on
host cores:

    recmv.ops.linear_act / MatmulNT / gemm_nt     -> torch (F.linear + activation, @)      [torch-CPU sgemm]
    recmv.GridSamplerMine.forward/backward/dbackward -> oracle.gs3d_*                       [C oracle, OpenMP]
    recmv.FastMinv.Fast3x3Minv(_backward)          -> oracle.inv3x3_*
    recmv.interp2x_boundary3d.forward/backward     -> oracle.interp2x_*
    recmv.MCGpu.mc_gpu                             -> oracle.mc

Nothing in rec-mv_amd/ imports this module; `uninstall()` restores the product bindings.
"""
import torch
import torch.nn.functional as F

from . import oracle as orc

_saved = {}


def _act(z, act, p):
    from recmv import ops
    if act == ops.ACT_RELU:
        return torch.relu(z)
    if act == ops.ACT_SOFTPLUS:
        return F.softplus(z, beta=p)
    if act == ops.ACT_TANH:
        return torch.tanh(z)
    return z


def _linear_act(x, W, b=None, act=0, act_param=0.0):
    return _act(F.linear(x, W, b), act, act_param)


def _gemm_nt(A, B, bias=None, act=0, act_param=0.0, out_scale=1.0, out=None):
    r = _act(F.linear(A, B, bias), act, act_param) * out_scale
    if out is not None:
        out.copy_(r)
        return out
    return r


class _MatmulNT:
    @staticmethod
    def apply(A, B):
        return A @ B.t()


class _MatmulTN:
    @staticmethod
    def apply(A, B):
        return A.t() @ B


def _rasterize_meshes(face_verts, mesh_first_face, mesh_num_faces, image_size, blur_radius=0.0, faces_per_pixel=1,
                      perspective_correct=True, clip_barycentric_coords=False, cull_backfaces=False,
                      max_faces_per_mesh=None):
    from recmv.raster import Fragments
    assert faces_per_pixel == 1 and not clip_barycentric_coords
    return Fragments(*orc.rasterize_meshes(face_verts, mesh_first_face, mesh_num_faces, image_size, blur_radius,
                                           perspective_correct, cull_backfaces, scan=True))


class _RasterizePointsCPU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, first, num, H, W, radius, K):
        idx, zbuf, dists = orc.rasterize_points(points.detach(), first, num, (H, W), radius, K, scan=True)
        ctx.save_for_backward(points.detach(), idx)
        ctx.mark_non_differentiable(idx)
        return idx, zbuf, dists

    @staticmethod
    def backward(ctx, _gi, g_zbuf, g_dists):
        points, idx = ctx.saved_tensors
        return orc.rasterize_points_backward(points, idx, g_dists, g_zbuf), None, None, None, None, None, None


def _rasterize_points(points, cloud_first_point, cloud_num_points, image_size, radius, points_per_pixel=8,
                      max_points_per_cloud=None):
    from recmv.raster import PointFragments
    return PointFragments(*_RasterizePointsCPU.apply(points, cloud_first_point, cloud_num_points, int(image_size[0]),
                                                     int(image_size[1]), float(radius), int(points_per_pixel)))


class _AlphaCompositeCPU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, alphas, features):
        ctx.save_for_backward(idx, alphas.detach(), features.detach())
        return orc.alpha_composite_forward(idx, alphas.detach(), features.detach())

    @staticmethod
    def backward(ctx, g):
        idx, alphas, features = ctx.saved_tensors
        ga, gf = orc.alpha_composite_backward(idx, alphas, features, g, ctx.needs_input_grad[2])
        return None, ga, gf


def install():
    import recmv.FastMinv as FM
    import recmv.GridSamplerMine as GS
    import recmv.MCGpu as MC
    import recmv.interp2x_boundary3d as IP
    import recmv.loop as LP
    import recmv.ops as ops
    import recmv.raster as RS
    import recmv.utils.utils as UU
    if _saved:
        return
    _saved.update(dict(la=ops.linear_act, nt=ops.MatmulNT, tn=ops.MatmulTN, gnt=ops.gemm_nt,
                       gf=GS.forward, gb=GS.backward, gd=GS.dbackward, fm=FM.Fast3x3Minv, fmb=FM.Fast3x3Minv_backward,
                       uu=UU.Fast3x3Minv, uub=UU.Fast3x3Minv_backward, lp=LP.Fast3x3Minv, ipf=IP.forward,
                       ipb=IP.backward, mc=MC.mc_gpu, rs=RS.rasterize_meshes, rp=RS.rasterize_points,
                       ac=RS.alpha_composite, acd=RS.alpha_composite_dists))
    ops.linear_act, ops.MatmulNT, ops.MatmulTN, ops.gemm_nt = _linear_act, _MatmulNT, _MatmulTN, _gemm_nt
    GS.forward = lambda i, g, a, b: orc.gs3d_forward(i, g)
    GS.backward = lambda i, g, go, a, b, need_grad_input=True: orc.gs3d_backward(i, g, go, need_grad_input)
    GS.dbackward = lambda gI, gG, i, g, go, a, b, need_grad_input=True: orc.gs3d_dbackward(
        None if gI is None else gI.contiguous(), gG.contiguous(), i, g, go, need_grad_input)
    FM.Fast3x3Minv = UU.Fast3x3Minv = LP.Fast3x3Minv = orc.inv3x3_forward
    FM.Fast3x3Minv_backward = UU.Fast3x3Minv_backward = orc.inv3x3_backward
    IP.forward, IP.backward = orc.interp2x_forward, orc.interp2x_backward
    MC.mc_gpu = orc.mc
    RS.rasterize_meshes = _rasterize_meshes
    RS.rasterize_points = _rasterize_points
    RS.alpha_composite = _AlphaCompositeCPU.apply
    RS.alpha_composite_dists = lambda idx, dists, radius, features: _AlphaCompositeCPU.apply(
        idx, 1 - dists / (radius * radius), features)


def uninstall():
    import recmv.FastMinv as FM
    import recmv.GridSamplerMine as GS
    import recmv.MCGpu as MC
    import recmv.interp2x_boundary3d as IP
    import recmv.loop as LP
    import recmv.ops as ops
    import recmv.raster as RS
    import recmv.utils.utils as UU
    if not _saved:
        return
    ops.linear_act, ops.MatmulNT, ops.MatmulTN, ops.gemm_nt = _saved['la'], _saved['nt'], _saved['tn'], _saved['gnt']
    GS.forward, GS.backward, GS.dbackward = _saved['gf'], _saved['gb'], _saved['gd']
    FM.Fast3x3Minv, FM.Fast3x3Minv_backward = _saved['fm'], _saved['fmb']
    UU.Fast3x3Minv, UU.Fast3x3Minv_backward = _saved['uu'], _saved['uub']
    LP.Fast3x3Minv = _saved['lp']
    IP.forward, IP.backward = _saved['ipf'], _saved['ipb']
    MC.mc_gpu = _saved['mc']
    RS.rasterize_meshes = _saved['rs']
    RS.rasterize_points, RS.alpha_composite, RS.alpha_composite_dists = _saved['rp'], _saved['ac'], _saved['acd']
    _saved.clear()
