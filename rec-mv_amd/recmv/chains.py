"""Whole-pass launch chains over librecmv_hip.so (csrc/mlp_chain.hip, csrc/lbs_fused.hip).

One C call enqueues every kernel of a graph-free pass — the SDF net with its input gradient, the deformer's
offset MLP with a vector-Jacobian product to its input, the fused skinning of ray points — instead of ~30 Python
level launches each.  These are the passes the surface root finder repeats up to 20 times per iteration and
garment (utils/FindSurfacePs.py:273-353 of the reference) and the no-grad grid queries of Seg3dLossless.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class MlpChain:
    """A recmv_mlp descriptor over tensors that the caller keeps alive (weights are referenced, not copied)."""

    def __init__(self, weights, biases, weights_t, dims, rows, multires, cond_dim=0, skip_layer=-1,
                 hidden_act=L.ACT_RELU, act_param=0.0, residual=False, pe_weights=None):
        n = len(weights)
        assert n <= L.MLP_MAX_LAYERS and len(dims) == n + 1 and len(rows) == n
        self.device = weights[0].device
        self._keep = (list(weights), list(biases), list(weights_t) if weights_t is not None else None)
        m = L.Mlp()
        m.n_layers, m.multires, m.cond_dim, m.skip_layer = n, multires, cond_dim, skip_layer
        m.hidden_act, m.residual, m.act_param = hidden_act, int(bool(residual)), float(act_param)
        for l in range(n):
            W = weights[l]
            assert W.is_contiguous() and W.dtype == torch.float32 and W.shape == (rows[l], dims[l]), (W.shape, l)
            m.W[l] = W.data_ptr()
            m.bias[l] = biases[l].data_ptr() if biases[l] is not None else None
            if weights_t is not None:
                Wt = weights_t[l]
                assert Wt.is_contiguous() and Wt.shape == (dims[l], rows[l])
                m.Wt[l] = Wt.data_ptr()
            m.rows[l] = rows[l]
        for l in range(n + 1):
            m.dims[l] = dims[l]
        for i in range(32):
            m.pe_weights[i] = float(pe_weights[i]) if (pe_weights is not None and i < len(pe_weights)) else 1.0
        self.m = m
        self.n_layers = n
        self.rows_last = rows[-1]
        self._ws = {}

    def _workspace(self, P, keep, slot=None):
        """Activation workspace; `slot` separates concurrent users of one chain (e.g. two garments evaluated on two
        streams through the same deformer MLP)."""
        need = int(L.lib().recmv_mlp_workspace_bytes(C.byref(self.m), P, keep))
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
            self._ws[slot] = ws
        return ws

    def forward(self, x, cond=None, cond_index=None, n_out=None, keep=False, out=None, slot=None):
        """x [P,3] -> [P, n_out] (the first n_out outputs of the last layer)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 3 and x.is_contiguous()
        P = x.shape[0]
        n_out = self.rows_last if n_out is None else n_out
        if out is None:
            out = torch.empty((P, n_out), dtype=torch.float32, device=x.device)
        ws = self._workspace(P, int(keep), slot)
        ld_cond = 0
        if cond is not None:
            assert cond.dtype == torch.float32 and cond.stride(-1) == 1 and cond.dim() == 2
            ld_cond = cond.stride(0)
            if cond_index is not None:
                assert cond_index.dtype == torch.int64 and cond_index.is_contiguous() and cond_index.numel() == P
        with torch.cuda.device(x.device):
            L.check(L.lib().recmv_mlp_forward(C.byref(self.m), L.ptr(x), L.ptr(cond), ld_cond, L.ptr(cond_index), P,
                                              n_out, L.ptr(out), out.stride(0) if P > 1 else n_out, L.ptr(ws), ws.numel(),
                                              int(keep), L.stream_ptr(x.device)), "mlp_forward")
        return out

    def vjp_input(self, x, g_out=None, n_out=None, slot=None):
        """J(x)^T g_out -> [P,3]; call after forward(keep=True) with the same x.  g_out None = ones (n_out 1)."""
        P = x.shape[0]
        n_out = (1 if g_out is None else g_out.shape[1]) if n_out is None else n_out
        gx = torch.empty((P, 3), dtype=torch.float32, device=x.device)
        ws = self._workspace(P, 1, slot)
        ldg = 0
        if g_out is not None:
            assert g_out.dtype == torch.float32 and g_out.stride(1) == 1 and g_out.shape == (P, n_out)
            ldg = g_out.stride(0) if P > 1 else n_out
        with torch.cuda.device(x.device):
            L.check(L.lib().recmv_mlp_vjp_input(C.byref(self.m), L.ptr(x), P, n_out, L.ptr(g_out), ldg, L.ptr(gx),
                                                L.ptr(ws), ws.numel(), L.stream_ptr(x.device)), "mlp_vjp_input")
        return gx


def lbs_grid(ws_volume, center, scale):
    """recmv_lbs_grid over a channels-last [1,24,D,H,W] skinning volume; center/scale: python floats (3 each)."""
    assert ws_volume.dim() == 5 and ws_volume.shape[0] == 1 and ws_volume.shape[1] == 24
    assert ws_volume.is_contiguous(memory_format=torch.channels_last_3d) and ws_volume.dtype == torch.float32
    g = L.LbsGrid()
    g.volume = ws_volume.data_ptr()
    g.D, g.H, g.W = ws_volume.shape[2], ws_volume.shape[3], ws_volume.shape[4]
    for i in range(3):
        g.center[i] = float(center[i])
        g.scale[i] = float(scale[i])
    return g


def lbs_forward(ps, frame, A, trans, grid, cam=None, rays=None):
    """d [P,3] (and, with rays, (loss2 [P], angle [P], g_d [P,3])) — recmv_lbs_forward."""
    P, B = ps.shape[0], A.shape[0]
    dev = ps.device
    d = torch.empty((P, 3), dtype=torch.float32, device=dev)
    loss2 = angle = g_d = None
    if rays is not None:
        loss2 = torch.empty(P, dtype=torch.float32, device=dev)
        angle = torch.empty(P, dtype=torch.float32, device=dev)
        g_d = torch.empty((P, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().recmv_lbs_forward(L.ptr(ps), L.ptr(frame), P, L.ptr(A), L.ptr(trans), B, C.byref(grid),
                                          L.ptr(cam), L.ptr(rays), L.ptr(d), L.ptr(loss2), L.ptr(angle), L.ptr(g_d),
                                          L.stream_ptr(dev)), "lbs_forward")
    return d, loss2, angle, g_d


def lbs_vjp_input(ps, frame, A, grid, g_d):
    P, B = ps.shape[0], A.shape[0]
    g_p = torch.empty((P, 3), dtype=torch.float32, device=ps.device)
    with torch.cuda.device(ps.device):
        L.check(L.lib().recmv_lbs_vjp_input(L.ptr(ps), L.ptr(frame), P, L.ptr(A), B, C.byref(grid), L.ptr(g_d),
                                            L.ptr(g_p), L.stream_ptr(ps.device)), "lbs_vjp_input")
    return g_p


def rootfind_update(p, f, gf, loss2, angle, gd, unfinished, counter, dthreshold, athreshold, w1, w2, do_update):
    with torch.cuda.device(p.device):
        L.check(L.lib().recmv_rootfind_update(L.ptr(p), L.ptr(f), L.ptr(gf), L.ptr(loss2), L.ptr(angle), L.ptr(gd),
                                              L.ptr(unfinished), L.ptr(counter), p.shape[0], float(dthreshold),
                                              float(athreshold), float(w1), float(w2), int(do_update),
                                              L.stream_ptr(p.device)), "rootfind_update")
