"""Whole-pass launch chains over librecmv_hip.so (csrc/mlp_chain.hip, csrc/lbs_fused.hip).

One C call enqueues every kernel of a graph-free pass — the SDF net with its input gradient, the deformer's
offset MLP with a vector-Jacobian product to its input, the fused skinning of ray points — instead of ~30 Python
level launches each.  These are the passes the surface root finder repeats up to 20 times per iteration and
garment (utils/FindSurfacePs.py:273-353 of the reference) and the no-grad grid queries of Seg3dLossless.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L

# Passes of MLP_ROWS_MIN..MLP_ROWS_MAX rows run as ONE launch of the row-tile kernels (csrc/mlp_rows.hip).  A workgroup owns 16 rows,
# so up to 4 096 rows are one round of workgroups on the 256 CUs and a pass takes the same ~350 us (SDF value + input gradient)
# whatever the row count, where the per-layer chain takes 278 us at 2 048 rows, 362 at 3 072 and 449 at 4 096
# (profiles/r04_mlp_rows_bench_v5.txt): the window below is where one launch is also the faster one.  A layer's products run at 77 %
# of the matrix pipe; the epilogue and the barrier between two layers (~4 us, pipe idle: one workgroup per CU at these row counts)
# bring a pass to 63 % (profiles/r04_mlp_rows_clock.txt).  RECMV_MLP_ROWS=0 keeps the per-layer chains everywhere;
# RECMV_MLP_ROWS_MIN / _MAX move the window (tools/ab_interleaved.py rows).
MLP_ROWS_MIN = int(os.environ.get("RECMV_MLP_ROWS_MIN", "3328"))
MLP_ROWS_MAX = int(os.environ.get("RECMV_MLP_ROWS_MAX", "4096"))
if os.environ.get("RECMV_MLP_ROWS_RT"):        # rows per workgroup / 16 (recmv_set_mlp_rows_tile): 1, 2, or 0 = by row count
    L.check(L.lib().recmv_set_mlp_rows_tile(int(os.environ["RECMV_MLP_ROWS_RT"])), "set_mlp_rows_tile")


def _rows_enabled():
    return os.environ.get("RECMV_MLP_ROWS", "1") != "0"


class MlpChain:
    """A recmv_mlp descriptor over tensors that the caller keeps alive (weights are referenced, not copied)."""

    def __init__(self, weights, biases, weights_t, dims, rows, multires, cond_dim=0, skip_layer=-1,
                 hidden_act=L.ACT_RELU, act_param=0.0, residual=False, pe_weights=None, second=None):
        """`second` = (weights, biases, weights_t) of a second net of the same shape: a call with `split_row` evaluates rows
        [split_row, P) with it (the two garments' SDF nets over one block of rays — one launch per layer for both)."""
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
        if second is not None:
            W2, b2, Wt2 = second
            self._keep += (list(W2), list(b2), list(Wt2) if Wt2 is not None else None)
            for l in range(n):
                assert W2[l].is_contiguous() and W2[l].dtype == torch.float32 and W2[l].shape == weights[l].shape, l
                assert (b2[l] is None) == (biases[l] is None)
                m.W2[l] = W2[l].data_ptr()
                m.bias2[l] = b2[l].data_ptr() if b2[l] is not None else None
                if Wt2 is not None:
                    assert Wt2[l].is_contiguous() and Wt2[l].shape == (dims[l], rows[l])
                    m.Wt2[l] = Wt2[l].data_ptr()
        self.has_second = second is not None
        self.m = m
        self.n_layers = n
        self.rows_last = rows[-1]
        self._ws = {}
        # row-tile form (csrc/mlp_rows.hip): the packed weights are built on first use, once per descriptor = per weight version
        self._rows_ok = second is None and bool(L.lib().recmv_mlp_rows_supported(C.byref(m)))
        self._packed = None
        self._packed_ev = None
        self._packed_seen = set()
        self._rows_ws = {}
        self._rows_last = {}          # slot -> did the last forward(keep=True) take the row-tile path?

    def _use_rows(self, P, n_out, split_row):
        return (self._rows_ok and max(MLP_ROWS_MIN, 1) <= P <= MLP_ROWS_MAX and n_out <= 16 and (split_row is None or not self.has_second)
                and _rows_enabled())

    def _pack(self, dev):
        """The packed weights, valid on the current stream of `dev` (packed on the first caller's stream; other streams wait for
        that once)."""
        st = torch.cuda.current_stream(dev)
        if self._packed is None:
            n = int(L.lib().recmv_mlp_pack_bytes(C.byref(self.m)))
            self._packed = L.scratch(n, torch.uint8, dev)
            with L.device_guard(dev):
                L.check(L.lib().recmv_mlp_pack(C.byref(self.m), L.ptr(self._packed), n, L.stream_ptr(dev)), "mlp_pack")
            self._packed_ev = torch.cuda.Event()
            self._packed_ev.record(st)
            self._packed_seen.add(st.cuda_stream)
        elif st.cuda_stream not in self._packed_seen:
            st.wait_event(self._packed_ev)
            self._packed_seen.add(st.cuda_stream)
        return self._packed

    def _rows_workspace(self, P, slot):
        need = int(L.lib().recmv_mlp_rows_workspace_bytes(C.byref(self.m), P))
        key = (slot, L.raw_stream(self.device))
        ws = self._rows_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = L.scratch(max(need, 256), torch.uint8, self.device)
            self._rows_ws[key] = ws
        return ws

    def _workspace(self, P, keep, slot=None):
        """Activation workspace; `slot` separates concurrent users of one chain (e.g. two garments evaluated on two
        streams through the same deformer MLP)."""
        need = int(L.lib().recmv_mlp_workspace_bytes(C.byref(self.m), P, keep))
        key = (slot, L.raw_stream(self.device))       # a chain object is shared by the streams of an iteration: no shared scratch
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = L.scratch(max(need, 256), torch.uint8, self.device)
            self._ws[key] = ws
        return ws

    def _split(self, split_row, P):
        if split_row is None or not self.has_second or split_row >= P:
            self.m.split_row = 0
        else:
            assert split_row > 0 and split_row % 128 == 0, "the second net starts at a multiple of 128 rows"
            self.m.split_row = int(split_row)

    def forward(self, x, cond=None, cond_index=None, n_out=None, keep=False, out=None, slot=None, split_row=None):
        """x [P,3] -> [P, n_out] (the first n_out outputs of the last layer); rows >= split_row through the second net."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 3 and x.is_contiguous()
        P = x.shape[0]
        self._split(split_row, P)
        n_out = self.rows_last if n_out is None else n_out
        if out is None:
            out = L.scratch((P, n_out), torch.float32, x.device)
        ld_cond = 0
        if cond is not None:
            assert cond.dtype == torch.float32 and cond.stride(-1) == 1 and cond.dim() == 2
            ld_cond = cond.stride(0)
            if cond_index is not None:
                assert cond_index.dtype == torch.int64 and cond_index.is_contiguous() and cond_index.numel() == P
        rows = self._use_rows(P, n_out, split_row)
        if keep:
            # (keyed like the workspaces: a forward(keep=True) on one stream says nothing about another stream's buffers)
            self._rows_last[(slot, L.raw_stream(self.device))] = rows
        if rows:
            packed = self._pack(x.device)
            ws = self._rows_workspace(P, slot) if keep else None
            with L.device_guard(x.device):
                L.check(L.lib().recmv_mlp_rows_forward(C.byref(self.m), L.ptr(packed), L.ptr(x), L.ptr(cond), ld_cond,
                                                       L.ptr(cond_index), P, n_out, L.ptr(out), out.stride(0) if P > 1 else n_out,
                                                       L.ptr(ws), ws.numel() if ws is not None else 0, int(keep),
                                                       L.stream_ptr(x.device)), "mlp_rows_forward")
            return out
        ws = self._workspace(P, int(keep), slot)
        with L.device_guard(x.device):
            L.check(L.lib().recmv_mlp_forward(C.byref(self.m), L.ptr(x), L.ptr(cond), ld_cond, L.ptr(cond_index), P,
                                              n_out, L.ptr(out), out.stride(0) if P > 1 else n_out, L.ptr(ws), ws.numel(),
                                              int(keep), L.stream_ptr(x.device)), "mlp_forward")
        return out

    def vjp_input(self, x, g_out=None, n_out=None, slot=None, split_row=None):
        """J(x)^T g_out -> [P,3]; call after forward(keep=True) with the same x (and split_row).  g_out None = ones (n_out 1)."""
        P = x.shape[0]
        self._split(split_row, P)
        n_out = (1 if g_out is None else g_out.shape[1]) if n_out is None else n_out
        gx = L.scratch((P, 3), torch.float32, x.device)
        ldg = 0
        if g_out is not None:
            assert g_out.dtype == torch.float32 and g_out.stride(1) == 1 and g_out.shape == (P, n_out)
            ldg = g_out.stride(0) if P > 1 else n_out
        key = (slot, L.raw_stream(self.device))
        if key not in self._rows_last:
            raise RuntimeError("MlpChain.vjp_input: no forward(keep=True) ran for slot %r on this stream — the kept activations "
                               "live in a per-(slot, stream) workspace" % (slot,))
        if self._rows_last[key]:                      # the activations are where the row-tile forward left them
            packed = self._pack(x.device)
            ws = self._rows_ws.get(key)
            if ws is None:
                raise RuntimeError("MlpChain.vjp_input: the row-tile workspace of slot %r on this stream is gone" % (slot,))
            with L.device_guard(x.device):
                L.check(L.lib().recmv_mlp_rows_vjp_input(C.byref(self.m), L.ptr(packed), L.ptr(x), P, n_out, L.ptr(g_out), ldg,
                                                         L.ptr(gx), L.ptr(ws), ws.numel(), L.stream_ptr(x.device)),
                        "mlp_rows_vjp_input")
            return gx
        ws = self._workspace(P, 1, slot)
        with L.device_guard(x.device):
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
    d = L.scratch((P, 3), torch.float32, dev)
    loss2 = angle = g_d = None
    if rays is not None:
        loss2 = L.scratch(P, torch.float32, dev)
        angle = L.scratch(P, torch.float32, dev)
        g_d = L.scratch((P, 3), torch.float32, dev)
    with L.device_guard(dev):
        L.check(L.lib().recmv_lbs_forward(L.ptr(ps), L.ptr(frame), P, L.ptr(A), L.ptr(trans), B, C.byref(grid),
                                          L.ptr(cam), L.ptr(rays), L.ptr(d), L.ptr(loss2), L.ptr(angle), L.ptr(g_d),
                                          L.stream_ptr(dev)), "lbs_forward")
    return d, loss2, angle, g_d


def lbs_vjp_input(ps, frame, A, grid, g_d):
    P, B = ps.shape[0], A.shape[0]
    g_p = L.scratch((P, 3), torch.float32, ps.device)
    with L.device_guard(ps.device):
        L.check(L.lib().recmv_lbs_vjp_input(L.ptr(ps), L.ptr(frame), P, L.ptr(A), B, C.byref(grid), L.ptr(g_d),
                                            L.ptr(g_p), L.stream_ptr(ps.device)), "lbs_vjp_input")
    return g_p


def rootfind_update(p, f, gf, loss2, angle, gd, unfinished, counter, dthreshold, athreshold, w1, w2, do_update):
    with L.device_guard(p.device):
        L.check(L.lib().recmv_rootfind_update(L.ptr(p), L.ptr(f), L.ptr(gf), L.ptr(loss2), L.ptr(angle), L.ptr(gd),
                                              L.ptr(unfinished), L.ptr(counter), p.shape[0], float(dthreshold),
                                              float(athreshold), float(w1), float(w2), int(do_update),
                                              L.stream_ptr(p.device)), "rootfind_update")


def rootfind_step(p, f, gf, loss2, angle, gd, unfinished, counters, marks, state, dthreshold, athreshold, w1, w2, times):
    """recmv_rootfind_step: the update with its step index on the device (every step is the same launch)."""
    with L.device_guard(p.device):
        L.check(L.lib().recmv_rootfind_step(L.ptr(p), L.ptr(f), L.ptr(gf), L.ptr(loss2), L.ptr(angle), L.ptr(gd),
                                            L.ptr(unfinished), L.ptr(counters), L.ptr(marks), L.ptr(state), p.shape[0],
                                            float(dthreshold), float(athreshold), float(w1), float(w2), int(times),
                                            L.stream_ptr(p.device)), "rootfind_step")


# --------------------------------------------------------------------------------------------------
# MLP jet: value + input Jacobian forward, explicit first-order reverse (csrc/mlp_jet.hip)
# --------------------------------------------------------------------------------------------------
_eye3 = {}


def _eye(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _eye3.get(key)
    if t is None:
        t = torch.eye(3, dtype=torch.float32, device=device).contiguous()
        torch.cuda.current_stream(device).synchronize()      # built once, read from every stream afterwards
        _eye3[key] = t
    return t


class MlpJet(torch.autograd.Function):
    """(y [P, rows_last], tang [3P, n_j]) = jet of an MLP at x; tang[k*P + p, j] = d mlp_j / d x_k.

    forward : one C call (recmv_mlp_jet_forward); the activations stay in a per-call workspace tensor.
    backward: one C call (recmv_mlp_jet_backward) -> gradients of x, the per-frame codes, every weight and bias.
    Differentiable ONCE: the loss only needs first-order gradients of (y, tang) — the second-order terms of the
    reference's autograd formulation are what the tangent rows carry."""

    @staticmethod
    def forward(ctx, cfg, x, cond, *wb):
        n = cfg['n_layers']
        Ws, bs = list(wb[:n]), list(wb[n:])
        dev = x.device
        xd = x.detach().contiguous()
        P = xd.shape[0]
        Wd = [W.detach().contiguous() for W in Ws]
        bd = [b.detach().contiguous() if b is not None else None for b in bs]
        from .ops import transposed
        Wt = [transposed(W) for W in Ws]           # cached on the parameter objects while their data is unchanged
        ctx.set_materialize_grads(False)
        ch = MlpChain(Wd, bd, Wt, cfg['dims'], [W.shape[0] for W in Wd], cfg['multires'], cond_dim=cfg['cond_dim'],
                      skip_layer=cfg['skip_layer'], hidden_act=cfg['hidden_act'], act_param=cfg['act_param'],
                      residual=cfg['residual'], pe_weights=cfg['pe_weights'])
        n_j = cfg['n_j']
        n_out = Wd[-1].shape[0]
        lib = L.lib()
        ws = L.scratch(int(lib.recmv_mlp_jet_workspace_bytes(C.byref(ch.m), P)), torch.uint8, dev)
        y = L.scratch((P, n_out), torch.float32, dev)
        tang = L.scratch((3 * P, n_j), torch.float32, dev)
        cond_d = cond.detach() if cond is not None else None
        cidx = cfg['cond_index']
        ld_cond = cond_d.stride(0) if cond_d is not None else 0
        if cond_d is not None:
            assert cond_d.dim() == 2 and cond_d.stride(1) == 1 and cond_d.dtype == torch.float32
        eye = _eye(dev)
        with L.device_guard(dev):
            L.check(lib.recmv_mlp_jet_forward(C.byref(ch.m), L.ptr(xd), L.ptr(cond_d), ld_cond, L.ptr(cidx), L.ptr(eye),
                                              P, n_j, L.ptr(y), n_out, L.ptr(tang), L.ptr(ws), ws.numel(),
                                              L.stream_ptr(dev)), "mlp_jet_forward")
        ctx.cfg, ctx.chain, ctx.ws, ctx.xd = cfg, ch, ws, xd
        ctx.cond_shape = None if cond is None else tuple(cond.shape)
        ctx.n = n
        ctx.has_bias = [b is not None for b in bs]
        return y, tang

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy, gtang):
        cfg, ch, ws, xd, n = ctx.cfg, ctx.chain, ctx.ws, ctx.xd, ctx.n
        dev = xd.device
        P = xd.shape[0]
        n_j = cfg['n_j']
        lib = L.lib()
        need = ctx.needs_input_grad            # (cfg, x, cond, W..., b...)
        gy = gy.contiguous() if gy is not None else None
        gtang = gtang.contiguous() if gtang is not None else None
        Wd = ch._keep[0]
        gWs = [L.scratch_like(Wd[l]) if need[3 + l] else None for l in range(n)]
        gbs = [L.scratch(Wd[l].shape[0], torch.float32, dev)
               if (ctx.has_bias[l] and need[3 + n + l]) else None for l in range(n)]
        gW_arr = (C.c_void_p * n)(*[g.data_ptr() if g is not None else None for g in gWs])
        gb_arr = (C.c_void_p * n)(*[g.data_ptr() if g is not None else None for g in gbs])
        want_cond = ctx.cond_shape is not None and need[2]
        ld_in = (cfg['dims'][0] + 3) // 4 * 4
        g_in = L.scratch((4 * P, ld_in), torch.float32, dev) if want_cond else None
        gx = L.scratch((P, 3), torch.float32, dev) if need[1] else None
        eye = _eye(dev)
        with L.device_guard(dev):
            L.check(lib.recmv_mlp_jet_backward(C.byref(ch.m), L.ptr(xd), L.ptr(eye), P, n_j, L.ptr(gy),
                                               gy.stride(0) if gy is not None else 0, L.ptr(gtang),
                                               C.cast(gW_arr, C.c_void_p), C.cast(gb_arr, C.c_void_p), L.ptr(g_in),
                                               L.ptr(gx), L.ptr(ws), ws.numel(), L.stream_ptr(dev)), "mlp_jet_backward")
        gcond = None
        if want_cond:
            d_pe = 3 + 6 * cfg['multires']
            gc = g_in[:P, d_pe:d_pe + cfg['cond_dim']]
            cidx = cfg['cond_index']
            if cidx is None:
                gcond = gc.sum(0, keepdim=True).expand(ctx.cond_shape) if ctx.cond_shape[0] == 1 else None
                assert gcond is not None, "a per-point frame index is needed when there are several codes"
            elif cfg.get('cond_blocks'):
                # frame-major blocks of equal length: a fixed-order sum per frame instead of atomics
                nb = cfg['cond_blocks']
                gcond = gc.reshape(nb, P // nb, gc.shape[1]).sum(1)
            else:
                from .ops import rows_sum_by_index
                gcond = rows_sum_by_index(gc, cidx, ctx.cond_shape[0])       # fixed order, no float atomics
        ctx.ws = None
        return (None, gx, gcond) + tuple(gWs) + tuple(gbs)


def mlp_jet(x, cond, cond_index, Ws, bs, dims, multires, pe_weights, cond_dim, skip_layer, hidden_act, act_param,
            residual, n_j, cond_blocks=0):
    """(y [P, rows_last], J [P, n_j, 3]) with J[p, j, k] = d mlp_j / d x_k (the residual's identity NOT included)."""
    cfg = dict(n_layers=len(Ws), dims=list(dims), multires=multires,
               pe_weights=None if pe_weights is None else tuple(float(w) for w in pe_weights), cond_dim=cond_dim,
               skip_layer=skip_layer, hidden_act=hidden_act, act_param=act_param, residual=residual, n_j=n_j,
               cond_index=cond_index, cond_blocks=cond_blocks)
    y, tang = MlpJet.apply(cfg, x, cond, *Ws, *bs)
    P = x.shape[0]
    return y, tang.view(3, P, n_j).permute(1, 2, 0)


# --------------------------------------------------------------------------------------------------
# Fused linear-blend skinning with first-order backward (csrc/lbs_fused.hip)
# --------------------------------------------------------------------------------------------------
def lbs_vjp_params(ps, frame, A_shape, grid, g_d):
    """(gA [B,24,4,4], gtrans [B,3]) — staged kernel + MFMA gemm_tn + fixed-order column sum."""
    from . import ops
    P, B = ps.shape[0], A_shape[0]
    dev = ps.device
    W = L.scratch((P, 24), torch.float32, dev)
    Q = L.scratch((P, B * 12), torch.float32, dev)
    Gs = L.scratch((P, B * 3), torch.float32, dev)
    lib = L.lib()
    with L.device_guard(dev):
        L.check(lib.recmv_lbs_vjp_params_stage(L.ptr(ps), L.ptr(frame), P, B, C.byref(grid), L.ptr(g_d), L.ptr(W),
                                               L.ptr(Q), L.ptr(Gs), L.stream_ptr(dev)), "lbs_vjp_params_stage")
        gAm = ops.gemm_tn(W, Q)                                            # [24, B*12]
        need = int(lib.recmv_colsum_workspace_bytes(P, B * 3))
        ws = L.scratch(max(need, 256), torch.uint8, dev)
        gt = L.scratch(B * 3, torch.float32, dev)
        L.check(lib.recmv_colsum(L.ptr(Gs), B * 3, P, B * 3, L.ptr(gt), L.ptr(ws), ws.numel(), L.stream_ptr(dev)),
                "colsum")
    gA = torch.zeros(A_shape, dtype=torch.float32, device=dev)
    gA[:, :, :3, :] = gAm.view(24, B, 3, 4).permute(1, 0, 2, 3)
    return gA, gt.view(B, 3)


class LbsJet(torch.autograd.Function):
    """(v [P,3], J [P,3,3]) = the skinning stage and its Jacobian d v / d p at `ps` in ONE launch (recmv_lbs_jet_forward) instead of
    the sampler -> blend -> transform composition plus three create_graph autograd.grad calls through it (utils/utils.py:133-156 of
    the reference).  backward: one launch for the per-point part (recmv_lbs_jet_backward_stage: the gradient wrt ps, with the mixed
    second derivatives of the trilinear weights) + the two fixed-order reductions of the parameter side (gA, gtrans).
    Differentiable ONCE: the loss needs first-order gradients of (v, J) only."""

    @staticmethod
    def forward(ctx, ps, A, trans, frame, grid):
        psd = ps.detach().contiguous()
        Ad = A.detach().contiguous()
        td = trans.detach().contiguous()
        P, B = psd.shape[0], Ad.shape[0]
        dev = psd.device
        v = L.scratch((P, 3), torch.float32, dev)
        J = L.scratch((P, 3, 3), torch.float32, dev)
        with L.device_guard(dev):
            L.check(L.lib().recmv_lbs_jet_forward(L.ptr(psd), L.ptr(frame), P, L.ptr(Ad), L.ptr(td), B, C.byref(grid), L.ptr(v),
                                                  L.ptr(J), L.stream_ptr(dev)), "lbs_jet_forward")
        ctx.save_for_backward(psd, Ad)
        ctx.frame, ctx.grid, ctx.A_shape = frame, grid, tuple(A.shape)
        ctx.set_materialize_grads(False)
        return v, J

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gv, gJ):
        from . import ops
        psd, Ad = ctx.saved_tensors
        need = ctx.needs_input_grad
        if gv is None and gJ is None:
            return None, None, None, None, None
        P, B = psd.shape[0], Ad.shape[0]
        dev = psd.device
        gv = gv.contiguous() if gv is not None else None
        gJ = gJ.contiguous() if gJ is not None else None
        g_p = L.scratch((P, 3), torch.float32, dev)
        W4 = L.scratch((4 * P, 24), torch.float32, dev)
        Q4 = L.scratch((4 * P, B * 12), torch.float32, dev)
        Gs = L.scratch((P, B * 3), torch.float32, dev)
        lib = L.lib()
        gA = gt = None
        with L.device_guard(dev):
            L.check(lib.recmv_lbs_jet_backward_stage(L.ptr(psd), L.ptr(ctx.frame), P, L.ptr(Ad), B, C.byref(ctx.grid), L.ptr(gv),
                                                     L.ptr(gJ), L.ptr(g_p), L.ptr(W4), L.ptr(Q4), L.ptr(Gs), L.stream_ptr(dev)),
                    "lbs_jet_backward_stage")
            if need[1] and P > 0:
                gAm = ops.gemm_tn(W4, Q4)                                          # [24, B*12]
                gA = torch.zeros(ctx.A_shape, dtype=torch.float32, device=dev)
                gA[:, :, :3, :] = gAm.view(24, B, 3, 4).permute(1, 0, 2, 3)
            if need[2] and P > 0 and gv is not None:
                ws = L.scratch(max(int(lib.recmv_colsum_workspace_bytes(P, B * 3)), 256), torch.uint8, dev)
                gt = L.scratch(B * 3, torch.float32, dev)
                L.check(lib.recmv_colsum(L.ptr(Gs), B * 3, P, B * 3, L.ptr(gt), L.ptr(ws), ws.numel(), L.stream_ptr(dev)), "colsum")
                gt = gt.view(B, 3)
        return (g_p if need[0] else None), gA, gt, None, None


class LbsFused(torch.autograd.Function):
    """d = (sum_j w_j(p) A[frame,j]) [p;1] + trans[frame] in one kernel; backward = the input VJP kernel plus the staged
    parameter VJP.  When a graph is being built in backward (create_graph=True — the Jacobian terms of the loss), it
    falls back to differentiating `classic`, the composition of differentiable ops (sampler with double backward)."""

    @staticmethod
    def forward(ctx, ps, A, trans, frame, grid, classic):
        psd = ps.detach().contiguous()
        Ad = A.detach().contiguous()
        td = trans.detach().contiguous()
        d = lbs_forward(psd, frame, Ad, td, grid)[0]
        ctx.save_for_backward(ps, A, trans)
        ctx.frame, ctx.grid, ctx.classic = frame, grid, classic
        return d

    @staticmethod
    def backward(ctx, gd):
        ps, A, trans = ctx.saved_tensors
        frame, grid = ctx.frame, ctx.grid
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            with torch.enable_grad():
                v = ctx.classic(ps, A, trans, frame)
                ins = [t for t, n in zip((ps, A, trans), need[:3]) if n]
                gs = list(torch.autograd.grad(v, ins, gd, create_graph=True, allow_unused=True))
            out = [gs.pop(0) if n else None for n in need[:3]]
            return out[0], out[1], out[2], None, None, None
        gd = gd.contiguous()
        psd = ps.detach().contiguous()
        g_ps = lbs_vjp_input(psd, frame, A.detach().contiguous(), grid, gd) if need[0] else None
        gA = gt = None
        if need[1] or need[2]:
            gA, gt = lbs_vjp_params(psd, frame, tuple(A.shape), grid, gd)
        return g_ps, (gA if need[1] else None), (gt if need[2] else None), None, None, None
