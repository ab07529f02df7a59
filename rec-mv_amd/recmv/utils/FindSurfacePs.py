"""Surface-point finder and ray/surface root finder (utils/FindSurfacePs.py of the reference).

  FindSurfacePs             :7-60     pixels whose first valid face has all barycentrics > 0
  OptimizeSurfacePs         :145-207  the same iteration for one SDF net and any deformer (base-class loop)
  OptimizeGarmentSurfaceSinlge :210-272  ... for one garment net with its offset slot (visualisation / evaluation paths)
  OptimizeGarmentSurfacePs  :273-353  find canonical p with |SDF(p)| < dthreshold whose deformed image lies on
                                      the pixel's ray (angle < athreshold degrees), <= `times` steps of
                                      p <- p - E * gradE / |gradE|^2,  E = w1*|f(p)| + w2*|(d-c) x v| / |d-c|

The SDF / deformer evaluations inside run on the recmv kernels.  The active set is kept as an index list
that shrinks on device; the loop exits early through a single count read per step (the reference does the
same sync through `curPs.shape[0]==0` after boolean indexing, :311-313).
"""
import os

import numpy as np
import torch

__all__ = ["FindSurfacePs", "OptimizeGarmentSurfacePs", "OptimizeGarmentSurfaceSinlge", "OptimizeSurfacePs", "prepare_root_finder"]


def FindSurfacePs(TmpVs, TmpFaces, frags):
    """frags: object with pix_to_face [N,H,W,K] (int64, -1 = empty) and bary_coords [N,H,W,K,3].
    Returns (batch_inds, row_inds, col_inds, initTmpPs, finds) in `nonzero` (row-major) order."""
    N, H, W, K = frags.pix_to_face.shape
    pix_to_face = frags.pix_to_face
    bary_coords = frags.bary_coords
    if K == 1:
        # one face per pixel (the reference's raster setting): the first inner fragment IS fragment 0, so the
        # scatter-min / gathers below reduce to plain indexing — same outputs, a third of the device passes
        innerCheck = (bary_coords[:, :, :, 0] > 0.0).all(-1) & (pix_to_face[:, :, :, 0] >= 0)
        batch_inds, row_inds, col_inds = innerCheck.nonzero(as_tuple=True)
        finds = pix_to_face[batch_inds, row_inds, col_inds, 0] % TmpFaces.shape[0]
        ws = bary_coords[batch_inds, row_inds, col_inds, 0]
        initTmpPs = (TmpVs[TmpFaces[finds].view(-1)].view(-1, 3, 3) * ws[:, :, None]).sum(1)
        return batch_inds, row_inds, col_inds, initTmpPs, finds
    innerCheck = (bary_coords > 0.0).all(-1) * (pix_to_face >= 0)
    rows, cols = innerCheck.view(-1, K).nonzero(as_tuple=True)
    index = torch.ones(N * H * W, dtype=torch.long, device=rows.device) * K
    # torch_scatter.scatter(cols, rows, reduce='min', out=index)  (:30)
    index = index.scatter_reduce(0, rows, cols, reduce='amin', include_self=True).view(N, H, W)
    innerCheck = innerCheck.any(dim=-1)
    batch_inds, row_inds, col_inds = innerCheck.nonzero(as_tuple=True)
    finds = torch.gather(pix_to_face[innerCheck], 1, index[innerCheck].view(-1, 1)).view(-1)
    finds = finds % TmpFaces.shape[0]
    ws = torch.gather(bary_coords[innerCheck], 1, index[innerCheck].view(-1, 1, 1).expand(-1, 1, 3)).view(-1, 3)
    initTmpPs = (TmpVs[TmpFaces[finds].view(-1)].view(-1, 3, 3) * ws[:, :, None]).sum(1)
    return batch_inds, row_inds, col_inds, initTmpPs, finds


def _ray_angle_deg(direct, rays):
    up = torch.linalg.cross(direct, rays, dim=1)
    return torch.arcsin(up.norm(dim=1) / direct.norm(dim=1)) * 180. / np.pi


_side_streams = {}


# The unfinished-ray marks reach the host after the first two steps (the compaction reads the second) and then every 4th step: each
# copy is a blit launch + a completion signal on the garment's stream, and an early exit seen up to three steps late costs three steps
# that change nothing.  Measured on the MI355X, bench scene: 98.5-99.0 ms per iteration against 99.4 with a copy per step.
_MARK_EVERY = max(1, int(os.environ.get('RECMV_ROOT_MARK_EVERY', '4')))


def _streams(device, n):
    key = device.index if device.index is not None else torch.cuda.current_device()
    pool = _side_streams.setdefault(key, [])
    while len(pool) < n:
        pool.append(_make_stream(device))
    return pool[:n]


class _RootState:
    """One garment's root-finding iteration on the graph-free passes: per step two C calls for the SDF net (value +
    input gradient), four for the deformer (offset MLP, fused skinning + ray energy, and their VJPs) and one fused
    stopping-test / update kernel — about 55 launches on a few thousand rays.  All rays are carried through every step
    (rows of the kernels are independent, so the active rays get the same updates); finished rays are simply not
    updated.  The post-update check of step i is the forward pass of step i+1, evaluated once.

    Every step is the SAME sequence of launches on the same buffers (the step index lives on the device,
    recmv_rootfind_step).  With `RECMV_ROOT_GRAPH=1` the first step runs eagerly, the second is captured into a hipGraph
    and the remaining ones replay it.  That cuts the host's share (capture 0.3 ms per garment, replay 32 us per step
    against ~360 us of launches) but not the step's time on the device — 21 steps take 21 ms on a garment's stream either
    way (1 ms per step: ~26 products of 23-27 us at 3 k rows), 39 ms beside the mask loss — and the iteration as a whole
    measured slower with it (149.9 vs 131.4 ms), so it is off by default; results are bit-identical either way.
    The early exit polls a pinned copy of the per-step unfinished-ray marks and never blocks the host; the extra steps
    this can cost change nothing (no unfinished ray = no update)."""

    _pools = {}      # one graph memory pool per stream: the garments' graphs replay concurrently and must not share blocks

    def __init__(self, cam_pos, rays, initTmpPs, batch_inds, tmpSdf, ratio, deformer, defconds, smpl_conds, name,
                 dthreshold, athreshold, w1, w2, times, stream):
        dev = initTmpPs.device
        self.stream = stream
        self.args = (dthreshold, athreshold, w1, w2)
        self.times, self.it, self.finished = times, 0, False
        self.tmpSdf, self.deformer, self.ratio, self.name = tmpSdf, deformer, ratio, name
        self.conds = [defconds, smpl_conds]
        self.graph = None
        self.use_graph = os.environ.get('RECMV_ROOT_GRAPH', '0') == '1'
        with torch.cuda.stream(stream):
            self.p = initTmpPs.detach().clone().contiguous()
            P = self.p.shape[0]
            self.cam = cam_pos.detach().reshape(3).contiguous().float()
            self.rays = rays.detach().contiguous()
            self.frame = batch_inds.contiguous()
            self.unfinished = torch.ones(P, dtype=torch.uint8, device=dev)
            # [counters | marks | state]: unfinished rays per step, the same + 1 once the step has run, the step index
            self.ints = torch.zeros(2 * (times + 2) + 1, dtype=torch.int32, device=dev)
        self.counters = self.ints[:times + 2]
        self.marks = self.ints[times + 2:2 * (times + 2)]
        self.state = self.ints[2 * (times + 2):]
        self.host = torch.zeros(times + 2, dtype=torch.int32).pin_memory()
        self.host_np = self.host.numpy()          # the same pinned words, read without creating tensors
        self.sdf_chain = tmpSdf.chain(tmpSdf._pe_weights(ratio), need_t=True)
        assert tmpSdf.d_out == 1
        self.live = P                 # rows the steps run over: all of them until the compaction (see _compact)
        self.perm = None
        self.mark_ev = None
        if P == 0:
            self.finished = True

    def _enqueue(self):
        """One step on the current stream — identical every time it is called."""
        from .. import chains
        dthr, athr, w1, w2 = self.args
        n = self.live
        p, frame, rays, unfinished = self.p[:n], self.frame[:n], self.rays[:n], self.unfinished[:n]
        f = self.sdf_chain.forward(p, n_out=1, keep=True)
        gf = self.sdf_chain.vjp_input(p, None)
        _, loss2, angle, gd = self.deformer.ray_energy_and_vjp(p, self.conds, frame, self.cam, rays,
                                                               ratio=self.ratio, offset_type=self.name)
        chains.rootfind_step(p, f, gf, loss2, angle, gd, unfinished, self.counters, self.marks, self.state, dthr,
                             athr, w1, w2, self.times)
        if self.it < 2 or self.it % _MARK_EVERY == 0:
            self.host.copy_(self.marks, non_blocking=True)
        if self.it == 1:
            self.mark_ev = torch.cuda.Event()
            self.mark_ev.record()

    # The reference shrinks the active ray set every step (utils/FindSurfacePs.py:300-303); carrying every ray through all 21 steps
    # costs the rows of the rays that are done — on the settled bench scene 17 % of them after the first update, then ~1 % more per
    # step (profiles/r03_rootfind_unfinished_per_step.txt).  ONE compaction after the first update takes most of that: the host waits
    # for that step's unfinished-ray count (a few ms into a phase in which it has slack), the unfinished rays move to the front
    # (stable), and the remaining steps run over exactly those rows.  Rows are independent and the 64 x 32 product kernel serves
    # every row count these launches have, so every ray gets the same bits as without the compaction; result() undoes the order.
    COMPACT_MIN_ROWS = 1024
    COMPACT_MAX_KEEP = 0.93

    def _compact(self):
        P = self.p.shape[0]
        if (self.perm is not None or self.use_graph or P < self.COMPACT_MIN_ROWS or self.mark_ev is None
                or os.environ.get('RECMV_ROOT_COMPACT', '1') == '0'):
            return
        self.mark_ev.synchronize()
        n = int(self.host_np[1]) - 1
        if n <= 0 or n > self.COMPACT_MAX_KEEP * P:
            return
        with torch.cuda.stream(self.stream):
            un = self.unfinished != 0
            order = torch.cat([torch.nonzero_static(un, size=n).view(-1), torch.nonzero_static(~un, size=P - n).view(-1)])
            self.p = self.p.index_select(0, order)
            self.rays = self.rays.index_select(0, order)
            self.frame = self.frame.index_select(0, order)
            self.unfinished = self.unfinished.index_select(0, order)
        self.perm, self.live = order, n

    def step(self):
        """Enqueue one iteration on this garment's stream; returns False once the iteration is over."""
        if self.finished:
            return False
        it = self.it
        if it >= 1 and bool((self.host_np[:it] == 1).any()):
            # early exit, never waited for: a step whose mark has already arrived and says "no unfinished ray"; the
            # host keeps queueing otherwise (it has other streams to feed); a step queued past the end changes nothing
            self.finished = True
            return False
        if it > self.times:
            self.finished = True
            return False
        with torch.cuda.stream(self.stream):
            if not self.use_graph or it == 0:
                self._enqueue()                       # (the first step also warms every cache the sequence touches)
            else:
                if self.graph is None:
                    keep = _RootState._pools.get(self.stream.cuda_stream)
                    if keep is None:
                        # the allocator drops a pool with its last graph: a one-node graph that is never replayed keeps
                        # this stream's pool (and the blocks the per-iteration graphs reuse) alive
                        pool = torch.cuda.graph_pool_handle()
                        k = torch.cuda.CUDAGraph()
                        k.capture_begin(pool=pool, capture_error_mode="thread_local")
                        try:
                            k_t = torch.zeros(1, device=self.p.device)
                        finally:
                            k.capture_end()
                        keep = _RootState._pools[self.stream.cuda_stream] = (pool, k, k_t)
                    pool = keep[0]
                    if self.stream == torch.cuda.default_stream(self.p.device):
                        raise RuntimeError("RECMV_ROOT_GRAPH=1: a root finder on the legacy default stream cannot be captured "
                                           "(RECMV_SERIAL=1 puts the first garment there); run it on a side stream")
                    g = torch.cuda.CUDAGraph()
                    g.capture_begin(pool=pool, capture_error_mode="thread_local")
                    try:
                        self._enqueue()
                    finally:
                        g.capture_end()
                    self.graph = g
                self.graph.replay()
        self.it += 1
        if self.it == 2:
            self._compact()
        return True

    def steps_trace(self):
        self.host.copy_(self.marks)               # (blocking: the trace is a diagnostic)
        return [int(m) - 1 for m in self.host_np[:self.it]]

    def result(self):
        if self.perm is None:
            return self.p, self.unfinished == 0
        p = torch.empty_like(self.p).index_copy_(0, self.perm, self.p)
        ok = torch.empty_like(self.unfinished).index_copy_(0, self.perm, self.unfinished) == 0
        return p, ok


class _RootGroup(_RootState):
    """The root-finding iteration of ALL garments as ONE block of rows on one stream.  The offset MLP and the skinner are shared by
    the garments (a ray's per-frame deformation code is a row of the concatenated code table), the update is per ray, and the SDF
    nets — the only per-garment weights — are evaluated by row-segmented products (rows [0, split) through the first garment's
    net, the rest through the second's: recmv_gemm_nt_seg, csrc/mlp_chain.hip), so a step is the SAME ~55 launches for two
    garments as for one, on twice the rows (64 x 64 tiles instead of 64 x 32 at ~3 k rays per garment).  The first garment's
    rows are padded to a multiple of 128 (the tile height) with finished copies of its first ray.  Per-ray arithmetic is what
    _RootState does: same points, same convergence flags; the iteration ends when no garment has an unfinished ray."""

    def __init__(self, cam_pos, rays_list, initTmpPs_list, batch_inds_list, nets, ratio, deformer, defconds_list, smpl_conds,
                 dthreshold, athreshold, w1, w2, times, stream):
        dev = initTmpPs_list[0].device
        self.stream = stream
        self.args = (dthreshold, athreshold, w1, w2)
        self.times, self.it, self.finished = times, 0, False
        self.deformer, self.ratio, self.name = deformer, ratio, 'rootfind'
        self.graph, self.use_graph = None, False
        self.perm = self.mark_ev = None                   # (no compaction in the one-block form)
        sizes = [int(p.shape[0]) for p in initTmpPs_list]
        live = [g for g, n in enumerate(sizes) if n > 0]
        assert len(initTmpPs_list) <= 2, 'only support less or equal than 2 garment_type'
        self.sizes, self.offsets = sizes, [0] * len(sizes)
        if len(live) == 2:
            self.offsets[1] = -(-sizes[0] // 128) * 128
        P = max((self.offsets[g] + sizes[g] for g in live), default=0)
        self.split = self.offsets[1] if len(live) == 2 else None
        n_frames = defconds_list[0].shape[0]
        with torch.cuda.stream(stream):
            self.p = torch.empty(P, 3, device=dev)
            self.rays = torch.empty(P, 3, device=dev)
            self.frame = torch.empty(P, dtype=torch.int64, device=dev)
            self.cond_index = torch.empty(P, dtype=torch.int64, device=dev)
            self.unfinished = torch.zeros(P, dtype=torch.uint8, device=dev)
            for g in live:
                o, n = self.offsets[g], sizes[g]
                end = self.offsets[g + 1] if g + 1 < len(sizes) and (g + 1) in live else o + n
                self.p[o:o + n] = initTmpPs_list[g].detach()
                self.rays[o:o + n] = rays_list[g].detach()
                self.frame[o:o + n] = batch_inds_list[g]
                self.unfinished[o:o + n] = 1
                if end > o + n:                               # padding: finished copies of the garment's first ray
                    self.p[o + n:end] = initTmpPs_list[g][:1].detach()
                    self.rays[o + n:end] = rays_list[g][:1].detach()
                    self.frame[o + n:end] = batch_inds_list[g][:1]
                self.cond_index[o:end] = self.frame[o:end] + g * n_frames
            self.cam = cam_pos.detach().reshape(3).contiguous().float()
            self.cond_table = torch.cat([c.detach() for c in defconds_list], dim=0).contiguous()
            self.ints = torch.zeros(2 * (times + 2) + 1, dtype=torch.int32, device=dev)
        self.conds = [self.cond_table, smpl_conds]
        self.counters = self.ints[:times + 2]
        self.marks = self.ints[times + 2:2 * (times + 2)]
        self.state = self.ints[2 * (times + 2):]
        self.host = torch.zeros(times + 2, dtype=torch.int32).pin_memory()
        self.host_np = self.host.numpy()
        if len(live) == 2:
            ws = nets[0]._pe_weights(ratio)
            self.sdf_chain = nets[0].pair_chain(nets[1], ws)
        elif live:
            net = nets[live[0]]
            self.sdf_chain = net.chain(net._pe_weights(ratio), need_t=True)
        if P == 0:
            self.finished = True

    def _enqueue(self):
        from .. import chains
        dthr, athr, w1, w2 = self.args
        p = self.p
        f = self.sdf_chain.forward(p, n_out=1, keep=True, split_row=self.split)
        gf = self.sdf_chain.vjp_input(p, None, split_row=self.split)
        _, loss2, angle, gd = self.deformer.ray_energy_and_vjp(p, self.conds, self.frame, self.cam, self.rays, ratio=self.ratio,
                                                               offset_type=self.name, cond_index=self.cond_index)
        chains.rootfind_step(p, f, gf, loss2, angle, gd, self.unfinished, self.counters, self.marks, self.state, dthr,
                             athr, w1, w2, self.times)
        if self.it < 2 or self.it % _MARK_EVERY == 0:
            self.host.copy_(self.marks, non_blocking=True)

    def results(self):
        outs, oks = [], []
        for g, n in enumerate(self.sizes):
            o = self.offsets[g]
            outs.append(self.p[o:o + n].clone())
            oks.append(self.unfinished[o:o + n] == 0)
        return outs, oks


@torch.no_grad()
def prepare_root_finder(tmpSdf_nets, deformer, smpl_conds, ratio):
    """Everything the garments' root finders share (weight-normed weights and their transposes, posed skeleton, chain
    descriptors), produced on the CURRENT stream.  A caller that wants the root finder to start before the rest of its
    queue has drained calls this early, records an event, and passes it as `after` to OptimizeGarmentSurfacePs."""
    with torch.no_grad():
        for net in tmpSdf_nets:
            net.chain(net._pe_weights(ratio), need_t=True)
        deformer.prepare_explicit([None, smpl_conds], ratio=ratio)


def _optimize_explicit_all(cam_pos, rays_list, initTmpPs_list, batch_inds_list, tmpSdf_nets, ratio, deformer,
                           defconds_list, smpl_conds, garment_names, dthreshold, athreshold, w1, w2, times, after=None):
    """All garments at once, one HIP stream per garment: a garment's step is a train of ~60 small kernels on a few
    thousand rays that cannot fill 256 CUs; the garments are independent, so their trains overlap."""
    dev = initTmpPs_list[0].device
    main = torch.cuda.current_stream(dev)
    # the first garment's chain runs on the caller's stream (the ray pipeline is one chain anyway: sampling -> root finder -> render
    # loss), only the others get side streams: HIP spreads streams over 4 hardware queues, and a fifth stream shares one — with one
    # side stream per garment the curve branch's stream sat behind a root finder's 38 ms of launches (tools/phase_overlap.py)
    streams = [main] + _streams(dev, len(initTmpPs_list) - 1)
    if os.environ.get('RECMV_ROOT_GRAPH', '0') == '1' and main == torch.cuda.default_stream(dev):
        streams = _streams(dev, len(initTmpPs_list))       # (a capture cannot start on the legacy default stream)
    # everything the garments share (weight-normed weights and their transposes, posed skeleton, chain descriptors) is
    # produced BEFORE the side streams fork: on the main stream right here, or — `after` given — earlier by the caller
    # (prepare_root_finder), in which case the side streams wait for the caller's events / streams only and start while
    # the main stream still works through whatever was queued after them
    if after is None:
        prepare_root_finder(tmpSdf_nets[:len(initTmpPs_list)], deformer, smpl_conds, ratio)
    if os.environ.get('RECMV_ROOT_GROUPED', '0') == '1' and len(initTmpPs_list) <= 2:
        # all garments as one block of rows, one launch per layer for all of them (_RootGroup).  OFF by default: measured on the
        # MI355X (tools/ab_rootfind.sh, profiles/r03_rootfind_grouped_ab.txt) the single chain of twice-as-large launches takes
        # 23.0 ms per iteration against 18.9 ms for the two garments' chains running beside each other on two streams (the chain
        # is bound by its ~1150 dependent launches of 15-25 us, which two streams overlap and one block does not), the iteration
        # 122-125 ms against 116-121 ms.
        st = streams[0]
        if after is None:
            st.wait_stream(main)
        else:
            for dep in after:
                st.wait_stream(dep) if isinstance(dep, torch.cuda.Stream) else st.wait_event(dep)
        group = _RootGroup(cam_pos, rays_list, initTmpPs_list, batch_inds_list, tmpSdf_nets, ratio, deformer, defconds_list,
                           smpl_conds, dthreshold, athreshold, w1, w2, times, st)
        while group.step():
            pass
        with torch.cuda.stream(st):
            outs, oks = group.results()
        main.wait_stream(st)
        for t in outs + oks:
            t.record_stream(main)
        if os.environ.get('RECMV_ROOT_TRACE'):
            torch.cuda.synchronize()
            print('rootfind (grouped) %d steps; unfinished per step:' % group.it, group.steps_trace(), flush=True)
        return outs, oks
    states = []
    for g, (initTmpPs, batch_inds, defconds, rays, name) in enumerate(
            zip(initTmpPs_list, batch_inds_list, defconds_list, rays_list, garment_names)):
        if streams[g] is main and after is None:
            pass
        elif after is None:
            streams[g].wait_stream(main)
        else:
            for dep in after:
                if isinstance(dep, torch.cuda.Stream):
                    streams[g].wait_stream(dep)
                else:
                    streams[g].wait_event(dep)
        states.append(_RootState(cam_pos, rays, initTmpPs, batch_inds, tmpSdf_nets[g], ratio, deformer, defconds,
                                 smpl_conds, name, dthreshold, athreshold, w1, w2, times, streams[g]))
    trace = bool(os.environ.get('RECMV_ROOT_TRACE'))
    if trace:
        for st in states:
            st.ev0 = torch.cuda.Event(enable_timing=True)
            st.ev0.record(st.stream)
    live = True
    while live:
        live = False
        for st in states:
            live = st.step() or live
    if trace:
        for st in states:
            st.ev1 = torch.cuda.Event(enable_timing=True)
            st.ev1.record(st.stream)
    outs, oks = [], []
    for st in states:
        if st.stream is not main:
            main.wait_stream(st.stream)
        with torch.cuda.stream(st.stream):
            p, ok = st.result()
        if st.stream is not main:
            main.wait_stream(st.stream)
            p.record_stream(main)
            ok.record_stream(main)
        if os.environ.get('RECMV_ROOT_TRACE'):
            torch.cuda.synchronize()
            print('rootfind %d steps on its stream: %.2f ms; unfinished per step:' % (st.it, st.ev0.elapsed_time(st.ev1)),
                  st.steps_trace(), flush=True)
        outs.append(p)
        oks.append(ok)
    return outs, oks


def OptimizeGarmentSurfacePs(cam_pos, rays_list, initTmpPs_list, batch_inds_list, tmpSdf_nets, ratio, deformer,
                             defconds_list, garment_names, dthreshold=5.e-5, athreshold=0.02, w1=3.05, w2=1.,
                             times=5, after=None):
    smpl_conds = defconds_list[1]
    if (len(initTmpPs_list) > 0 and all(t.is_cuda for t in initTmpPs_list) and hasattr(deformer, 'ray_energy_and_vjp')
            and all(hasattr(n, 'chain') for n in tmpSdf_nets)):
        outs, oks = _optimize_explicit_all(cam_pos, rays_list, initTmpPs_list, batch_inds_list, tmpSdf_nets, ratio,
                                           deformer, defconds_list[0], smpl_conds, garment_names, dthreshold,
                                           athreshold, w1, w2, times, after=after)
        return [o.detach() for o in outs], oks
    optimized_init_tmp_ps_list = []
    optimized_check_list = []
    for garment_idx, (initTmpPs, batch_inds, defconds, rays, garment_name) in enumerate(
            zip(initTmpPs_list, batch_inds_list, defconds_list[0], rays_list, garment_names)):
        pts, ok = _optimize_generic(cam_pos, rays, initTmpPs, batch_inds, tmpSdf_nets[garment_idx], ratio,
                                    lambda p, b, dc=defconds, n=garment_name: deformer(p, [dc, smpl_conds], b, ratio=ratio, offset_type=n),
                                    dthreshold, athreshold, w1, w2, times)
        optimized_init_tmp_ps_list.append(pts)
        optimized_check_list.append(ok)
    return optimized_init_tmp_ps_list, optimized_check_list


def _optimize_generic(cam_pos, rays, initTmpPs, batch_inds, tmpSdf, ratio, deform, dthreshold, athreshold, w1, w2, times):
    """The iteration of utils/FindSurfacePs.py:145-207 / :210-272 / :273-353 on autograd, for any deformer: `deform(points, frame
    indices)` -> deformed points.  `initTmpPs` is updated in place like the reference's (`initTmpPs[unfinished] = curPs`)."""
    with torch.no_grad():
        check1 = tmpSdf(initTmpPs, ratio).view(-1).abs() < dthreshold
        direct = deform(initTmpPs, batch_inds) - cam_pos.view(1, 3)
        check2 = _ray_angle_deg(direct, rays) < athreshold
        unfinished = ~(check1 * check2)
    for ind in range(times):
        active = unfinished.nonzero(as_tuple=True)[0]
        if active.numel() == 0:
            break
        curPs = initTmpPs[active].detach().clone()
        curPs.requires_grad_(True)
        loss1 = (tmpSdf(curPs, ratio).abs()).view(-1)
        defPs = deform(curPs, batch_inds[active])
        direct = defPs - cam_pos.view(1, 3)
        up = torch.linalg.cross(direct, rays[active], dim=1)
        loss2 = (up.norm(dim=1) / direct.norm(dim=1)).abs()
        loss = w1 * loss1 + w2 * loss2
        grad = torch.autograd.grad(loss.sum(), curPs, retain_graph=False, create_graph=False,
                                   only_inputs=True)[0]
        t = -loss / (grad * grad).sum(1)
        curPs = (curPs + t.view(-1, 1) * grad).detach()
        initTmpPs[active] = curPs
        with torch.no_grad():
            check1 = tmpSdf(curPs, ratio).view(-1).abs() < dthreshold
            direct = deform(curPs, batch_inds[active]) - cam_pos.view(1, 3)
            check2 = _ray_angle_deg(direct, rays[active]) < athreshold
            unfinished[active[check1 * check2]] = False
    return initTmpPs.detach(), ~unfinished


def _is_garment_deformer(deformer, defconds, initTmpPs, tmpSdf):
    """The graph-free HIP passes serve a CompositeDeformer (offset MLP + skinner) with conds [code table, [poses, trans]]."""
    return (initTmpPs.is_cuda and hasattr(deformer, 'ray_energy_and_vjp') and hasattr(tmpSdf, 'chain')
            and isinstance(defconds, (list, tuple)) and len(defconds) == 2 and torch.is_tensor(defconds[0])
            and isinstance(defconds[1], (list, tuple)))


def OptimizeGarmentSurfaceSinlge(cam_pos, rays, initTmpPs, batch_inds, tmpSdf, ratio, deformer, defconds, dthreshold=5.e-5,
                                 athreshold=0.02, w1=3.05, w2=1., times=5, offset_type=None):
    """utils/FindSurfacePs.py:210-272 (name as the reference spells it): the root finder for ONE garment net — the same iteration as
    OptimizeGarmentSurfacePs on one (rays, start points, net, code table) with `offset_type` naming the garment's offset slot.
    The reference's callers (the visualisation / evaluation paths, OptimGarmentNetwork.py:2109, :2837, :3187, :3282) stop at
    |f| < 1e-4 after at most 30 steps.  Returns (points, converged); on the device this is the loop's solver (csrc kernels)."""
    if _is_garment_deformer(deformer, defconds, initTmpPs, tmpSdf):
        outs, oks = _optimize_explicit_all(cam_pos, [rays], [initTmpPs], [batch_inds], [tmpSdf], ratio, deformer, [defconds[0]],
                                           defconds[1], [offset_type], dthreshold, athreshold, w1, w2, times)
        return outs[0].detach(), oks[0]
    return _optimize_generic(cam_pos, rays, initTmpPs, batch_inds, tmpSdf, ratio,
                             lambda p, b: deformer(p, defconds, b, ratio=ratio, offset_type=offset_type),
                             dthreshold, athreshold, w1, w2, times)


def OptimizeSurfacePs(cam_pos, rays, initTmpPs, batch_inds, tmpSdf, ratio, deformer, defconds, dthreshold=5.e-5, athreshold=0.02,
                      w1=3.05, w2=1., times=5):
    """utils/FindSurfacePs.py:145-207: the base-class loop's root finder (OptimNetwork.py:523: times=10; :268, :328: 1e-4 / 30) —
    one SDF net, any deformer, NO `offset_type` handed on (with the garment deformer the reference's MLPTranslator raises KeyError
    on that, model/Deformer.py:177; here its offset slot is `None`)."""
    if _is_garment_deformer(deformer, defconds, initTmpPs, tmpSdf):
        outs, oks = _optimize_explicit_all(cam_pos, [rays], [initTmpPs], [batch_inds], [tmpSdf], ratio, deformer, [defconds[0]],
                                           defconds[1], [None], dthreshold, athreshold, w1, w2, times)
        return outs[0].detach(), oks[0]
    return _optimize_generic(cam_pos, rays, initTmpPs, batch_inds, tmpSdf, ratio,
                             lambda p, b: deformer(p, defconds, b, ratio=ratio),
                             dthreshold, athreshold, w1, w2, times)


def _make_stream(device):
    from .. import _lib
    return _lib.make_stream(device)
