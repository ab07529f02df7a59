"""Surface-point finder and ray/surface root finder (utils/FindSurfacePs.py of the reference).

  FindSurfacePs             :7-60     pixels whose first valid face has all barycentrics > 0
  OptimizeGarmentSurfacePs  :273-353  find canonical p with |SDF(p)| < dthreshold whose deformed image lies on
                                      the pixel's ray (angle < athreshold degrees), <= `times` steps of
                                      p <- p - E * gradE / |gradE|^2,  E = w1*|f(p)| + w2*|(d-c) x v| / |d-c|

The SDF / deformer evaluations inside run on the recmv kernels.  The active set is kept as an index list
that shrinks on device; the loop exits early through a single count read per step (the reference does the
same sync through `curPs.shape[0]==0` after boolean indexing, :311-313).
"""
import numpy as np
import torch

__all__ = ["FindSurfacePs", "OptimizeGarmentSurfacePs"]


def FindSurfacePs(TmpVs, TmpFaces, frags):
    """frags: object with pix_to_face [N,H,W,K] (int64, -1 = empty) and bary_coords [N,H,W,K,3].
    Returns (batch_inds, row_inds, col_inds, initTmpPs, finds) in `nonzero` (row-major) order."""
    N, H, W, K = frags.pix_to_face.shape
    pix_to_face = frags.pix_to_face
    bary_coords = frags.bary_coords
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


@torch.no_grad()
def _optimize_explicit(cam_pos, rays, initTmpPs, batch_inds, tmpSdf, ratio, deformer, defconds, smpl_conds, name,
                       dthreshold, athreshold, w1, w2, times):
    """Same iteration as the autograd version below on the graph-free passes: per step two C calls for the SDF net
    (value + input gradient), four for the deformer (offset MLP, fused skinning + ray energy, and their VJPs) and one
    fused stopping-test / update kernel.  All rays are carried through every step (rows of the kernels are
    independent, so the active rays get the same updates); finished rays are simply not updated.  The post-update
    check of step i is the forward pass of step i+1, evaluated once.  The early exit reads the unfinished-ray count
    of the PREVIOUS step (pinned host copy + event), so the host never waits on the step it has just enqueued; the
    one extra evaluation this can cost changes nothing (no unfinished ray = no update)."""
    from .. import chains
    dev = initTmpPs.device
    p = initTmpPs.detach().clone().contiguous()
    P = p.shape[0]
    cam = cam_pos.detach().reshape(3).contiguous().float()
    rays = rays.detach().contiguous()
    frame = batch_inds.contiguous()
    conds = [defconds, smpl_conds]
    unfinished = torch.ones(P, dtype=torch.uint8, device=dev)
    counters = torch.zeros(times + 1, dtype=torch.int32, device=dev)
    host = torch.zeros(times + 1, dtype=torch.int32).pin_memory()
    events = []
    sdf_chain = tmpSdf.chain(tmpSdf._pe_weights(ratio), need_t=True)
    assert tmpSdf.d_out == 1
    for it in range(times + 1):
        f = sdf_chain.forward(p, n_out=1, keep=True)
        gf = sdf_chain.vjp_input(p, None)
        _, loss2, angle, gd = deformer.ray_energy_and_vjp(p, conds, frame, cam, rays, ratio=ratio, offset_type=name)
        chains.rootfind_update(p, f, gf, loss2, angle, gd, unfinished, counters[it:it + 1], dthreshold, athreshold,
                               w1, w2, it < times)
        host[it:it + 1].copy_(counters[it:it + 1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        events.append(ev)
        if it >= 1:
            events[it - 1].synchronize()
            if int(host[it - 1]) == 0:
                break
    return p, unfinished == 0


def OptimizeGarmentSurfacePs(cam_pos, rays_list, initTmpPs_list, batch_inds_list, tmpSdf_nets, ratio, deformer,
                             defconds_list, garment_names, dthreshold=5.e-5, athreshold=0.02, w1=3.05, w2=1.,
                             times=5):
    smpl_conds = defconds_list[1]
    optimized_init_tmp_ps_list = []
    optimized_check_list = []
    for garment_idx, (initTmpPs, batch_inds, defconds, rays, garment_name) in enumerate(
            zip(initTmpPs_list, batch_inds_list, defconds_list[0], rays_list, garment_names)):
        tmpSdf = tmpSdf_nets[garment_idx]
        if initTmpPs.is_cuda and hasattr(tmpSdf, 'value_and_grad') and hasattr(deformer, 'value_and_vjp'):
            pts, ok = _optimize_explicit(cam_pos, rays, initTmpPs, batch_inds, tmpSdf, ratio, deformer, defconds,
                                         smpl_conds, garment_name, dthreshold, athreshold, w1, w2, times)
            optimized_init_tmp_ps_list.append(pts.detach())
            optimized_check_list.append(ok)
            continue
        with torch.no_grad():
            check1 = tmpSdf(initTmpPs, ratio).view(-1).abs() < dthreshold
            direct = deformer(initTmpPs, [defconds, smpl_conds], batch_inds, ratio=ratio,
                              offset_type=garment_name) - cam_pos.view(1, 3)
            check2 = _ray_angle_deg(direct, rays) < athreshold
            unfinished = ~(check1 * check2)
        for ind in range(times):
            active = unfinished.nonzero(as_tuple=True)[0]
            if active.numel() == 0:
                break
            curPs = initTmpPs[active].detach().clone()
            curPs.requires_grad_(True)
            loss1 = (tmpSdf(curPs, ratio).abs()).view(-1)
            defPs = deformer(curPs, [defconds, smpl_conds], batch_inds[active], ratio=ratio,
                             offset_type=garment_name)
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
                direct = deformer(curPs, [defconds, smpl_conds], batch_inds[active], ratio=ratio,
                                  offset_type=garment_name) - cam_pos.view(1, 3)
                check2 = _ray_angle_deg(direct, rays[active]) < athreshold
                unfinished[active[check1 * check2]] = False
        optimized_init_tmp_ps_list.append(initTmpPs.detach())
        optimized_check_list.append(~unfinished)
    return optimized_init_tmp_ps_list, optimized_check_list
