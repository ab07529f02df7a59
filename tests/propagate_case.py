"""Shared driver of the propagateTmpPsGrad golden test (CPU port and GPU): rebuild the fixture's state on recmv
modules, run `HotLoop.propagateTmpPsGrad` on a stand-in `self`, return the injected gradients."""
import types

import torch

SDF_KEYS = ["lin0.weight_v", "lin4.weight_g", "lin8.bias", "lin3.bias"]
TR_KEYS = ["lin0.weight", "lin2.bias", "lin4.weight"]


def run(g, sdf, tr, comp, device, large_pose=False):
    from recmv.loop import HotLoop
    from recmv.model import RectifiedPerspectiveCameras
    dev = torch.device(device)
    leaf = lambda t: t.detach().clone().to(dev).requires_grad_(True)
    leaves = dict(conds=leaf(g["conds"]), poses=leaf(g["poses"]), trans=leaf(g["trans"]), focal=leaf(g["focal"]),
                  pp=leaf(g["pp"]), T=leaf(g["T"]))
    p = leaf(g["p"])
    p.grad = g["grad_l_p"].clone().to(dev)
    R = g["R"].to(dev)
    cams = lambda: RectifiedPerspectiveCameras(leaves["focal"], leaves["pp"], R, leaves["T"], image_size=[(512, 512)])
    col, row, binds = g["col"].to(dev), g["row"].to(dev), g["binds"].to(dev)
    rays = cams().view_rays(torch.stack([col, row, torch.ones_like(col)], -1).float())
    fake = types.SimpleNamespace()
    fake.garment_size, fake.garment_names = 1, ["upper"]
    fake.TmpPs, fake.rays, fake.col_inds, fake.row_inds, fake.batch_inds = [p], [rays], [col], [row], [binds]
    fake.info = {}
    fake.get_grad_parameters = lambda frame_ids, d: ([None, leaves["conds"]], leaves["poses"], leaves["trans"], None)
    fake._cameras = cams
    fake.garment_nets, fake.deformer = [sdf], comp
    for m in (sdf, comp):
        for q in m.parameters():
            q.grad = None
    if large_pose:                             # OptimGarmentNetwork_LargePose.freeze_sdf (:130-137)
        fake.sdf = sdf
        HotLoop.freeze_sdf(fake)
    ratio = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}
    HotLoop.propagateTmpPsGrad(fake, torch.arange(3, device=dev), ratio)
    sp, tp = dict(sdf.named_parameters()), dict(tr.named_parameters())
    if large_pose:
        assert all(q.grad is None for q in sdf.parameters()), "frozen SDF nets receive no gradient"
    out = {} if large_pose else {"g_sdf_" + k.replace(".", "_"): sp[k].grad for k in SDF_KEYS}
    out.update({"g_tr_" + k.replace(".", "_"): tp[k].grad for k in TR_KEYS})
    out.update({"g_" + k: v.grad for k, v in leaves.items()})
    n_total, n_ok = fake.info["upper_invInfo"]
    return out, int(n_total), int(n_ok)


def compare(out, g, rtol, atol_rel):
    for k, got in out.items():
        want = g[k]
        assert got is not None, k
        got = got.detach().cpu()
        scale = float(want.abs().max())
        assert torch.allclose(got, want, rtol=rtol, atol=atol_rel * max(scale, 1e-12)), (
            k, float((got - want).abs().max()), scale)
