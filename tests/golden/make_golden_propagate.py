"""Golden vectors for the implicit differentiation of the surface point, produced by the REAL reference method
`OptimGarmentNetwork.propagateTmpPsGrad` (engineer/networks/OptimGarmentNetwork.py:2159-2313) called on a stand-in
`self` that carries exactly the attributes the method reads (reference SDF net, reference CompositeDeformer, per-frame
tensors, ray bookkeeping).  Two substitutions, both outside the arithmetic under test: `Fast3x3Minv` is the C oracle
(the CUDA extension cannot be built here) and `RectifiedPerspectiveCameras` is recmv's stand-alone restatement (the
reference class derives from pytorch3d's CamerasBase, absent here; its ray / centre formulas are pinned separately).

`propagate_large.npz`: the same state through `OptimGarmentNetwork_LargePose.propagateTmpPsGrad`
(engineer/networks/OptimGarmentNetwork_Large_Pose.py:326-475) after its `freeze_sdf` (:130-137): the SDF nets are
frozen and receive nothing, the deformer / per-frame / camera gradients are injected as before.

    python tests/golden/make_golden_propagate.py
"""
import sys
import types
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import ref_loader  # noqa: E402

ref_loader.install()
import common_setup as cs  # noqa: E402
from make_golden import save  # noqa: E402

SDF_KEYS = ["lin0.weight_v", "lin4.weight_g", "lin8.bias", "lin3.bias"]
TR_KEYS = ["lin0.weight", "lin2.bias", "lin4.weight"]


def inputs(sdf, comp, ratio):
    """Seeded state of one garment right after `loss.backward()`: P surface points near the zero level with a
    gradient, their pixels / frames, per-frame codes, poses, camera."""
    P, N = 96, 3
    p = cs.rootfind_init(sdf, P, seed=21).detach()
    g = torch.Generator().manual_seed(22)
    return dict(p=p, grad_l_p=torch.randn(P, 3, generator=g),
                col=torch.randint(40, 470, (P,), generator=g), row=torch.randint(40, 470, (P,), generator=g),
                binds=torch.randint(0, N, (P,), generator=g),
                focal=torch.tensor([[1000., 990.]]), pp=torch.tensor([[256., 250.]]),
                R=torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3), T=torch.tensor([[0.1, -0.2, 3.0]]))


def main(large_pose=False):
    N = ref_loader.ref_module("model.network")
    Dref = ref_loader.ref_module("model.Deformer")
    OGN = ref_loader.ref_module("engineer.networks.OptimGarmentNetwork")
    from recmv.model import RectifiedPerspectiveCameras as OurCameras
    OGN.RectifiedPerspectiveCameras = OurCameras
    if large_pose:
        OGNL = ref_loader.ref_module("engineer.networks.OptimGarmentNetwork_Large_Pose")
        OGNL.RectifiedPerspectiveCameras = OurCameras
    sdf = cs.build_sdf(N.getTmpSdf)
    tr = cs.build_translator(Dref.MLPTranslator)
    sk = cs.build_skinner(Dref.LBSkinner, Dref.batch_rodrigues)
    comp = Dref.CompositeDeformer([tr, sk])
    ratio = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}
    x = inputs(sdf, comp, ratio)
    conds, _ = cs.conds_and_inds(8, nframes=3, condlen=128, seed=4)
    poses, trans = cs.poses_trans(3, seed=7)
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    leaves = dict(conds=leaf(conds), poses=leaf(poses), trans=leaf(trans), focal=leaf(x["focal"]), pp=leaf(x["pp"]),
                  T=leaf(x["T"]))
    p = x["p"].clone().requires_grad_(True)
    p.grad = x["grad_l_p"].clone()
    cam0 = OurCameras(leaves["focal"], leaves["pp"], x["R"], leaves["T"], image_size=[(512, 512)])
    rays = cam0.view_rays(torch.stack([x["col"], x["row"], torch.ones_like(x["col"])], -1).float())

    fake = types.SimpleNamespace()
    fake.garment_size, fake.garment_names = 1, ["upper"]
    fake.TmpPs, fake.rays, fake.col_inds, fake.row_inds, fake.batch_inds = [p], [rays], [x["col"]], [x["row"]], [x["binds"]]
    fake.info = {}
    fake.get_grad_parameters = lambda frame_ids, device: ([None, leaves["conds"]], leaves["poses"], leaves["trans"], None)
    fake.dataset = types.SimpleNamespace(
        get_camera_parameters=lambda n, device: (leaves["focal"], leaves["pp"], x["R"], leaves["T"], 512, 512))
    fake.maskRender = types.SimpleNamespace(rasterizer=types.SimpleNamespace(cameras=None))
    fake.garment_nets, fake.deformer, fake.sdf = [sdf], comp, sdf
    for m in (sdf, comp):
        for q in m.parameters():
            q.grad = None
    if large_pose:
        fake.garment_nets = torch.nn.ModuleList([sdf])
        OGNL.OptimGarmentNetwork_LargePose.freeze_sdf(fake)                  # the reference's own freezing (:130-137)
        assert not any(q.requires_grad for q in sdf.parameters())
        OGNL.OptimGarmentNetwork_LargePose.propagateTmpPsGrad(fake, torch.arange(3), ratio)
        assert all(q.grad is None for q in sdf.parameters()), "frozen SDF nets receive no gradient"
    else:
        OGN.OptimGarmentNetwork.propagateTmpPsGrad(fake, torch.arange(3), ratio)
    sp, tp = dict(sdf.named_parameters()), dict(tr.named_parameters())
    out = {} if large_pose else {"g_sdf_" + k.replace(".", "_"): sp[k].grad for k in SDF_KEYS}
    out.update({"g_tr_" + k.replace(".", "_"): tp[k].grad for k in TR_KEYS})
    out.update({"g_" + k: v.grad for k, v in leaves.items()})
    n_inv = fake.info["upper_invInfo"]
    print("propagate: invertible %d / %d" % (n_inv[1], n_inv[0]))
    save("propagate_large" if large_pose else "propagate", p=x["p"], grad_l_p=x["grad_l_p"], col=x["col"], row=x["row"], binds=x["binds"], focal=x["focal"],
         pp=x["pp"], R=x["R"], T=x["T"], conds=conds.detach(), poses=poses.detach(), trans=trans.detach(), inv_total=n_inv[0], inv_ok=n_inv[1],
         **out)


if __name__ == "__main__":
    main()
    main(large_pose=True)
