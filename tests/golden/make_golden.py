"""Generate golden vectors by running the REAL reference Python code (imported from /root/reference
through tests/golden/ref_loader.py) on CPU.  Run in the build container:

    python tests/golden/make_golden.py

Writes tests/golden/*.npz.  The fixtures are small; networks are never stored — both sides rebuild them
from the same seed with the same construction order (tests verify a parameter fingerprint first).

What each file pins (reference file:line):
  embedder.npz     model/Embedder.py:4-65, utils/utils.py:40-46 (annealing weights)
  sdf.npz          model/network.py:27-133   forward, rendcond, gradient(), eikonal grad wrt params
  translator.npz   model/Deformer.py:141-206 forward (+offset), Jacobian, param grad
  render.npz       model/RenderNet.py:10-96
  lbs.npz          model/Deformer.py:216-445 (extensions backed by the oracle), Jacobian, 2nd-order grad
  rays.npz         utils/utils.py:133-250 compute_cardinal_rays / compute_deformed_normals
  rootfind.npz     utils/FindSurfacePs.py:273-353 OptimizeGarmentSurfacePs
  seg3d.npz        MCAcc/seg3d_lossless.py:233-428 (+ MCGpu contract through the oracle)
  camera.npz       model/CameraMine.py:146-208 (formulas only; the class needs pytorch3d)
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_loader  # noqa: E402

ref_loader.install()
import common_setup as cs  # noqa: E402  (shared seeded input builders, also used by the tests)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(HERE / (name + ".npz"), **out)
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items()})


def main():
    torch.set_num_threads(8)
    Emb = ref_loader.ref_module("model.Embedder")
    N = ref_loader.ref_module("model.network")
    Dref = ref_loader.ref_module("model.Deformer")
    Rref = ref_loader.ref_module("model.RenderNet")
    Uref = ref_loader.ref_module("utils.utils")
    Fref = ref_loader.ref_module("utils.FindSurfacePs")
    Sref = ref_loader.ref_module("MCAcc.seg3d_lossless")

    # ---------------------------------------------------------------- embedder + annealing weights
    x = cs.points(64, seed=1)
    embed, out_dim = Emb.get_embedder(6)
    ws = Uref.annealing_weights(6, 0.62)
    save("embedder", x=x, out_dim=out_dim, plain=embed(x), weighted=embed(x, ws), ws=np.array(ws),
         ws_grid=np.array([Uref.annealing_weights(6, r) for r in cs.RATIOS]), ratios=np.array(cs.RATIOS),
         embed4=Emb.get_embedder(4)[0](x))

    # ---------------------------------------------------------------- SDF net
    sdf = cs.build_sdf(N.getTmpSdf)
    ratio = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}
    xs = cs.points(257, seed=2, scale=0.7)
    with torch.no_grad():
        y = sdf(xs, ratio)
        rend = sdf.rendcond.clone()
        y_none = sdf(xs, 1.0)
    xg = xs.clone().requires_grad_(True)
    yg = sdf(xg, ratio)
    grad = sdf.gradient(xg, yg)
    eik = ((grad.norm(2, dim=-1) - 1) ** 2).mean() + 0.1 * yg.abs().mean()
    params = dict(sdf.named_parameters())
    gsel = torch.autograd.grad(eik, [params[k] for k in cs.SDF_GRAD_KEYS])
    save("sdf", x=xs, fingerprint=cs.fingerprint(sdf), y=y, rendcond=rend, y_ratio1=y_none, grad=grad, eik=eik,
         **{"g_" + k.replace(".", "_"): g for k, g in zip(cs.SDF_GRAD_KEYS, gsel)})

    # ---------------------------------------------------------------- deformer MLP
    tr = cs.build_translator(Dref.MLPTranslator)
    ps = cs.points(300, seed=3, scale=0.6)
    conds, binds = cs.conds_and_inds(300, nframes=3, condlen=128, seed=4)
    pg = ps.clone().requires_grad_(True)
    d = tr(pg, conds, binds, ratio=ratio, offset_type="upper")
    off = tr.offset["upper"].detach().clone()
    J = Uref.compute_Jacobian(pg, d, True, True)
    lossJ = (J ** 2).sum() + d.sum()
    gW = torch.autograd.grad(lossJ, [tr.lin0.weight, tr.lin4.bias, conds], allow_unused=True)
    psb = cs.points(2 * 50, seed=5, scale=0.6).view(2, 50, 3)
    with torch.no_grad():
        db = tr(psb, conds[:2], None, ratio=ratio, offset_type="upper")
    save("translator", fingerprint=cs.fingerprint(tr), ps=ps, conds=conds, binds=binds, d=d, offset=off, J=J, lossJ=lossJ, g_lin0_weight=gW[0], g_lin4_bias=gW[1], g_conds=gW[2], psb=psb, db=db)

    # ---------------------------------------------------------------- colour net
    rn = cs.build_render(Rref.RenderingNetwork_view_norm)
    rp, rnorm, rview, rfeat = cs.render_inputs(200, seed=6)
    rpg = rp.clone().requires_grad_(True)
    col = rn(rpg, rnorm, rview, rfeat, ratio)
    gcol = torch.autograd.grad(col.abs().sum(), [rpg, rn.lin0.weight_v, rn.lin4.weight_g])
    save("render", fingerprint=cs.fingerprint(rn), p=rp, n=rnorm, v=rview, f=rfeat, col=col, g_p=gcol[0],
         g_lin0_weight_v=gcol[1], g_lin4_weight_g=gcol[2])

    # ---------------------------------------------------------------- LBS (extensions on the oracle)
    sk = cs.build_skinner(Dref.LBSkinner, Dref.batch_rodrigues)
    poses, trans = cs.poses_trans(3, seed=7)
    poses.requires_grad_(True)
    trans.requires_grad_(True)
    lp = cs.points(240, seed=8, scale=0.5)
    lb = torch.randint(0, 3, (240,), generator=torch.Generator().manual_seed(9))
    lpg = lp.clone().requires_grad_(True)
    v = sk(lpg, [poses, trans], lb)
    Jl = Uref.compute_Jacobian(lpg, v, True, True)
    l2 = (Jl ** 2).sum() + (v ** 2).sum()
    g2 = torch.autograd.grad(l2, [poses, trans, lpg])
    with torch.no_grad():
        vb = sk(lp.view(3, 80, 3), [poses, trans], None)
        skel = sk.posedSkeleton([poses, trans])
    save("lbs", ps=lp, binds=lb, poses=poses, trans=trans, v=v, J=Jl, loss=l2, g_poses=g2[0], g_trans=g2[1],
         g_ps=g2[2], vb=vb, skel=skel, init_pose=sk.init_pose)

    # ---------------------------------------------------------------- cardinal rays / deformed normals
    comp = Dref.CompositeDeformer([tr, sk])
    rp2 = cs.points(150, seed=10, scale=0.45)
    rb = torch.randint(0, 3, (150,), generator=torch.Generator().manual_seed(11))
    rays = torch.nn.functional.normalize(cs.points(150, seed=12), dim=1)
    defconds = [conds, [poses.detach(), trans.detach()]]
    rpg2 = rp2.clone().requires_grad_(True)
    crays, ds = Uref.compute_cardinal_rays(comp, rpg2, rays, defconds, rb, ratio, 'train', offset_type="upper")
    rpg3 = rp2.clone().requires_grad_(True)
    nx, ds2 = Uref.compute_deformed_normals(sdf, comp, rpg3, defconds, rb, ratio, 'test', offset_type="upper")
    save("rays", ps=rp2, binds=rb, rays=rays, crays=crays, ds=ds, nx=nx, ds2=ds2)

    # ---------------------------------------------------------------- root finder
    cam_pos = torch.tensor([0.05, -0.02, 3.0])
    init = cs.rootfind_init(sdf, 400, seed=13)
    rbinds = torch.randint(0, 3, (400,), generator=torch.Generator().manual_seed(14))
    with torch.no_grad():
        d0 = comp(init, defconds, rbinds, ratio=ratio, offset_type="upper")
        rrays = torch.nn.functional.normalize(d0 - cam_pos.view(1, 3), dim=1)
        # perturb the start points so that the finder has work to do
        start = init + 0.002 * cs.points(400, seed=15)
    outs, checks = Fref.OptimizeGarmentSurfacePs(cam_pos, [rrays], [start.clone()], [rbinds], [sdf], ratio, comp,
                                                 [[conds], [poses.detach(), trans.detach()]], garment_names=["upper"],
                                                 dthreshold=5.e-5, athreshold=0.02, w1=3.05, w2=1., times=20)
    outs1, checks1 = Fref.OptimizeGarmentSurfacePs(cam_pos, [rrays], [start.clone()], [rbinds], [sdf], ratio, comp,
                                                   [[conds], [poses.detach(), trans.detach()]],
                                                   garment_names=["upper"], dthreshold=5.e-5, athreshold=0.02, w1=3.05,
                                                   w2=1., times=1)
    print("rootfind: converged %d / %d" % (int(checks[0].sum()), checks[0].numel()))
    save("rootfind", cam_pos=cam_pos, start=start, binds=rbinds, rays=rrays, out=outs[0], check=checks[0], out1=outs1[0],
         check1=checks1[0])

    # ---------------------------------------------------------------- Seg3dLossless + MC
    def query(points):
        with torch.no_grad():
            return sdf.forward(points.reshape(-1, 3), 1.0).reshape(1, 1, -1)

    eng = Sref.Seg3dLossless(query_func=query, b_min=[-1.0, -1.1, -0.9], b_max=[1.0, 1.1, 0.9],
                             resolutions=[(9, 11, 7), (17, 21, 13), (33, 41, 25)], align_corners=False,
                             balance_value=0.0, use_cuda_impl=False, faster=False)
    grid = eng.forward()
    import MCGpu as mc_stub
    verts, faces = mc_stub.mc_gpu(grid[0, 0].permute(2, 1, 0).contiguous(), eng.spacing_x, eng.spacing_y,
                                  eng.spacing_z, eng.bx, eng.by, eng.bz, 0.0)
    save("seg3d", grid=grid, verts=verts, faces=faces, spacing=np.array([eng.spacing_x, eng.spacing_y, eng.spacing_z]),
         b=np.array([eng.bx, eng.by, eng.bz]))


if __name__ == "__main__":
    main()
