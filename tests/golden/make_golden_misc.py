"""Golden vectors for small host-side pieces of the iteration, from the REAL reference code:

  misc.npz   utils/utils.py:293-305 DCTNullSpace; OptimGarmentNetwork.dct_poses_loss (:1221-1250) with the reference
             LBSkinner.posedSkeleton and the reference dataset's get_batchframe_data (dataset/dataset.py:438-457) on a
             stand-in `self`; utils.GMRobustError

    python tests/golden/make_golden_misc.py
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


def main():
    ref_loader.ref_module("model.network")
    Dref = ref_loader.ref_module("model.Deformer")
    Uref = ref_loader.ref_module("utils.utils")
    OGN = ref_loader.ref_module("engineer.networks.OptimGarmentNetwork")
    DS = ref_loader.ref_module("dataset.dataset")
    null = Uref.DCTNullSpace(10, 30)
    sk = cs.build_skinner(Dref.LBSkinner, Dref.batch_rodrigues)
    F = 40
    g = torch.Generator().manual_seed(31)
    poses = (0.15 * torch.randn(F, 24, 3, generator=g)).requires_grad_(True)
    trans = (0.02 * torch.randn(F, 3, generator=g)).requires_grad_(True)
    ds_cls = [v for v in vars(DS).values() if isinstance(v, type) and getattr(v, "__module__", "") == DS.__name__
              and "get_batchframe_data" in vars(v)][0]
    fake_ds = types.SimpleNamespace(video_segmented_index=[], frame_num=F, poses=poses, trans=trans)
    fake_ds.get_batchframe_data = lambda name, fids, bs: ds_cls.get_batchframe_data(fake_ds, name, fids, bs)
    fake = types.SimpleNamespace(dctnull=null, dataset=fake_ds, deformer=types.SimpleNamespace(defs=[None, sk]),
                                 info={}, conf=types.SimpleNamespace(get_float=lambda k: 2.0))
    frame_ids = torch.tensor([0, 17, 39])                      # window clamped at both ends and free in the middle
    loss = OGN.OptimGarmentNetwork.dct_poses_loss(fake, poses[frame_ids], trans[frame_ids], frame_ids.clone(), 3)
    gp, gt = torch.autograd.grad(loss, [poses, trans], allow_unused=True)
    gt = torch.zeros_like(trans) if gt is None else gt
    win, rel = fake_ds.get_batchframe_data('poses', torch.tensor([0, 17, 39]), 30)
    # ---- compute_garment_pc_loss (:621-667) on a stand-in self: IoU of a soft silhouette + LBS-consistency term
    from recmv.hocon import ConfigFactory
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf")).get_config('loss_coarse')
    gg = torch.Generator().manual_seed(32)
    Np, V, Hh, Ww = 3, 64, 20, 24
    imgs = torch.rand(Np, Hh, Ww, 1, generator=gg).requires_grad_(True)
    gtM = (torch.rand(Np, Hh, Ww, generator=gg) > 0.5).float()
    gverts = (0.4 * torch.randn(V, 3, generator=gg)).requires_grad_(True)
    pz, tz = cs.poses_trans(Np, seed=33)
    pz, tz = pz.detach(), tz.detach()
    defv = (sk(gverts.view(1, -1, 3).expand(Np, -1, 3), [pz, tz]) + 0.02 * torch.randn(Np, V, 3, generator=gg))
    fake2 = types.SimpleNamespace(conf=conf, info={'pc_loss': {}}, deformer=types.SimpleNamespace(defs=[None, sk]))
    fake_mesh = types.SimpleNamespace(verts_padded=lambda: defv)
    pc = OGN.OptimGarmentNetwork.compute_garment_pc_loss(fake2, fake_mesh, [None, [pz, tz]], imgs, gtM, None, 'upper',
                                                         gverts, torch.zeros(4, 3, dtype=torch.long), 0)
    g_img, g_v = torch.autograd.grad(pc, [imgs, gverts])
    x = torch.linspace(0, 0.3, 50)
    save("misc", dctnull=null, poses=poses, trans=trans, frame_ids=frame_ids, dct_loss=loss, g_poses=gp, g_trans=gt,
         window=win, rel=rel, pc_imgs=imgs, pc_gt=gtM, pc_verts=gverts, pc_poses=pz, pc_trans=tz, pc_def=defv,
         pc_loss=pc, pc_g_img=g_img, pc_g_verts=g_v, gm_x=x, gm_true=Uref.GMRobustError(x, 0.01, True), gm_false=Uref.GMRobustError(x, 0.5, False))




def sample_rays_fixture():
    """OptimGarmentNetwork.sample_train_ray (:983-1055) for real: ground-truth-mask selection, host-RNG Bernoulli
    subset, rays of the kept pixels (camera: recmv's restatement, see make_golden_propagate.py)."""
    ref_loader.ref_module("model.network")
    OGN = ref_loader.ref_module("engineer.networks.OptimGarmentNetwork")
    from recmv.model import RectifiedPerspectiveCameras as OurCameras
    OGN.RectifiedPerspectiveCameras = OurCameras
    g = torch.Generator().manual_seed(51)
    N, H, W = 3, 48, 40
    lists = []
    for P in (2500, 900):                                          # garment 0 is subsampled, garment 1 is not
        lists.append(dict(b=torch.randint(0, N, (P,), generator=g), r=torch.randint(0, H, (P,), generator=g),
                          c=torch.randint(0, W, (P,), generator=g), p=torch.randn(P, 3, generator=g)))
    masks = [(torch.rand(N, H, W, generator=g) > 0.3).float() for _ in range(2)]
    focal, pp = torch.tensor([[80., 78.]]), torch.tensor([[20., 24.]])
    R, T = torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3), torch.tensor([[0.1, -0.2, 3.0]])
    fake = types.SimpleNamespace(conf={}, garment_size=2, pcRender=None,
                                 dataset=types.SimpleNamespace(get_camera_parameters=lambda n, d: (focal, pp, R, T, H, W)),
                                 maskRender=types.SimpleNamespace(rasterizer=types.SimpleNamespace(cameras=None)))
    torch.manual_seed(52)
    out = OGN.OptimGarmentNetwork.sample_train_ray(
        fake, N, 1024, masks, [l["b"] for l in lists], [l["r"] for l in lists], [l["c"] for l in lists],
        [l["p"] for l in lists], torch.arange(N), "cpu")
    sb, sr, sc, sp, rays, _cams = out
    arrs = dict(masks=torch.stack(masks), focal=focal, pp=pp, R=R, T=T)
    for i, l in enumerate(lists):
        arrs.update({f"in{i}_{k}": v for k, v in l.items()})
        arrs.update({f"out{i}_b": sb[i], f"out{i}_r": sr[i], f"out{i}_c": sc[i], f"out{i}_p": sp[i], f"out{i}_rays": rays[i]})
    print("sample_train_ray: kept", [int(t.numel()) for t in sb])
    save("sample_rays", **arrs)


if __name__ == "__main__":
    main()
    sample_rays_fixture()
