"""Golden vectors for the feature-curve branch as a WHOLE, from the REAL reference method `OptimGarmentNetwork.project_2d_loss`
(engineer/networks/OptimGarmentNetwork.py:1772-1883) on a stand-in `self`, with the reference's own `deform_feature_line`
(:1507-1603), `fl_visible_by_body_zbuff` (:1374-1448), `compute_fl_proj_loss` (:1605-1711), `fl_proj_loss`, `Intersect_Free_Curve`,
CompositeDeformer and SDF nets underneath: per-garment projection losses, canonical-curve SDF terms, the total, the gradients on
the curve parameters and the parameters after the AdamW step.

What is stood in: the mesh rasteriser behind `maskRender` (pytorch3d, absent) by the C oracle's, `Meshes` by a holder,
`chamfer_distance` by the restatement of recmv.curves, the camera by recmv's restatement; `Intersect_Free_Curve` is constructed
without its mesh-extraction front end (trimesh), as in make_golden_curves.py.

    python tests/golden/make_golden_project2d.py
"""
import sys
import types
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import ref_loader  # noqa: E402

ref_loader.install()
import common_setup as cs  # noqa: E402
import mask_loss_case as mlc  # noqa: E402
import project2d_case as pc  # noqa: E402
from make_golden import save  # noqa: E402


def main():
    Nref = ref_loader.ref_module("model.network")
    Dref = ref_loader.ref_module("model.Deformer")
    G = ref_loader.ref_module("engineer.utils.garment_structure")
    OGN = ref_loader.ref_module("engineer.networks.OptimGarmentNetwork")
    from oracle import oracle as orc
    from recmv import curves as ours
    from recmv.hocon import ConfigFactory
    from recmv.model import RectifiedPerspectiveCameras as OurCameras
    OGN.fl_proj_loss.__globals__["chamfer_distance"] = lambda a, b, point_reduction='sum': (ours.chamfer_distance_sum(a, b), None)
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf")).get_config('loss_coarse')
    st = pc.state()
    cam = OurCameras(st['focal'], st['pp'], st['R'], st['T'], image_size=[(pc.W, pc.H)])

    class Meshes:
        def __init__(self, verts, faces):
            self.verts, self.faces = verts, faces

    OGN.Meshes = Meshes

    class MaskRender:
        rasterizer = types.SimpleNamespace(cameras=cam)

        def __call__(self, meshes):
            verts = torch.stack([v.detach() for v in meshes.verts])
            faces = meshes.faces[0]
            n, f = verts.shape[0], faces.shape[0]
            ndc = cam.transform_points_ndc(verts.reshape(-1, 3)).view(n, -1, 3)
            fv = ndc[:, faces.reshape(-1)].reshape(-1, 3, 3)
            p2f, zbuf, bary, dists = orc.rasterize_meshes(fv, torch.arange(n) * f, torch.full((n,), f), (pc.H, pc.W))
            return None, types.SimpleNamespace(zbuf=zbuf, pix_to_face=p2f)

    sdfs = mlc.build_sdfs(Nref.getTmpSdf)
    tr = cs.build_translator(Dref.MLPTranslator)
    sk = cs.build_skinner(Dref.LBSkinner, Dref.batch_rodrigues)
    comp = Dref.CompositeDeformer([tr, sk])
    ref = object.__new__(G.Intersect_Free_Curve)
    torch.nn.Module.__init__(ref)
    ref.cano2canosmpl = lambda lst, nm: [0.9 * c for c in lst]
    ref.fl_names, ref.sample_num = list(pc.NAMES), pc.S
    ref.initialize_parameters([c.clone() for c in st['curves']])
    with torch.no_grad():
        ref.scale.copy_(st['scale'])
        ref.nx_scale.copy_(st['nx_scale'])
    names = ['short_sleeve_upper', 'long_pants']                       # the reference's garment names key FL_EXTRACT
    fake = types.SimpleNamespace(conf=conf, info={}, garment_size=2, garment_names=names, garment_vs=[st['verts_u'], st['verts_b']],
                                 garment_fs=[st['faces_u'], st['faces_b']], garment_nets=sdfs, deformer=comp, sdfShrinkRadius=0.0,
                                 tmpBodyVs=st['body_v'], tmpBodyFs=st['body_f'], inter_free_curve=ref, fl_names=list(pc.NAMES),
                                 maskRender=MaskRender(), dataset=types.SimpleNamespace(fl_weights=dict(pc.WEIGHTS)))
    fake.get_grad_parameters = lambda fids, dev: ([None, st['conds_u'], st['conds_b']], st['poses'], st['trans'], None)
    for name in ('deform_feature_line', 'fl_visible_by_body_zbuff', 'compute_fl_proj_loss'):
        setattr(fake, name, types.MethodType(getattr(OGN.OptimGarmentNetwork, name), fake))
    returned = []                                    # what compute_fl_proj_loss hands back per garment (projection + regulariser)
    inner = fake.compute_fl_proj_loss

    def recording(*a, **k):
        out = inner(*a, **k)
        returned.append(out[0].detach())
        return out

    fake.compute_fl_proj_loss = recording
    fake.fl_optimizer = torch.optim.AdamW(ref.parameters(), lr=1e-4)
    OGN.OptimGarmentNetwork.project_2d_loss(fake, pc.N, pc.H, pc.W, torch.arange(pc.N), pc.RATIO, cam, st['fl_masks'], st['gt'], 'cpu')
    info = fake.info['fl_loss']
    print({k: (float(v) if not isinstance(v, dict) else v) for k, v in info.items()})
    sdf_w = conf.get_float('fl_weight.sdf_weight')
    total = sum(10. * sdf_w * info['pc_%s_loss_sdf' % n] for n in names) + float(sum(returned))      # :1865
    res = dict(total=torch.tensor(total), g_scale=ref.scale.grad, g_nx=ref.nx_scale.grad, new_scale=ref.scale.detach(),
               new_nx=ref.nx_scale.detach())
    for n in names:
        res['proj_' + n] = torch.tensor(float(info['%s_project loss' % n]))
        res['sdf_' + n] = torch.tensor(float(info['pc_%s_loss_sdf' % n]))
    res.update({'in_' + k: v for k, v in st.items()})
    save("project2d", **res)


if __name__ == "__main__":
    main()
