"""The project_2d_loss case shared by tests/golden/make_golden_project2d.py (reference method) and the tests (HotLoop on the CPU
port): two explicit garment meshes, a body mesh, six explicit curves, three frames of 60 x 48 with 2-D feature lines."""
import math

import torch

N, H, W, S, M = 3, 60, 48, 30, 20
UPPER = ['neck', 'left_cuff', 'right_cuff', 'upper_bottom']              # FL_EXTRACT['short_sleeve_upper']
BOTTOM = ['left_pant', 'right_pant']                                      # FL_EXTRACT['long_pants']
NAMES = UPPER + BOTTOM
RATIO = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}
WEIGHTS = {'neck': 1.0, 'left_cuff': 2.0, 'right_cuff': 0.5, 'upper_bottom': 1.5, 'left_pant': 1.2, 'right_pant': 0.8}


def state():
    import common_setup as cs
    import mask_loss_case as mlc
    g = torch.Generator().manual_seed(81)
    vu, fu = mlc._blob(9, 14, 0.30, (0.0, 0.18, 0.0), 92)
    vb, fb = mlc._blob(8, 12, 0.26, (0.0, -0.30, 0.0), 93)
    body_v, body_f = mlc._blob(8, 12, 0.24, (0.0, -0.05, 0.0), 94)
    t = torch.linspace(0, 2 * math.pi, S + 1)[:-1]
    ring_y = lambda y, r, cx=0.: torch.stack([cx + r * torch.cos(t), torch.full_like(t, y), r * torch.sin(t)], -1)
    ring_x = lambda x, r, y: torch.stack([torch.full_like(t, x), y + r * torch.cos(t), r * torch.sin(t)], -1)
    curves = torch.stack([ring_y(0.42, 0.12), ring_x(0.27, 0.08, 0.22), ring_x(-0.27, 0.08, 0.22), ring_y(-0.08, 0.27),
                          ring_y(-0.52, 0.09, 0.1), ring_y(-0.52, 0.09, -0.1)]).float()
    curves = curves + 0.004 * torch.randn(curves.shape, generator=g)
    conds_u, _ = cs.conds_and_inds(8, nframes=N, condlen=128, seed=4)
    conds_b, _ = cs.conds_and_inds(8, nframes=N, condlen=128, seed=5)
    poses, trans = cs.poses_trans(N, seed=7)
    scale = 1.0 + 0.2 * torch.randn(len(NAMES), S, 1, generator=g)
    scale[0, :4] = -0.3                                                  # some radial scales below zero: the ReLU branch
    nx_scale = 0.03 * torch.randn(len(NAMES), S, 1, generator=g)
    gt = torch.rand(N, len(NAMES) * M, 2, generator=g) * torch.tensor([W - 8., H - 8.]) + 4.
    fl_masks = torch.tensor([[1., 1., 0., 1., 1., 1.], [1., 1., 1., 1., 0., 1.], [0., 1., 1., 0., 1., 1.]])
    return dict(verts_u=vu, faces_u=fu, verts_b=vb, faces_b=fb, body_v=body_v, body_f=body_f, curves=curves,
                conds_u=conds_u.detach(), conds_b=conds_b.detach(), poses=poses.detach(), trans=trans.detach(), scale=scale,
                nx_scale=nx_scale, gt=gt, fl_masks=fl_masks, focal=torch.tensor([[105., 102.]]), pp=torch.tensor([[24., 30.]]),
                R=torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3), T=torch.tensor([[0.02, -0.05, 2.4]]))


def run(g, device, rtol=5e-4, rtol_grad=5e-3):
    """HotLoop.project_2d_loss on a stand-in `self` built from the fixture's inputs, against the reference method's outputs;
    returns the largest relative deviation per compared quantity."""
    import types
    from pathlib import Path
    import common_setup as cs
    import mask_loss_case as mlc
    from recmv import curves as fl
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    from recmv.model import CompositeDeformer, LBSkinner, MLPTranslator, RectifiedPerspectiveCameras, getTmpSdf
    repo = Path(__file__).resolve().parent.parent
    conf = ConfigFactory.parse_file(str(repo / "configs" / "synthetic" / "people_snapshot_like.conf")).get_config('loss_coarse')
    dev = torch.device(device)
    st = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith('in_')}
    sdfs = [n.to(dev) for n in mlc.build_sdfs(getTmpSdf)]
    tr = cs.build_translator(MLPTranslator).to(dev)
    sk = cs.build_skinner(LBSkinner).to(dev)
    comp = CompositeDeformer([tr, sk])
    curve = fl.Intersect_Free_Curve(list(st['curves']), list(0.9 * st['curves']), NAMES).to(dev)
    with torch.no_grad():
        curve.scale.copy_(st['scale'])
        curve.nx_scale.copy_(st['nx_scale'])
    fake = types.SimpleNamespace(conf=conf, info={}, device=device, garment_size=2, garment_names=['upper', 'bottom'],
                                 garment_vs=[st['verts_u'], st['verts_b']], garment_fs=[st['faces_u'].long(), st['faces_b'].long()],
                                 garment_nets=sdfs, deformer=comp, sdfShrinkRadius=0.0, curves=True, large_pose=False, _allreduce=None,
                                 tmpBodyVs=st['body_v'], tmpBodyFs=st['body_f'].long(), inter_free_curve=curve, fl_names=list(NAMES),
                                 fl_extract={'upper': UPPER, 'bottom': BOTTOM}, _frag_cache={},
                                 dataset=types.SimpleNamespace(H=H, W=W, fl_weights=dict(WEIGHTS)))
    fake.get_grad_parameters = lambda fids, d: ([None, st['conds_u'], st['conds_b']], st['poses'], st['trans'], None)
    fake._gt_feature_lines = lambda fids: (st['gt'], st['fl_masks'])
    for name in ('_deform_garments', '_garment_fragments', 'fl_visible_by_body_zbuff', 'compute_fl_proj_loss'):
        setattr(fake, name, types.MethodType(getattr(HotLoop, name), fake))
    fake.fl_optimizer = torch.optim.AdamW(curve.parameters(), lr=1e-4)
    cams = RectifiedPerspectiveCameras(st['focal'], st['pp'], st['R'], st['T'], image_size=[(W, H)])
    HotLoop.project_2d_loss(fake, N, torch.arange(N, device=dev), RATIO, cams)
    worst = {}

    def close(name, got, want, rt):
        want = want.to(torch.float32)
        got = torch.as_tensor(got).detach().cpu().to(torch.float32).reshape(want.shape)
        scale = max(float(want.abs().max()), 1e-12)
        worst[name] = float((got - want).abs().max()) / scale
        assert torch.allclose(got, want, rtol=rt, atol=rt * scale), (name, worst[name])

    info = fake.info['fl_loss']
    close('total', info['total'], g['total'], rtol)
    for mine, theirs in (('upper', 'short_sleeve_upper'), ('bottom', 'long_pants')):
        close(mine + ' project loss', info['%s_project loss' % mine], g['proj_' + theirs], rtol)
        close(mine + ' curve sdf', info['pc_%s_loss_sdf' % mine], g['sdf_' + theirs], rtol)
    close('d loss / d scale', curve.scale.grad, g['g_scale'], rtol_grad)
    close('d loss / d nx_scale', curve.nx_scale.grad, g['g_nx'], rtol_grad)
    close('scale after the AdamW step', curve.scale, g['new_scale'], 1e-5)
    close('nx_scale after the AdamW step', curve.nx_scale, g['new_nx'], 1e-4)
    return worst
