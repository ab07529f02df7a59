"""The mask_loss case shared by tests/golden/make_golden_mask_loss.py (reference method) and the tests (HotLoop.mask_loss on the
CPU port): two small explicit garment meshes, three frames of 40 x 32, a pinhole looking at them, ground-truth masks."""
import math

import torch

N, H, W, K, RADIUS = 3, 40, 32, 50, 0.06
TR_KEYS = ["lin0.weight", "lin2.bias", "lin4.weight"]
SDF_KEYS = ["lin0.weight_v", "lin8.bias", "lin8.weight_v"]
ROWS = 24                                  # leading rows of every compared parameter gradient kept in the fixture


def build_sdfs(getTmpSdf):
    import common_setup as cs
    nets = []
    for i, bias in enumerate((0.55, 0.45)):
        torch.manual_seed(510 + i)
        nets.append(cs.perturb(getTmpSdf("cpu", 6, bias=bias), 500 + i, 0.003))
    return nets


def _blob(n_lat, n_lon, radius, centre, seed):
    g = torch.Generator().manual_seed(seed)
    th = torch.linspace(0.15, math.pi - 0.15, n_lat)
    ph = torch.linspace(0, 2 * math.pi, n_lon + 1)[:-1]
    v = torch.stack([torch.sin(th)[:, None] * torch.cos(ph)[None], torch.cos(th)[:, None].expand(-1, n_lon),
                     torch.sin(th)[:, None] * torch.sin(ph)[None]], -1).reshape(-1, 3)
    v = v * radius * (1 + 0.03 * torch.randn(v.shape[0], 1, generator=g)) + torch.tensor(centre)
    faces = []
    for a in range(n_lat - 1):
        for b in range(n_lon):
            p, q = a * n_lon + b, a * n_lon + (b + 1) % n_lon
            faces += [[p, q, p + n_lon], [q, q + n_lon, p + n_lon]]
    return v.float(), torch.tensor(faces).long()


def state():
    import common_setup as cs
    g = torch.Generator().manual_seed(91)
    vu, fu = _blob(9, 14, 0.30, (0.0, 0.18, 0.0), 92)
    vb, fb = _blob(8, 12, 0.26, (0.0, -0.30, 0.0), 93)
    conds_u, _ = cs.conds_and_inds(8, nframes=N, condlen=128, seed=4)
    conds_b, _ = cs.conds_and_inds(8, nframes=N, condlen=128, seed=5)
    poses, trans = cs.poses_trans(N, seed=7)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    gt_u = torch.stack([(((yy - 14 - i) / 9) ** 2 + ((xx - 16 + i) / 8) ** 2 < 1).float() for i in range(N)])
    gt_b = torch.stack([(((yy - 27 + i) / 8) ** 2 + ((xx - 15 - i) / 7) ** 2 < 1).float() for i in range(N)])
    return dict(verts_u=vu, faces_u=fu, verts_b=vb, faces_b=fb, conds_u=conds_u.detach(), conds_b=conds_b.detach(),
                poses=poses.detach(), trans=trans.detach(), gt_u=gt_u, gt_b=gt_b,
                focal=torch.tensor([[70., 72.]]), pp=torch.tensor([[16., 20.]]),
                R=torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3), T=torch.tensor([[0.02, -0.03, 2.6]]))


def run(g, device, rtol=2e-4, rtol_grad=3e-3):
    """HotLoop.mask_loss (+ pc_sdf_terms) on a stand-in `self` built from the fixture's inputs, against the reference method's
    outputs; returns the largest relative deviation per compared quantity."""
    import types
    import common_setup as cs
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    from recmv.model import CompositeDeformer, LBSkinner, MLPTranslator, RectifiedPerspectiveCameras, getTmpSdf
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    conf = ConfigFactory.parse_file(str(repo / "configs" / "synthetic" / "people_snapshot_like.conf")).get_config('loss_coarse')
    dev = torch.device(device)
    st = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith('in_')}
    sdfs = [n.to(dev) for n in build_sdfs(getTmpSdf)]
    tr = cs.build_translator(MLPTranslator).to(dev)
    sk = cs.build_skinner(LBSkinner).to(dev)
    comp = CompositeDeformer([tr, sk])
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    leaves = dict(conds_u=leaf(st['conds_u']), conds_b=leaf(st['conds_b']), poses=leaf(st['poses']), trans=leaf(st['trans']))
    verts = [leaf(st['verts_u']), leaf(st['verts_b'])]
    fake = types.SimpleNamespace(conf=conf, info={}, device=device, garment_size=2, garment_names=['upper', 'bottom'], garment_vs=verts,
                                 garment_fs=[st['faces_u'].long(), st['faces_b'].long()], garment_nets=sdfs, deformer=comp,
                                 sdfShrinkRadius=0.0, pc_radius=RADIUS, curves=False, _allreduce=None,
                                 dataset=types.SimpleNamespace(H=H, W=W))
    fake.get_grad_parameters = lambda fids, d: ([None, leaves['conds_u'], leaves['conds_b']], leaves['poses'], leaves['trans'], None)
    fake._gt_garment_mask = lambda g_i, fids: (st['gt_u'], st['gt_b'])[g_i]
    for name in ('_deform_garments', 'compute_garment_pc_loss', 'pc_sdf_terms', 'curve_aware_loss'):
        setattr(fake, name, types.MethodType(getattr(HotLoop, name), fake))
    fake._backward_early = HotLoop._backward_early
    fake.garment_optimizer = torch.optim.SGD(verts, lr=0.05, momentum=0.9)
    cams = RectifiedPerspectiveCameras(st['focal'], st['pp'], st['R'], st['T'], image_size=[(W, H)])
    ratio = {"sdfRatio": 0.8, "deformerRatio": 0.7, "renderRatio": 1.0}
    def_vs, pc_sdf = HotLoop.mask_loss(fake, N, torch.arange(N, device=dev), ratio, cams)
    assert not pc_sdf.requires_grad          # (the |SDF| terms are differentiated where they are formed: the reference's
                                            #  `loss.backward()` on them has already happened, HotLoop._backward_early)
    worst = {}

    def close(name, got, want, rt):
        want = want.to(torch.float32)
        got = got.detach().cpu().to(torch.float32).reshape(want.shape)
        scale = max(float(want.abs().max()), 1e-12)
        worst[name] = float((got - want).abs().max()) / scale
        assert torch.allclose(got, want, rtol=rt, atol=rt * scale), (name, worst[name])

    close('pc_sdf_loss', pc_sdf, g['pc_sdf_loss'], rtol)
    close('def_u', def_vs[0], g['def_u'], rtol)
    close('def_b', def_vs[1], g['def_b'], rtol)
    close('new_verts_u', verts[0], g['new_verts_u'], rtol)
    close('new_verts_b', verts[1], g['new_verts_b'], rtol)
    for mine, theirs in (('pc_upper_mask_loss', 'info_upper_mask'), ('pc_bottom_mask_loss', 'info_bottom_mask'),
                         ('pc_upper_loss_sdf', 'info_upper_sdf'), ('pc_bottom_loss_sdf', 'info_bottom_sdf')):
        close(mine, torch.as_tensor(fake.info[mine]), g[theirs], rtol)
    tp = dict(tr.named_parameters())
    for k in TR_KEYS:
        close('g_tr_' + k, tp[k].grad[:ROWS], g['g_tr_' + k.replace('.', '_')], rtol_grad)
    for k, v in leaves.items():
        close('g_' + k, v.grad if v.grad is not None else torch.zeros_like(v), g['g_' + k], rtol_grad)
    for i, net in enumerate(sdfs):
        sp = dict(net.named_parameters())
        for k in SDF_KEYS:
            close('g_sdf%d_%s' % (i, k), sp[k].grad[:ROWS], g['g_sdf%d_' % i + k.replace('.', '_')], rtol_grad)
    return worst
