"""The whole-iteration case shared by tests/golden/make_golden_forward.py (the REFERENCE's OptimGarmentNetwork.forward +
backward + propagateTmpPsGrad) and the tests (recmv's HotLoop): the state of tests/project2d_case.py (two explicit garment
meshes, body, six curves, camera, 2-D feature lines) on a 40-frame sequence, plus images, normal maps, garment masks and
per-frame colour codes."""
import numpy as np
import torch

import project2d_case as pc

N, H, W, F = pc.N, pc.H, pc.W, 40
FRAME_IDS = [0, 17, 39]
SAMPLE_PIX = 40
RADIUS, K = 0.06, 50
SEED = 123
TR_KEYS = ["lin0.weight", "lin2.bias", "lin4.weight"]
SDF_KEYS = ["lin0.weight_v", "lin4.weight_g", "lin8.bias", "lin8.weight_v"]
RN_KEYS = ["lin0.weight_v", "lin4.bias"]
ROWS = 24
BOX = ((-0.8, -1.1, -0.6), (0.8, 1.1, 0.6))
RESOLUTIONS = [(9, 11, 7), (17, 21, 13)]
BODY_BIAS = 0.5
# the trajectory case (row g: north_star's "canonical-mesh Chamfer within 1e-4 of reference"): TRAJ_ITERS optimiser iterations from
# the fixture's state, the scheduled re-mesh at forward_time 30 inside, canonical extraction on a finer pyramid at the end
TRAJ_ITERS = 35
TRAJ_SHORT_ITERS, TRAJ_SHORT_REMESH = 14, 10     # the same structure inside the window where the reference agrees with itself
TRAJ_LR = 2e-5            # a fifth of the reference's train.learning_rate: on this 64 x 64 scene with freshly initialised nets Adam at 1e-4 moves the
# canonical surfaces by 0.1 in 35 steps and two runs of the REFERENCE ITSELF (4 vs 1 sgemm threads) end 2.4e-4 apart in Chamfer;
# at 2e-5 the reference's own envelope is a few 1e-5 — below the north_star's 1e-4 — while the surfaces still move by ~100x that
TRAJ_CANONICAL_RES = [(9, 11, 7), (17, 21, 13), (33, 41, 25), (65, 81, 49)]
# trajectory_c2: the same 14 iterations with the re-mesh on BASELINE configs[1]'s own coarse pyramid (train.py:42-47) — ~1e5 vertices
# per garment instead of ~1e2 — every TRAJ_C2_REMESH iterations (forward_time starts at 1: iterations 4 and 9), canonical extraction
# on the same pyramid
TRAJ_C2_RES = [(15, 21, 9), (29, 41, 17), (57, 81, 33), (113, 161, 65), (225, 321, 129)]
TRAJ_C2_ITERS, TRAJ_C2_REMESH = 14, 5


def trajectory_frames(it):
    """The three frames of iteration `it` (a fixed schedule over the 40-frame sequence; the images stay the fixture's)."""
    return [(7 * it) % F, (11 * it + 3) % F, (13 * it + 5) % F]


GARMENT_TYPE = 'male-1-casual'      # TEMPLATE_GARMENT: short_sleeve_upper + long_pants, the garment names the fixture was made with


def state():
    st = pc.state()
    g = torch.Generator().manual_seed(141)
    st['poses_all'] = 0.15 * torch.randn(F, 24, 3, generator=g)
    st['trans_all'] = 0.02 * torch.randn(F, 3, generator=g)
    st['cu_all'] = 0.1 * torch.randn(F, 128, generator=g)
    st['cb_all'] = 0.1 * torch.randn(F, 128, generator=g)
    st['rend_all'] = 0.1 * torch.randn(F, 256, generator=g)
    st['img'] = torch.rand(N, H, W, 3, generator=g) * 2 - 1
    st['normal'] = torch.nn.functional.normalize(torch.randn(N, H, W, 3, generator=g), dim=-1)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    st['gt_u'] = torch.stack([(((yy - 22 - i) / 13) ** 2 + ((xx - 24 + i) / 12) ** 2 < 1).float() for i in range(N)])
    st['gt_b'] = torch.stack([(((yy - 43 + i) / 12) ** 2 + ((xx - 23 - i) / 11) ** 2 < 1).float() for i in range(N)])
    for k in ('conds_u', 'conds_b', 'poses', 'trans'):
        del st[k]
    return st


SINGLE_TYPE = 'leyang_jump'          # TEMPLATE_GARMENT: ['dress'], a one-piece garment (train.is_upper_bottom = True)
SINGLE_LINES = ['neck', 'left_cuff', 'right_cuff', 'bottom_curve']        # FL_INFOS['leyang_jump'] = FL_EXTRACT['dress']
SINGLE_WEIGHTS = {'neck': 1.0, 'left_cuff': 2.0, 'right_cuff': 0.5, 'bottom_curve': 1.5}


def single_state():
    """The one-garment case: the upper mesh / net / code of state(), its four lines (the waist ring as the hem)."""
    st = state()
    L, M = len(SINGLE_LINES), pc.M
    st['curves'], st['scale'], st['nx_scale'] = st['curves'][:L], st['scale'][:L], st['nx_scale'][:L]
    st['gt'], st['fl_masks'] = st['gt'][:, :L * M], st['fl_masks'][:, :L]
    return st


# ----------------------------------------------------------------------------------------------- recmv side
class CaseDataset:
    """What HotLoop asks of a caller's dataset (recmv/model/network.py getOptNet), over the fixture's tensors."""

    video_segmented_index = []

    def __init__(self, st, device, single=False):
        leaf = lambda t: t.detach().clone().to(device).requires_grad_(True)
        self.F = self.frame_num = F
        self.H, self.W = H, W
        self.garment_type = SINGLE_TYPE if single else GARMENT_TYPE
        self.poses, self.trans = leaf(st['poses_all']), leaf(st['trans_all'])
        self.dcond = leaf(torch.cat([torch.zeros(F, 128), st['cu_all'].cpu()] + ([] if single else [st['cb_all'].cpu()]), dim=1))
        self.rend = leaf(st['rend_all'])
        self.conds = [self.dcond, self.rend]
        self.focal, self.pp, self.T = leaf(st['focal']), leaf(st['pp']), leaf(st['T'])
        self.R = st['R'].to(device)
        self.camera_params = {'focal_length': self.focal, 'princeple_points': self.pp, 'world2cam_coord_trans': self.T}
        self.fl_weights = dict(SINGLE_WEIGHTS if single else pc.WEIGHTS)
        self.shape = torch.zeros(10)

    def __len__(self):
        return F

    def get_grad_parameters(self, fids, device):
        return self.poses[fids], self.trans[fids], self.dcond[fids], self.rend[fids]

    def get_camera_parameters(self, n, device):
        return self.focal.expand(n, 2), self.pp.expand(n, 2), self.R.expand(n, 3, 3), self.T.expand(n, 3), H, W

    def get_batchframe_data(self, name, fids, batchsize):
        from recmv.dataset import SceneDataset
        return SceneDataset.get_batchframe_data(self, name, fids, batchsize)

    def learnable_weights(self):
        return [self.poses, self.trans, self.dcond, self.rend, self.focal, self.pp, self.T]


def build(g, device, large_pose=False, inputs=None, remesh=False, single=False, trajectory=False, lr=1e-3):
    """recmv's loop on the fixture's state: the facade object getOptNet returns with the fixture's networks, meshes, curves, camera
    and per-frame tensors put in (see run() for the switches; `trajectory`: explicit meshes given AND a body net for the scheduled
    re-mesh).  Returns (optNet, dataset, optimizer, frame_ids, modules) with modules = (sdfs, translator, skinner, colour net, curve)."""
    from pathlib import Path
    import numpy as np
    import common_setup as cs
    import mask_loss_case as mlc
    from recmv import curves as fl
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop, dct_nullspace
    from recmv.model import CompositeDeformer, LBSkinner, MLPTranslator, RenderingNetwork_view_norm, getTmpSdf
    from recmv.model.network import getOptNet
    repo = Path(__file__).resolve().parent.parent
    conf = ConfigFactory.parse_file(str(repo / "configs" / "synthetic" / "people_snapshot_like.conf"))
    conf.put('train.garment_type', SINGLE_TYPE if single else GARMENT_TYPE)
    conf.put('train.is_upper_bottom', bool(single))
    dev = torch.device(device)
    st = {k[3:]: v.to(dev) for k, v in (inputs if inputs is not None else g).items() if k.startswith('in_')}
    ds = CaseDataset(st, dev, single=single)
    res = [tuple(int(v) for v in r) for r in g['resolutions'].tolist()] if 'resolutions' in g else RESOLUTIONS      # (trajectory_c2)
    optNet, _ = getOptNet(ds, None, N, (-0.8, -1.1, -0.6), (0.8, 1.1, 0.6), res, device, conf, curves=False,
                          skin_grid=(5, 9, 7), opt_large=large_pose)
    assert optNet.large_pose == large_pose
    leaf = lambda t: t.detach().clone().requires_grad_(True)
    sdfs = [n.to(dev) for n in mlc.build_sdfs(getTmpSdf)][:1 if single else 2]
    tr = cs.build_translator(MLPTranslator).to(dev)
    sk = cs.build_skinner(LBSkinner).to(dev)
    rn = cs.build_render(RenderingNetwork_view_norm).to(dev)
    optNet.garment_nets = torch.nn.ModuleList(sdfs)
    if large_pose:
        optNet.freeze_sdf()
    optNet.deformer = CompositeDeformer([tr, sk])
    optNet.netRender = rn
    optNet.tmpBodyVs, optNet.tmpBodyFs = st['body_v'], st['body_f'].long()
    if remesh or trajectory:
        torch.manual_seed(520)
        optNet.sdf = cs.perturb(getTmpSdf("cpu", 6, bias=BODY_BIAS), 502, 0.003).to(dev)
        optNet.body_vs = optNet.body_fs = None
        assert [tuple(r) for r in optNet.engine.resolutions.tolist()] == res
    if not remesh:
        verts = [leaf(st['verts_u']), leaf(st['verts_b'])][:len(sdfs)]
        optNet.garment_vs, optNet.garment_fs = verts, [st['faces_u'].long(), st['faces_b'].long()][:len(sdfs)]
        optNet.body_vs, optNet.body_fs = st['body_v'], st['body_f'].long()
        optNet.garment_optimizer = torch.optim.SGD(verts, lr=0.05, momentum=0.9)
    line_names = list(SINGLE_LINES if single else pc.NAMES)
    curve = fl.Intersect_Free_Curve(list(st['curves']), list(0.9 * st['curves']), line_names).to(dev)
    with torch.no_grad():
        curve.scale.copy_(st['scale'])
        curve.nx_scale.copy_(st['nx_scale'])
    optNet.inter_free_curve, optNet.fl_names, optNet.curves = curve, line_names, True
    optNet.fl_extract, names = optNet._feature_line_tables()
    if single:
        assert optNet.garment_names == ['dress'] and optNet.mask_keys == ['upper_bottom'] and optNet.is_upper_bottom
        assert names == line_names and optNet.fl_extract == {'dress': line_names}
    else:
        assert optNet.garment_names == ['short_sleeve_upper', 'long_pants'] and optNet.mask_keys == ['upper', 'bottom']
        assert names == list(pc.NAMES) and optNet.fl_extract == {'short_sleeve_upper': pc.UPPER, 'long_pants': pc.BOTTOM}
    optNet.fl_optimizer = torch.optim.AdamW(curve.parameters(), lr=1e-4)
    optNet.forward_time, optNet.remesh_intersect, optNet.pc_radius, optNet.sample_pix = (0 if remesh else 1), 30, RADIUS, SAMPLE_PIX
    optNet.angThred = optNet._cameras().angThreshold(0.5)
    optNet.dctnull = dct_nullspace(30, 10, dev)
    optNet._datas = dict(img=st['img'], normal=st['normal'], fl_pts=st['gt'], fl_masks=st['fl_masks'], upper=st['gt_u'],
                         bottom=st['gt_b'])
    if single:
        optNet._datas = dict(img=st['img'], normal=st['normal'], fl_pts=st['gt'], fl_masks=st['fl_masks'], upper_bottom=st['gt_u'])
    cs.TrimeshStandIn.rng = np.random.RandomState(SEED)

    def sampler(verts_, faces_, n):
        pts = cs.TrimeshStandIn(verts_.detach().cpu().numpy(), faces_.cpu().numpy()).sample(n)
        return torch.from_numpy(pts).float().to(dev)

    optNet.curve_aware_loss = lambda ratio: HotLoop.curve_aware_loss(optNet, ratio, sampler=sampler)
    opt = optNet.rebuild_optimizer(lr=lr)
    frame_ids = torch.tensor(FRAME_IDS, device=dev)
    return optNet, ds, opt, frame_ids, (sdfs, tr, sk, rn, curve)


def run(g, device, rtol=1e-3, rtol_grad=1e-2, large_pose=False, inputs=None, remesh=False, rtol_loss=None, rtol_cam=None, single=False):
    """One whole iteration of recmv's loop — HotLoop.forward, backward, propagateTmpPsGrad — on the fixture's state against
    what the reference's forward / backward / propagateTmpPsGrad produced; returns the largest relative deviations.
    `large_pose`: the large-pose stage on both sides (OptimGarmentNetwork_LargePose: SDF nets frozen, curve terms zero-weighted);
    `inputs`: the fixture that holds the `in_*` state when `g` has outputs only; `remesh`: the iteration starts with the re-mesh
    (forward_time = 0: Seg3dLossless pyramid + MC of the body net and both garment nets) instead of given explicit meshes;
    `single`: the one-piece-garment case (`leyang_jump` = ['dress'], train.is_upper_bottom: single_state()).
    Tolerances (relative to the largest reference entry of each tensor): `rtol_loss` for the total loss (default `rtol`), `rtol`
    for the per-term info values and the stepped vertices, `rtol_grad` for the gradients the main optimiser consumes, `rtol_cam`
    (default `rtol_grad`) for the two camera-intrinsic gradients — sums of thousands of signed per-ray terms that cancel to a
    few per cent of their magnitude."""
    from recmv.loop import HotLoop
    optNet, ds, opt, frame_ids, (sdfs, tr, sk, rn, curve) = build(g, device, large_pose, inputs, remesh, single)
    verts = optNet.garment_vs
    torch.manual_seed(SEED)
    loss = HotLoop.forward(optNet, frame_ids, pc.RATIO, global_optimizer=opt)
    loss.backward()
    optNet.propagateTmpPsGrad(frame_ids, pc.RATIO)
    worst = {}

    def close(name, got, want, rt):
        want = want.to(torch.float32)
        got = torch.as_tensor(got).detach().cpu().to(torch.float32).reshape(want.shape)
        scale = max(float(want.abs().max()), 1e-12)
        worst[name] = float((got - want).abs().max()) / scale
        assert torch.allclose(got, want, rtol=rt, atol=rt * scale), (name, worst[name])

    if remesh:
        verts = optNet.garment_vs
        assert torch.equal(optNet.garment_fs[0].cpu(), g['faces_u'].long()) and torch.equal(optNet.garment_fs[1].cpu(), g['faces_b'].long())
        assert torch.equal(optNet.body_fs.cpu(), g['body_f'].long())
        close('re-meshed body vertices', optNet.body_vs, g['body_v'], 1e-4)      # (interpolated along edges from f32 SDF values)
    close('loss', loss, g['loss'], rtol if rtol_loss is None else rtol_loss)
    info = optNet.info
    ref_name = {n: n for n in optNet.garment_names}           # (the loop's info is keyed on the reference's garment names)
    for mine, theirs in ref_name.items():
        for key_mine, key_ref in (('%s_grad_loss', 'info_%s_grad_loss'), ('def_%s_loss', 'info_def_%s_loss'),
                                  ('%s_color_loss', 'info_%s_color_loss'), ('%s_normal_loss', 'info_%s_normal_loss'),
                                  ('pc_%s_loss_sdf', 'info_pc_%s_loss_sdf'), ('pc_%s_mask_loss', 'info_pc_loss__%s_mask_loss')):
            close(key_mine % mine, info[key_mine % mine], g[key_ref % theirs], rtol)
        assert tuple(int(v) for v in info['%s_invInfo' % mine]) == tuple(int(v) for v in g['info_%s_invInfo' % theirs]), mine
        close('fl %s project loss' % mine, info['fl_loss']['%s_project loss' % mine], g['info_fl_loss__%s_project_loss' % theirs], rtol)
    want_rays = [tuple(int(v) for v in g['info_%s_rayInfo' % n]) for n in ref_name.values()]     # (entering, converged) per garment
    assert int(info['rays_total']) == sum(r[0] for r in want_rays), (info['rays_total'], want_rays)
    assert [int(v) for v in info['rays_converged']] == [r[1] for r in want_rays], (info['rays_converged'], want_rays)
    close('dct_loss', info['dct_loss'], g['info_dct_loss'], rtol)
    if single:                           # no waist line, not a CURVE_AWARE capture: no disc term on either side (:794, :816)
        assert not any('circle_loss' in k for k in info) and not any('circle_loss' in k for k in g)
    else:
        close('curve-aware disc', info['pc_upper_bottom_circle_loss_sdf'], g['info_pc_upper_bottom_circle_loss_sdf'], rtol)
        close('new_verts_b', verts[1], g['new_verts_b'], rtol)
    close('new_verts_u', verts[0], g['new_verts_u'], rtol)
    close('curve scale after its step', curve.scale, g['new_scale'], 1e-5)
    close('curve nx_scale after its step', curve.nx_scale, g['new_nx'], 1e-4)
    tp, rp = dict(tr.named_parameters()), dict(rn.named_parameters())
    for k in TR_KEYS:
        close('g_tr_' + k, tp[k].grad[:ROWS], g['g_tr_' + k.replace('.', '_')], rtol_grad)
    for k in RN_KEYS:
        close('g_rn_' + k, rp[k].grad[:ROWS], g['g_rn_' + k.replace('.', '_')], rtol_grad)
    for i, net in enumerate(sdfs):
        sp = dict(net.named_parameters())
        for k in SDF_KEYS:
            if large_pose:
                assert sp[k].grad is None, "frozen SDF nets receive no gradient"
            else:
                close('g_sdf%d_%s' % (i, k), sp[k].grad[:ROWS], g['g_sdf%d_' % i + k.replace('.', '_')], rtol_grad)
    zero = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)
    close('g_poses', zero(ds.poses), g['g_poses_all'], rtol_grad)
    close('g_trans', zero(ds.trans), g['g_trans_all'], rtol_grad)
    close('g_cond_upper', zero(ds.dcond)[:, 128:256], g['g_cu_all'], rtol_grad)
    if not single:
        close('g_cond_bottom', zero(ds.dcond)[:, 256:], g['g_cb_all'], rtol_grad)
    assert float(zero(ds.dcond)[:, :128].abs().max()) == 0.0          # (the body's slot of the code is never used by the loop)
    close('g_rendcond', zero(ds.rend), g['g_rend_all'], rtol_grad)
    close('g_focal', zero(ds.focal), g['g_focal'], rtol_grad if rtol_cam is None else rtol_cam)
    close('g_pp', zero(ds.pp), g['g_pp'], rtol_grad if rtol_cam is None else rtol_cam)
    close('g_T', zero(ds.T), g['g_T'], rtol_grad)
    if 'loss2' in g:
        # a second iteration after the main optimiser's step: SGD momentum of the explicit vertices, AdamW state of the curves and
        # forward_time carry over (train.py:317-328)
        opt.step()
        opt.zero_grad()
        loss2 = HotLoop.forward(optNet, frame_ids, pc.RATIO, global_optimizer=opt)
        close('loss of the second iteration', loss2, g['loss2'], 20 * rtol)
        want2 = [int(v) for v in g['rays2']]
        got2 = [optNet.info['rays_total'], *[int(v) for v in optNet.info['rays_converged']]]
        assert got2[0] == want2[0] + want2[2] and abs(got2[1] - want2[1]) <= 2 and abs(got2[2] - want2[3]) <= 2, (got2, want2)
    return worst


def chamfer_vertices(a, b, chunk=2048):
    """Symmetric Chamfer distance between two vertex sets in pytorch3d's convention (mean over points of the SQUARED distance to
    the nearest neighbour, both directions summed) and the mean UNsquared nearest-neighbour distance, in float64 (rows of the
    distance matrix in chunks: the C2-sized meshes have ~1e5 vertices)."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()

    def nearest(p, q):
        if p.shape[0] * q.shape[0] <= 1 << 24:                      # small sets: the plain difference form
            return ((p[:, None, :] - q[None, :, :]) ** 2).sum(-1).min(1).values
        out, q2 = [], (q * q).sum(1)
        for i in range(0, p.shape[0], chunk):                       # |p|^2 + |q|^2 - 2 p.q in float64, then the exact difference at the argmin
            pc_ = p[i:i + chunk]
            j = ((pc_ * pc_).sum(1, keepdim=True) + q2[None, :] - 2.0 * pc_ @ q.t()).argmin(1)
            out.append(((pc_ - q[j]) ** 2).sum(1))
        return torch.cat(out)
    ab, ba = nearest(a, b), nearest(b, a)
    return float(ab.mean() + ba.mean()), float(0.5 * (ab.sqrt().mean() + ba.sqrt().mean()))


def run_trajectory(g, inputs, device, iters=None):
    """Row (g) of the scope table — north_star's "canonical-mesh Chamfer within 1e-4 of reference": recmv's loop for TRAJ_ITERS
    optimiser iterations in train.py's order (train.py:317-328) from the fixture's state, the scheduled re-mesh (forward_time 30)
    inside, then the canonical meshes of the body net and both garment nets on the finer pyramid — against the reference's own loop
    run the same way (tests/golden/make_golden_forward.py trajectory -> trajectory.npz).  Returns a dict of what was measured; the
    caller asserts."""
    from recmv.MCAcc import Seg3dLossless
    from recmv.loop import HotLoop
    T = int(g['losses'].shape[0]) if iters is None else iters
    optNet, ds, opt, _, (sdfs, tr, sk, rn, curve) = build(g, device, inputs=inputs, trajectory=True,
                                                          lr=float(g['lr']) if 'lr' in g else TRAJ_LR)
    optNet.remesh_intersect = int(g['remesh_period']) if 'remesh_period' in g else 30
    dev = torch.device(device)
    losses, rays, verts_n, faces_equal = [], [], [], None
    for it in range(T):
        fids = torch.tensor(trajectory_frames(it), device=dev)
        torch.manual_seed(SEED + it)
        opt.zero_grad()
        remesh_now = optNet.forward_time % optNet.remesh_intersect == 0
        loss = HotLoop.forward(optNet, fids, pc.RATIO, global_optimizer=opt)
        if loss.requires_grad:
            loss.backward()
        optNet.propagateTmpPsGrad(fids, pc.RATIO)
        opt.step()
        losses.append(float(loss.detach()))
        conv = [int(v) for v in optNet.info['rays_converged']]
        rays.append((int(optNet.info['rays_total']), *conv))
        verts_n.append([int(v.shape[0]) for v in optNet.garment_vs])
        if remesh_now:
            faces_equal = [tuple(f.shape) == tuple(g[k].shape) and bool(torch.equal(f.cpu(), g[k].long()))
                           for f, k in zip(optNet.garment_fs, ('final_faces_u', 'final_faces_b'))]
    out = dict(losses=losses, rays=rays, verts_n=verts_n, remesh_faces_equal=faces_equal)
    ref_l = g['losses'].double()
    out['loss_rel_dev'] = [abs(a - float(b)) / max(abs(float(b)), 1e-12) for a, b in zip(losses, ref_l)]
    ref_rays = g['rays']                                         # per garment (entering, converged)
    out['rays_ref'] = [(int(r[0] + r[2]), int(r[1]), int(r[3])) for r in ref_rays[:T]]
    if T == int(g['losses'].shape[0]):
        canon_res = ([tuple(int(v) for v in r) for r in g['canonical_res'].tolist()] if 'canonical_res' in g else TRAJ_CANONICAL_RES)
        fine = Seg3dLossless(query_func=None, b_min=list(BOX[0]), b_max=list(BOX[1]), resolutions=canon_res,
                             align_corners=False, balance_value=0.0, use_cuda_impl=dev.type == 'cuda', faster=False).to(dev)
        vs, fs = optNet.discretizeSDF(pc.RATIO, fine, 0.)
        out['canon_verts'] = {tag: v.detach().cpu() for tag, v in zip(('body', 'u', 'b'), vs)}
        for tag, v, f in zip(('body', 'u', 'b'), vs, fs):
            sq, lin = chamfer_vertices(v, g['canon_v_' + tag])
            out['canon_%s' % tag] = dict(chamfer_sq=sq, mean_dist=lin, moved_sq=float(g['canon_moved_' + tag][0]),
                                         moved_dist=float(g['canon_moved_' + tag][1]), verts=(int(v.shape[0]), int(g['canon_v_' + tag].shape[0])),
                                         faces_equal=tuple(f.shape) == tuple(g['canon_f_' + tag].shape)
                                         and bool(torch.equal(f.cpu(), g['canon_f_' + tag].long())))
        for tag, v in zip(('u', 'b'), optNet.garment_vs):
            ref_v = g['final_verts_' + tag]
            sq, lin = chamfer_vertices(v, ref_v)
            same = tuple(v.shape) == tuple(ref_v.shape)
            out['explicit_%s' % tag] = dict(chamfer_sq=sq, mean_dist=lin,
                                            max_abs_dev=float((v.detach().cpu() - ref_v).abs().max()) if same else None)
    return out


TRAJ_HEAD = 8          # iterations over which the reference's two runs (4 vs 1 sgemm threads) still agree to <= 2e-5 on the loss


def check_trajectory(out, g, other_arithmetic=False):
    """Assertions on run_trajectory()'s result.  What is asked of the implementation is tied to what the REFERENCE does against
    itself under another summation order (the `self_*` entries of a fixture: its loop run with 4 and with 1 sgemm threads) — the
    optimisation is a chaotic map (Adam on 2 M parameters, rays entering / leaving the converged set), rounding differences grow by
    orders of magnitude over tens of iterations, for the reference's own two runs as for anybody else's:
      * while the reference's two runs agree with each other (loss <= 3e-4 apart, identical ray counts: the whole of
        trajectory_short.npz, the first 18 iterations of trajectory.npz) this implementation agrees with the reference to
        rounding in the typical iteration (median loss deviation <= 1e-5) and to a threshold ray otherwise (max <= 2e-2, converged-
        ray counts within 2), with the same rays entering per iteration, and — where the re-mesh falls inside that window
        and the reference's runs extract identical faces — with bit-identical faces;
      * at the end of a whole run the north_star's acceptance number: symmetric Chamfer distance (pytorch3d convention: mean squared
        nearest-neighbour distance, both directions summed) between this implementation's canonical meshes and the reference's,
        body and both garments: <= 1e-4 where the reference's own two runs end <= 1e-4 / 3 apart, and <= 3x the reference's own
        distance otherwise (never above 1e-3); the loss curve stays inside 3x the reference's own running envelope."""
    if other_arithmetic:
        return _check_trajectory_device(out, g)
    n = len(out['losses'])
    total = int(g['losses'].shape[0])
    self_dev = g['self_loss_rel_dev'].double()
    rays_eq = [bool(v) for v in g['self_rays_equal']]
    agree = 0                                            # the window in which the reference agrees with itself
    while agree < total and float(self_dev[agree]) <= 3e-4 and rays_eq[agree]:
        agree += 1
    head = min(n, agree)
    assert head >= min(n, TRAJ_HEAD), ("fixture: the reference's own runs must agree over the head", agree)
    dev = out['loss_rel_dev']
    # inside the window: the typical iteration agrees to rounding (median <= 1e-5); single iterations may be off by a ray that sits
    # on the root finder's stopping threshold and converges on one side only — with ~10 converged rays for the upper garment of
    # this scene one such ray moves the colour / normal terms by 2 % and the total by 3e-3 (measured at the re-mesh iteration of the
    # short run, every other info entry equal to 6 digits) — hence max <= 2e-2 and converged-ray counts within 2
    window = sorted(dev[:head])
    assert window[len(window) // 2] <= 1e-5 and window[-1] <= 2e-2, ("loss inside the reference's own agreement window", dev[:head])
    for mine, ref in zip(out['rays'][:head], out['rays_ref'][:head]):
        assert mine[0] == ref[0] and all(abs(a - b) <= 2 for a, b in zip(mine[1:], ref[1:])), (out['rays'][:head], out['rays_ref'][:head])
    report = {"agreement_window": agree, "loss_rel_dev_median_in_window": window[len(window) // 2], "loss_rel_dev_max_in_window": window[-1]}
    if n < total:
        return report
    env = self_dev.clone()
    for i in range(1, len(env)):                        # running maximum: the envelope only widens
        env[i] = max(float(env[i]), float(env[i - 1]))
    for i, d in enumerate(dev):                         # (3 iterations of slack: when the first threshold ray flips is itself chance)
        e = float(env[min(i + 3, len(env) - 1)])
        assert d <= min(max(3 * e, 2e-2), 0.5), ("loss curve outside 3x the reference's own envelope", i, d, e)
    period = int(g['remesh_period']) if 'remesh_period' in g else 30
    ref_faces_eq = [bool(g['self_faces_equal_' + t]) for t in ('u', 'b')]
    if period - 1 < agree and all(ref_faces_eq):          # the re-mesh (iteration period - 1) inside the window
        assert out['remesh_faces_equal'] == [True, True], "faces of the scheduled re-mesh"
    for tag in ('body', 'u', 'b'):
        c = out['canon_' + tag]
        ref_self = float(g['self_canon_chamfer_' + tag][0])
        bound = 1e-4 if 3 * ref_self <= 1e-4 else min(3 * ref_self, 1e-3)
        assert c['chamfer_sq'] <= bound, ("canonical-mesh Chamfer", tag, c, ref_self)
        report['canon_' + tag] = dict(chamfer_sq=c['chamfer_sq'], bound=bound, mean_dist=c['mean_dist'], moved_sq=c['moved_sq'],
                                      reference_vs_itself=ref_self, verts=c['verts'], faces_equal=c['faces_equal'])
    report['loss_rel_dev_max'] = max(dev)
    report['reference_self_loss_rel_dev_max'] = float(self_dev.max())
    report['remesh_faces_equal'] = out['remesh_faces_equal']
    report['reference_self_remesh_faces_equal'] = ref_faces_eq
    report['explicit'] = {t: out['explicit_' + t] for t in ('u', 'b')}
    return report


def reference_envelope():
    """tests/golden/trajectory_envelope.npz (make_golden_envelope.py): all pairwise canonical-mesh Chamfer distances between six runs
    of the reference's own 35-iteration loop that differ only in the number of sgemm threads."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory_envelope.npz")
    if not os.path.isfile(path):
        return None
    return dict(np.load(path))


def disturbed_envelope():
    """tests/golden/trajectory_perturbed.npz (make_golden_perturbed.py): canonical-mesh Chamfer distances between the reference's
    35-iteration run and runs of the SAME reference loop whose matrix products' results carry a relative error of 1 .. 16 f32 ulp."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory_perturbed.npz")
    if not os.path.isfile(path):
        return None
    return dict(np.load(path))


def _check_trajectory_device(out, g):
    """The same run on the DEVICE, whose matrix products (f32 MFMA fma chains) round differently from the torch-CPU sgemm the
    reference and the CPU port share: the first iteration differs by 1e-6 on the loss instead of 1e-8, and the map amplifies a
    difference by ~2.5x per iteration on this scene (measured on the MI355X: 1.2e-6, 2.6e-6, 5.9e-6, 1.1e-5, 4.7e-5, 8e-5, 3e-4 ...;
    the reference's own two runs go 0, 0, ..., 1e-6 at iteration 4, 3e-4 at 11, 6e-2 at 18 from their smaller start).  Asserted:
      * the first four iterations agree to 1e-4 on the loss and every iteration has the reference's number of rays ENTERING the
        root finder as long as the reference's two runs agree on it (the rasterised surface pixels and the seeded subsets are the
        same); converged-ray counts within 8;
      * trajectory_short (14 iterations, re-mesh at the 10th — inside the horizon over which a 1e-6 difference stays small): the
        north_star's number, canonical-mesh Chamfer <= 1e-4 for the body and both garments, on surfaces that moved by more;
      * trajectory (35 iterations): past that horizon for ANY implementation that is not the reference's own binary — the reference
        against itself ends 0.65e-4 / 1.6e-4 apart; held to 20 % of the distance the surfaces moved, and to 1e-3."""
    n = len(out['losses'])
    total = int(g['losses'].shape[0])
    dev = out['loss_rel_dev']
    # the first four iterations to 1e-4 — or to three times what the reference's own two runs differ by there (at the config's own
    # learning rate, trajectory_lr.npz, a ray changes sides in the reference's second run at iteration 4: 8e-4 against itself)
    own = [float(v) for v in g['self_loss_rel_dev'][:4]] if 'self_loss_rel_dev' in g else [0.0] * 4
    assert all(d <= max(1e-4, 3.0 * o) for d, o in zip(dev[:4], own)), ("loss over the first four iterations", dev[:4], own)
    rays_eq = [bool(v) for v in g['self_rays_equal']]
    for i, (mine, ref) in enumerate(zip(out['rays'], out['rays_ref'])):
        if not rays_eq[i]:
            break
        if i < 9:
            assert mine[0] == ref[0], ("rays entering the root finder", i, mine, ref)
        assert abs(mine[0] - ref[0]) <= 8 and all(abs(a - b) <= 8 for a, b in zip(mine[1:], ref[1:])), (i, mine, ref)
    report = {"loss_rel_dev": ["%.1e" % d for d in dev]}
    if n < total:
        return report
    short = total <= TRAJ_SHORT_ITERS
    env = reference_envelope() if not short else None
    dis = disturbed_envelope() if not short and 'resolutions' not in g else None
    for tag in ('body', 'u', 'b'):
        c = out['canon_' + tag]
        ref_self = float(g['self_canon_chamfer_' + tag][0])
        # 14 iterations: the north_star's 1e-4.  35 iterations: the spread the REFERENCE'S OWN loop shows when only the summation order
        # of its matrix products changes — the largest of the 15 pairwise distances between six runs (1, 2, 3, 4, 6, 8 sgemm
        # threads; tests/golden/make_golden_envelope.py -> trajectory_envelope.npz: upper garment 5.9e-6 .. 6.5e-5, bottom 9.7e-6 ..
        # 5.2e-4, body 0).  Since round 6 the device is inside it for BOTH garments: until then the upper garment ended at 2.2e-4,
        # 3.5x outside — seeded by fma contraction in the kinematic-chain kernel (a coherent last-bit difference of the posed
        # skeleton against the reference's separately rounded operations, tools/trajectory_seeds.py), not by "chaos": with that
        # kernel un-contracted it ends at 1.4e-5 and the 14-iteration distance falls from 7.6e-6 to 7.1e-8.
        # `reference_with_disturbed_products_*` (make_golden_perturbed.py) stays reported for context.
        inside = None
        if short:
            bound = 1e-4
        elif env is not None:
            bound = max(float(env['canon_chamfer_' + tag].max()), 1e-12) if tag != 'body' else 1e-4
            inside = True
        else:
            bound = 1e-4 if tag == 'body' else min(5e-4, 0.2 * c['moved_sq'])
        if env is not None:
            inside = bool(c['chamfer_sq'] <= max(float(env['canon_chamfer_' + tag].max()), 1e-10))      # (1e-10: vertices 1e-5 apart — f32 evaluation of the untrained body net)
        assert c['chamfer_sq'] <= bound, ("canonical-mesh Chamfer", tag, c, bound)
        if short and tag != 'body' and 'resolutions' not in g:
            assert c['moved_sq'] > 2.5 * 1e-4, ("fixture: the surfaces move by more than the tolerance", tag, c)
        report['canon_' + tag] = dict(chamfer_sq=c['chamfer_sq'], bound=bound, mean_dist=c['mean_dist'], moved_sq=c['moved_sq'],
                                      reference_vs_itself=ref_self, verts=c['verts'], faces_equal=c['faces_equal'])
        if env is not None:
            d = env['canon_chamfer_' + tag]
            iu = d[np.triu_indices(d.shape[0], 1)]
            report['canon_' + tag].update(reference_envelope_min_median_max=[float(iu.min()), float(np.median(iu)), float(iu.max())],
                                          inside_reference_envelope=inside)
        if dis is not None:           # the reference's loop with rounding-sized errors on its products' results (not asserted: reported)
            d = dis['canon_chamfer_' + tag]
            report['canon_' + tag].update(reference_with_disturbed_products_min_median_max=[float(d.min()), float(np.median(d)), float(d.max())],
                                          inside_disturbed_products_range=bool(c['chamfer_sq'] <= max(float(d.max()), 1e-10)))
    report['remesh_faces_equal'] = out['remesh_faces_equal']
    report['explicit'] = {t: out['explicit_' + t] for t in ('u', 'b')}
    return report
