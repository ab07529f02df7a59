"""The start-up stage that precedes the loop — feature-line template registration and the SDF pre-fit — against the
reference's own functions (tests/golden/make_golden_startup.py -> tests/golden/startup.npz).  CPU only: oracle/cpu_port
stands in for librecmv_hip.so; tests/test_gpu_startup.py runs the same drivers through the HIP library."""
import random
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent
for p in (str(REPO / "rec-mv_amd"), str(REPO), str(HERE), str(HERE / "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
import capture_fixture as cf  # noqa: E402
import startup_case as sc  # noqa: E402


def load():
    return {k: torch.from_numpy(v) for k, v in np.load(HERE / "golden" / "startup.npz").items()}


def test_feature_line_transforms_match_the_reference():
    """engineer/utils/matrix_transform.py: 6-D rotation (incl. nearly parallel axes) and the five per-line transforms
    (incl. a negative scale, clamped to zero)."""
    from recmv.engineer.utils import matrix_transform as mt
    g = load()
    lines = list(torch.split(g["mt_lines"], [int(n) for n in g["mt_split"]]))
    R = mt.compute_rotation_matrix_from_ortho6d(g["mt_poses"])
    torch.testing.assert_close(R, g["mt_R"], rtol=1e-6, atol=1e-7)
    T, S = g["mt_T"], g["mt_S"]
    cat = lambda lst: torch.cat(lst, 0)
    for name, got in (("icp", mt.icp_rotate_transfrom(lines, R, T)), ("scale_icp", mt.scale_icp_rotate_transfrom(lines, R, T, S)),
                      ("center", mt.center_transform(lines, R, T)), ("icp_center", mt.icp_rotate_center_transform(lines, R, T)),
                      ("scale_icp_center", mt.scale_icp_rotate_center_transform(lines, R, T, S))):
        torch.testing.assert_close(cat(got), g["mt_" + name], rtol=1e-5, atol=1e-6, msg=lambda m: name + ": " + m)
    # meshes and bare vertex tensors are interchangeable
    meshes = [mt.FeatureLineMesh(v, torch.zeros(1, 3, dtype=torch.long)) for v in lines]
    assert torch.equal(cat(mt.center_transform(meshes, R, T)), cat(mt.center_transform(lines, R, T)))
    moved = meshes[0].update_padded(lines[0][None] + 1.)
    assert torch.equal(moved.verts_packed(), lines[0] + 1.) and moved.faces_packed() is meshes[0].faces_packed()


def test_init_fl_dataset_matches_the_reference(tmp_path):
    """dataset/dataset.py `get_init_fl_datasets` / `Init_Fl_SceneDataset`: the supervising frames, their samples, the
    loader's order under the same seeds; PeopleSnapshot captures hand over the annotated frames only."""
    from recmv.dataset import Init_Fl_SceneDataset, People_Snapshot_SceneDataset, SceneDataset
    g = load()
    root = sc.write_capture(str(tmp_path))
    torch.manual_seed(31)
    ds = SceneDataset(root, dict(sc.CONDS), cf.GARMENT_TYPE, fl_sampling=sc.FL_SAMPLING, curve_sampling=1)
    random.seed(32)
    torch.manual_seed(32)
    loader = ds.get_init_fl_datasets(3, None, 0)
    init_ds = loader.dataset
    assert isinstance(init_ds, Init_Fl_SceneDataset)
    assert [float(i) for i in init_ds.idx] == g["init_idx"].tolist()
    assert [float(len(init_ds)), float(len(loader))] == g["init_len"].tolist()
    for k in (0, 4, 7):
        fid, sample = init_ds[k]
        assert float(fid) == float(g["init_%d_fid" % k])
        # (the reference's tensor is float64 when a line is missing from the frame — its zero filler is; the loop reads float32)
        torch.testing.assert_close(sample["fl_pts"], g["init_%d_fl_pts" % k].float(), rtol=1e-6, atol=1e-5)
        assert torch.equal(sample["fl_masks"].float(), g["init_%d_fl_masks" % k])
        assert torch.equal(sample["mask"], g["init_%d_mask" % k])
    assert [float(f) for fids, _ in loader for f in fids] == g["init_order"].tolist()
    ps = People_Snapshot_SceneDataset(root, dict(sc.CONDS), cf.GARMENT_TYPE, fl_sampling=sc.FL_SAMPLING, curve_sampling=1,
                                      a_pose=False)
    ps_init = ps.get_init_fl_datasets(2, None, 0).dataset
    assert [float(i) for i in ps_init.idx] == g["init_ps_idx"].tolist()
    assert torch.equal(torch.stack([ps_init[k][1]["fl_masks"].float() for k in range(len(ps_init))]), g["init_ps_masks"])


def test_feature_line_registration_matches_the_reference(tmp_path):
    """engineer/core/fl_optimizer.py `scale_rigid_optimizer` (:111-519: 50 + 10 + 50 epochs of Adam with the body z-buffer
    visibility test) and `rigid_optimizer` (:520-677) run for real vs recmv's on the same capture: stored transform,
    registered vertices, and the re-application of a stored file."""
    from oracle import cpu_port
    cpu_port.install()
    try:
        sc.run_registration(load(), sc.write_capture(str(tmp_path)), "cpu")
    finally:
        cpu_port.uninstall()


def test_sdf_prefit_matches_the_reference_method():
    """OptimGarmentNetwork.initializeSDF (:387-443) run for real on the reference's SDF net vs HotLoop.initializeSDF:
    parameters after three epochs, with and without the normal term."""
    from oracle import cpu_port
    cpu_port.install()
    try:
        sc.run_prefit(load(), "cpu")
    finally:
        cpu_port.uninstall()


def test_prefit_files_carry_the_reference_names_and_are_loaded_back(tmp_path):
    """train.py:179-206 + model/network.py:201-221: no initial_sdf_idr_*.pth -> getOptNet asks for the pre-fit (1200 epochs
    by default); train.prefit_sdf fits body + garment nets to the clouds of --init-points, writes the state dicts and the
    extracted meshes under the reference's names; the next getOptNet loads them and asks for nothing."""
    import argparse
    import os
    from oracle import cpu_port
    import train
    from recmv.dataset import getDatasetAndLoader
    from recmv.hocon import ConfigFactory
    from recmv.model.network import getOptNet
    from recmv.utils import read_ply
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    root = sc.write_capture(str(tmp_path))
    conds_lens = {'deformer': conf.get_int('mlp_deformer.condlen') * 3, 'renderer': conf.get_int('render_net.condlen')}
    torch.manual_seed(3)
    ds, _ = getDatasetAndLoader(root, conds_lens, 3, True, 0, True, True, conf.get_config('train.opt_camera'), cf.GARMENT_TYPE,
                                data_type='scene')
    g = torch.Generator().manual_seed(8)
    clouds = {}
    for name, r in (('body', 0.40), ('long_sleeve_upper', 0.50), ('long_pants', 0.45)):   # TEMPLATE_GARMENT['female-3-casual']
        d = torch.nn.functional.normalize(torch.randn(240, 3, generator=g), dim=1)
        clouds[name + '_vs'], clouds[name + '_ns'] = (d * r).numpy(), d.numpy()
    del clouds['long_pants_ns']                                            # a cloud without normals: fitted without the normal term
    np.savez(tmp_path / 'clouds.npz', **clouds)
    save_root = os.path.join(root, 'result')
    os.makedirs(save_root)
    cpu_port.install()
    try:
        res, box = [(9, 11, 7), (17, 21, 13)], ((-0.9, -1.2, -0.6), (0.9, 1.2, 0.6))
        make = lambda: getOptNet(ds, 'result', 3, box[0], box[1], res, 'cpu', conf, curves=False, skin_grid=(5, 9, 7))
        optNet, todo = make()
        assert todo == 1200
        before = [p.detach().clone() for p in optNet.garment_nets[0].parameters()]
        train.prefit_sdf(optNet, 2, conf, argparse.Namespace(init_points=None), save_root, 0)          # no clouds: untouched
        assert all(torch.equal(a, b) for a, b in zip(before, optNet.garment_nets[0].parameters()))
        train.prefit_sdf(optNet, 2, conf, argparse.Namespace(init_points=str(tmp_path / 'clouds.npz')), save_root, 0)
        names = ['initial_sdf_idr_6_0', 'initial_sdf_long_sleeve_upper_idr_6_0', 'initial_sdf_long_pants_idr_6_0']
        for n in names:
            assert os.path.isfile(os.path.join(save_root, n + '.pth')) and os.path.isfile(os.path.join(save_root, n + '.ply')), n
        assert not all(torch.equal(a, b) for a, b in zip(before, optNet.garment_nets[0].parameters()))
        again, todo = make()
        assert todo == -1
        for a, b in zip([optNet.sdf] + list(optNet.garment_nets), [again.sdf] + list(again.garment_nets)):
            assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
        v, f = read_ply(os.path.join(save_root, names[0] + '.ply'))
        assert torch.equal(again.tmp_sdf_body_vs, v) and torch.equal(again.tmp_sdf_face_vs, f.float())
        assert torch.equal(v, optNet.tmp_sdf_body_vs.cpu()) and int(f.max()) < v.shape[0] and f.shape[0] > 0
        # ... and travels in the checkpoint like the reference's registered buffers (OptimGarmentNetwork.py:176-178)
        sd = again.state_dict()
        assert torch.equal(sd['tmp_sdf_body_vs'], v) and torch.equal(sd['tmp_sdf_face_vs'], f.float())
        fresh, _ = getOptNet(ds, None, 3, box[0], box[1], res, 'cpu', conf, curves=False, skin_grid=(5, 9, 7))
        assert getattr(fresh, 'tmp_sdf_body_vs', None) is None
        fresh.load_state_dict(sd, strict=False)
        assert torch.equal(fresh.tmp_sdf_body_vs, v)
    finally:
        cpu_port.uninstall()


def test_a_stored_skinner_file_is_read_in_the_reference_layout(tmp_path):
    """model/network.py:223-236: `initial_skinner_<pose type>.pth` (ws, bmins, bmaxs, Js, parents, init_pose, tmpBodyVs,
    tmpBodyFs, betas, extra_trans, bbox_center, bbox_extend) + `diffused_skinning_weights.npy` beside the capture -> the
    loop's skinner, its SMPL template, the dataset's shape and the canonical box (volume box + margins, :291)."""
    import os
    import common_setup as cs
    from oracle import cpu_port
    from recmv.dataset import getDatasetAndLoader
    from recmv.hocon import ConfigFactory
    from recmv.model import LBSkinner
    from recmv.model.network import getOptNet
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    root = sc.write_capture(str(tmp_path))
    conds_lens = {'deformer': conf.get_int('mlp_deformer.condlen') * 3, 'renderer': conf.get_int('render_net.condlen')}
    torch.manual_seed(3)
    ds, _ = getDatasetAndLoader(root, conds_lens, 3, True, 0, True, True, conf.get_config('train.opt_camera'), cf.GARMENT_TYPE,
                                data_type='scene')
    # plays the skinner of the reference's first run: ONE normalisation extent (a cube, model/Deformer.py:609), not one per axis
    baked = LBSkinner(**dict(cs.skinner_args(), bbox_extend=torch.tensor(2.4)))
    geo = sc.geometry()
    betas = torch.linspace(-1, 1, 10)
    os.makedirs(os.path.join(root, 'result'))
    torch.save({'ws': torch.zeros_like(baked.ws), 'bmins': baked.b_min, 'bmaxs': baked.b_max, 'Js': baked.Js,
                'parents': baked.parents, 'init_pose': baked.init_pose, 'tmpBodyVs': geo['body_verts'],
                'tmpBodyFs': geo['body_faces'], 'betas': betas, 'extra_trans': torch.tensor([[0.01, -0.02, 0.03]]),
                'bbox_center': baked.bbox_center, 'bbox_extend': baked.bbox_extend},
               os.path.join(root, 'result', 'initial_skinner_0.pth'))
    np.save(os.path.join(root, 'diffused_skinning_weights.npy'), baked.ws[0].contiguous().numpy())
    cpu_port.install()
    try:
        res = [(9, 11, 7), (17, 21, 13)]
        optNet, _ = getOptNet(ds, 'result', 3, None, None, res, 'cpu', conf, curves=True)
        sk = optNet.deformer.defs[1]
        assert torch.equal(sk.ws, baked.ws) and torch.equal(sk.Js, baked.Js) and torch.equal(sk.init_pose, baked.init_pose)
        assert torch.equal(sk.extra_trans, torch.tensor([[0.01, -0.02, 0.03]]))
        assert torch.equal(ds.shape, betas)
        assert torch.equal(optNet.tmpBodyVs, geo['body_verts']) and torch.equal(optNet.tmpBodyFs, geo['body_faces'])
        lo, hi = baked.bbox_size()
        torch.testing.assert_close(optNet.engine.b_min.view(-1).cpu(), lo.view(-1))
        torch.testing.assert_close(optNet.engine.b_max.view(-1).cpu(), hi.view(-1))
        pts = 0.3 * torch.randn(2, 40, 3, generator=torch.Generator().manual_seed(1))
        poses, trans = cs.poses_trans(2, seed=5)
        want = baked(pts, [poses, trans])
        baked.extra_trans.copy_(torch.tensor([[0.01, -0.02, 0.03]]))
        torch.testing.assert_close(sk(pts, [poses, trans]), baked(pts, [poses, trans]))
        assert not torch.equal(want, baked(pts, [poses, trans]))                   # (the stored extra translation is in effect)
        grid = sk._lbs_grid()                                                      # what the fused HIP kernels are handed
        assert list(grid.scale) == pytest.approx([2.0 / 2.4] * 3) and list(grid.center) == pytest.approx(baked.bbox_center.tolist())
        ignored, _ = getOptNet(ds, 'result', 3, None, None, res, 'cpu', conf, curves=False, use_initial_skinner=False,
                               skin_grid=(5, 9, 7))
        assert ignored.deformer.defs[1].ws.shape[2:] == (5, 9, 7)
    finally:
        cpu_port.uninstall()


def test_closed_curve_resampling_and_body_inverse_match_the_reference():
    """engineer/utils/polygons.py `uniformsample3d` (edge-proportional fill and the farthest-point branch that returns one
    point less), `farthest_point_sample`; model/Deformer.py `Inverse_Fl_Body`."""
    from recmv.engineer.utils import matrix_transform as mt
    from recmv.engineer.utils.polygons import farthest_point_sample, uniformsample3d
    from recmv.model import Inverse_Fl_Body
    g = load()
    for tag in ('few', 'few_flipped', 'many'):
        got = uniformsample3d(g['us3d_%s_in' % tag].numpy(), int(g['us3d_%s_n' % tag]))
        assert got.shape == tuple(g['us3d_%s_out' % tag].shape), tag
        torch.testing.assert_close(torch.from_numpy(np.ascontiguousarray(got)).float(), g['us3d_%s_out' % tag], rtol=1e-6, atol=1e-7)
    assert g['us3d_many_out'].shape[0] == int(g['us3d_many_n']) - 1
    assert torch.equal(farthest_point_sample(g['fps_in'], 12).float(), g['fps_out'])
    lines = list(torch.split(g["mt_lines"], [int(n) for n in g["mt_split"]]))
    names = ['a', 'b', 'c', 'd', 'e']
    inv = Inverse_Fl_Body([mt.FeatureLineMesh(v, torch.zeros(1, 3, dtype=torch.long)) for v in lines], names, g['mt_T'], g['inv_S'])
    registered = list(torch.split(g['inv_in'], [int(n) for n in g["mt_split"]]))
    inv.set_rigid_center([v.mean(0, keepdim=True) for v in registered], names)
    torch.testing.assert_close(torch.cat(inv(registered, names), 0), g['inv_out'], rtol=1e-6, atol=1e-6)


def test_align_fl_turns_a_stored_registration_into_the_loops_curves(tmp_path):
    """OptimGarmentNetwork.align_fl (:3485-3546) on the registration the reference produced (startup.npz): every explicit
    curve lies on the longer boundary of its registered ribbon, its canonical-body counterpart on the template's, and the
    loop's curve branch runs on them."""
    import os
    from oracle import cpu_port
    from recmv import curves as fl
    from recmv.dataset import getDatasetAndLoader
    from recmv.engineer.utils.matrix_transform import scale_icp_rotate_center_transform
    from recmv.hocon import ConfigFactory
    from recmv.model.network import getOptNet
    g = load()
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    conf.put('train.garment_type', cf.GARMENT_TYPE)
    root = sc.write_capture(str(tmp_path))
    conds_lens = {'deformer': conf.get_int('mlp_deformer.condlen') * 3, 'renderer': conf.get_int('render_net.condlen')}
    torch.manual_seed(3)
    ds, _ = getDatasetAndLoader(root, conds_lens, 3, True, 0, True, True, conf.get_config('train.opt_camera'), cf.GARMENT_TYPE,
                                data_type='scene')
    os.makedirs(os.path.join(root, 'result', 'fl_init'))
    path = os.path.join(root, 'result', 'fl_init', 'init_trans_matrix.pth')
    torch.save({'rigid_R': g['srig_R'], 'rigid_T': g['srig_T'], 'rigid_scale': g['srig_scale']}, path)
    templates = sc.line_meshes(g, 'cpu')
    # (the two rims of these ribbons have equally many vertices: the second loop found is kept, the reference's tie rule)
    loops = {n: fl.longest_boundary_loop(m.faces_packed()) for n, m in templates.items()}
    assert all(len(l) == m.verts_packed().shape[0] // 2 for l, m in zip(loops.values(), templates.values()))
    cpu_port.install()
    try:
        res, box = [(9, 11, 7), (17, 21, 13)], ((-0.9, -1.2, -0.6), (0.9, 1.2, 0.6))
        optNet, _ = getOptNet(ds, 'result', 3, box[0], box[1], res, 'cpu', conf, curves=False, skin_grid=(5, 9, 7))
        assert not optNet.curves
        optNet.align_fl(path, fl_templates=templates, sample_num=40)
        assert optNet.curves and optNet.fl_names == sc.LINE_NAMES
        curves = optNet.inter_free_curve().detach()                                     # [6,S,3]
        assert curves.shape[0] == 6 and curves.shape[1] in (39, 40) and curves.shape[2] == 3
        moved = scale_icp_rotate_center_transform([templates[n] for n in sc.LINE_NAMES], g['srig_R'], g['srig_T'], g['srig_scale'])
        for i, n in enumerate(sc.LINE_NAMES):
            rim = moved[i][loops[n]]
            seg_a, seg_b = rim, rim.roll(-1, 0)                        # every sample lies on an edge of the registered rim
            ab = (seg_b - seg_a)[None]
            t = (((curves[i][:, None] - seg_a[None]) * ab).sum(-1) / (ab * ab).sum(-1)).clamp(0, 1)
            d = ((seg_a[None] + t[..., None] * ab) - curves[i][:, None]).norm(dim=-1).min(1).values
            assert float(d.max()) < 1e-5, (n, float(d.max()))
            # its canonical-body counterpart: the same samples with translation and scale undone
            body = optNet.inter_free_curve.query_canosmpl_verts([n])[0]
            centre = templates[n].verts_packed().mean(0, keepdim=True)
            want = ((curves[i] - g['srig_T'][i]) - centre) / g['srig_scale'][i] + centre
            torch.testing.assert_close(body, want, rtol=1e-5, atol=1e-6)
        assert set(optNet.fl_extract[optNet.garment_names[0]]) == {'neck', 'left_cuff', 'right_cuff', 'upper_bottom'}
        assert optNet.fl_extract[optNet.garment_names[1]] == ['left_pant', 'right_pant']
        # one iteration of the loop on these curves (train.py's sequence)
        from recmv import utils
        optNet, _ = utils.set_hierarchical_config(conf, 'coarse', optNet, None, res)
        optimizer = optNet.rebuild_optimizer()
        frames = [0, 3, 4]
        datas = torch.utils.data.default_collate([ds[i][1] for i in frames])
        ratio = {'sdfRatio': 1., 'deformerRatio': 0.5, 'renderRatio': 1.}
        optimizer.zero_grad()
        before = optNet.inter_free_curve.scale.detach().clone()
        loss = optNet(datas, 16, ratio, torch.tensor(frames), '/tmp/debug', global_optimizer=optimizer)
        loss.backward()
        optNet.propagateTmpPsGrad(torch.tensor(frames), ratio)
        optimizer.step()
        assert torch.isfinite(loss) and torch.isfinite(optNet.info['fl_loss']['total'])
        assert not torch.equal(before, optNet.inter_free_curve.scale.detach())          # the curve optimiser stepped
        # without a stored file the stand-in rings are drawn instead
        plain, _ = getOptNet(ds, 'result', 3, box[0], box[1], res, 'cpu', conf, curves=False, skin_grid=(5, 9, 7))
        plain.align_fl(os.path.join(root, 'missing.pth'), fl_templates=templates)
        assert plain.curves and plain.inter_free_curve().shape[1] == 200
    finally:
        cpu_port.uninstall()


def test_skinner_baking_matches_the_reference():
    """model/Deformer.py `smooth_weights` (:533-544), `compute_lbswField` (:546-591), `initialLBSkinner` (:594-626, on the
    stand-in SMPL of tests/startup_case.py) and utils/utils.py `smpl_tmp_Apose` (:68-99); the baked skinner then poses points
    like the reference's."""
    import common_setup as cs
    from oracle import cpu_port
    from recmv.utils import smpl_tmp_Apose
    g = load()
    for t in range(4):
        assert torch.equal(torch.from_numpy(smpl_tmp_Apose(t)), g['apose_%d' % t])
    cpu_port.install()
    try:
        sk = sc.run_skinner_baking(g, "cpu")
        assert sk.bbox_extend.dim() == 0                                    # one extent for the three axes
        torch.testing.assert_close(sk(g['bake_pts'], [g['bake_poses'], g['bake_trans']]), g['bake_posed'], rtol=1e-4, atol=1e-6)
    finally:
        cpu_port.uninstall()


def test_smpl_shape_fit_matches_the_reference(tmp_path):
    """engineer/core/beta_optimizer.py `smpl_beta_optimizer` (:132-245) run for real (150 Adam steps against the capture's 2-D
    joints) vs recmv's, the SMPL model replaced by the same stand-in on both sides."""
    from oracle import cpu_port
    cpu_port.install()
    try:
        worst = sc.run_beta_fit(load(), sc.write_joint_capture(str(tmp_path)), "cpu")
        assert max(worst.values()) < 1e-4, worst
    finally:
        cpu_port.uninstall()


def test_first_run_builds_and_stores_the_skinner_when_a_smpl_model_is_given(tmp_path):
    """model/network.py:250-276 through getOptNet: no initial_skinner file + a SMPL model -> shape fit to the 2-D joints, skinner
    baked in the A-pose, `initial_skinner_0.pth` written in the reference's layout; the next getOptNet reads it back (no model
    needed) and gets the same skinner."""
    import os
    from oracle import cpu_port
    from recmv.dataset import getDatasetAndLoader
    from recmv.hocon import ConfigFactory
    from recmv.model.network import getOptNet
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    root = sc.write_joint_capture(str(tmp_path))
    conds_lens = {'deformer': conf.get_int('mlp_deformer.condlen') * 3, 'renderer': conf.get_int('render_net.condlen')}
    torch.manual_seed(3)
    ds, _ = getDatasetAndLoader(root, conds_lens, 3, True, 0, True, True, conf.get_config('train.opt_camera'), cf.GARMENT_TYPE,
                                data_type='scene')
    shape0 = ds.shape.clone()
    cpu_port.install()
    try:
        res = [(9, 11, 7), (17, 21, 13)]
        first, _ = getOptNet(ds, 'result', 3, None, None, res, 'cpu', conf, curves=False, smpl=sc.StandInSMPL(),
                             skin_resolution=(9, 13, 7))
        path = os.path.join(root, 'result', 'initial_skinner_0.pth')
        stored = torch.load(path, weights_only=False)
        assert set(stored) == {'ws', 'bmins', 'bmaxs', 'Js', 'parents', 'init_pose', 'tmpBodyVs', 'tmpBodyFs', 'betas', 'extra_trans',
                               'bbox_center', 'bbox_extend'}
        assert stored['ws'].shape == (1, 24, 7, 13, 9) and stored['tmpBodyVs'].shape == (6890, 3) and stored['bbox_extend'].dim() == 0
        assert not torch.equal(ds.shape, shape0) and torch.equal(ds.shape, stored['betas'])         # the fitted shape
        assert stored['extra_trans'].shape == (1, 3) and float(stored['extra_trans'].abs().max()) > 0
        sk = first.deformer.defs[1]
        assert torch.equal(sk.ws.cpu(), stored['ws']) and torch.equal(first.tmpBodyVs.cpu(), stored['tmpBodyVs'])
        again, _ = getOptNet(ds, 'result', 3, None, None, res, 'cpu', conf, curves=False)           # no model this time
        sk2 = again.deformer.defs[1]
        for name in ('ws', 'b_min', 'b_max', 'Js', 'init_pose', 'extra_trans', 'bbox_center', 'bbox_extend'):
            assert torch.equal(getattr(sk, name), getattr(sk2, name)), name
        assert torch.equal(first.engine.b_min, again.engine.b_min) and torch.equal(first.engine.b_max, again.engine.b_max)
    finally:
        cpu_port.uninstall()


def test_train_py_registers_feature_line_templates_from_a_file(tmp_path):
    """train.py --fl-templates: the npz of ribbon meshes -> `initializeFL` (writes fl_init/init_trans_matrix.pth) -> `align_fl`
    (the loop's curves come from the registered ribbons); a second start re-applies the stored registration."""
    import os
    from oracle import cpu_port
    import train
    from recmv.dataset import getDatasetAndLoader
    from recmv.hocon import ConfigFactory
    from recmv.model.network import getOptNet
    g = load()
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    root = sc.write_capture(str(tmp_path))
    conds_lens = {'deformer': conf.get_int('mlp_deformer.condlen') * 3, 'renderer': conf.get_int('render_net.condlen')}
    torch.manual_seed(3)
    ds, _ = getDatasetAndLoader(root, conds_lens, 3, True, 0, True, True, conf.get_config('train.opt_camera'), cf.GARMENT_TYPE,
                                data_type='scene')
    meshes = sc.line_meshes(g, 'cpu')
    np.savez(tmp_path / 'lines.npz', **{n + k: (m.verts_packed() if k == '_verts' else m.faces_packed()).numpy()
                                        for n, m in meshes.items() for k in ('_verts', '_faces')})
    with pytest.raises(KeyError):
        train.load_fl_templates(str(tmp_path / 'lines.npz'), sc.LINE_NAMES + ['bottom_curve'])
    templates = train.load_fl_templates(str(tmp_path / 'lines.npz'), sc.LINE_NAMES)
    assert all(torch.equal(templates[n].verts_packed(), meshes[n].verts_packed()) for n in sc.LINE_NAMES)
    save_root = os.path.join(root, 'result')
    os.makedirs(save_root)
    cpu_port.install()
    try:
        res, box = [(9, 11, 7), (17, 21, 13)], ((-0.9, -1.2, -0.6), (0.9, 1.2, 0.6))
        optNet, _ = getOptNet(ds, 'result', 3, box[0], box[1], res, 'cpu', conf, curves=True, skin_grid=(5, 9, 7))
        assert optNet.garment_type == cf.GARMENT_TYPE                       # from the dataset: the config does not name it
        rings = optNet.inter_free_curve().detach().clone()
        loop = __import__('types').SimpleNamespace(batch_size=4, world_size=1, rank=0)
        import io
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            train.register_feature_lines(optNet, train.CaptureLoader(ds, loop), templates, save_root)
        path = os.path.join(save_root, 'fl_init', 'init_trans_matrix.pth')
        stored = torch.load(path)
        assert set(stored) == {'rigid_R', 'rigid_T', 'rigid_scale'} and stored['rigid_scale'].shape == (6,)
        curves = optNet.inter_free_curve().detach()
        assert curves.shape[0] == 6 and curves.shape != rings.shape or not torch.equal(curves, rings)
        again, _ = getOptNet(ds, 'result', 3, box[0], box[1], res, 'cpu', conf, curves=True, skin_grid=(5, 9, 7))
        stamp = os.path.getmtime(path)
        with contextlib.redirect_stdout(io.StringIO()):
            train.register_feature_lines(again, train.CaptureLoader(ds, loop), templates, save_root)
        assert os.path.getmtime(path) == stamp                              # re-applied, not re-fitted
        torch.testing.assert_close(again.inter_free_curve().detach(), curves, rtol=1e-6, atol=1e-7)
    finally:
        cpu_port.uninstall()
