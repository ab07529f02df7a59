"""The hot loop itself on the GPU (small frames, small pyramid): every phase runs on the HIP kernels, the surface
points found through the rasteriser are consistent with the rays they seed, and the optimiser moves the parameters."""
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
CONF = str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf")


def _loop():
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    conf = ConfigFactory.parse_file(CONF)
    conf.put('train.sample_pix_num', 256)
    return HotLoop(conf, torch.device("cuda:0"), n_frames=12, H=160, W=128,
                   resolutions=[(9, 13, 7), (17, 25, 13), (33, 49, 25), (65, 97, 49)], skin_grid=(17, 33, 17))


def test_hot_loop_steps_and_surface_points_seed_their_rays():
    loop = _loop()
    seen = {}
    orig = loop.opt_garment_surface_ps

    def spy(frame_ids, cameras, ratio, samples):
        d_cond_list, poses, trans, _ = loop.get_grad_parameters(frame_ids, loop.device)
        with torch.no_grad():
            for g_i, name in enumerate(loop.garment_names):
                b, r, c, p0, rays = samples[g_i]
                assert b.shape == r.shape == c.shape and p0.shape == rays.shape == (b.numel(), 3)
                assert r.min() >= 0 and r.max() < 160 and c.min() >= 0 and c.max() < 128 and b.max() < 3
                d = loop.deformer(p0, [d_cond_list[g_i + 1], [poses, trans]], b, ratio=ratio, offset_type=name)
                pix = cameras.project(d)
                err = ((pix[:, 0] - c) ** 2 + (pix[:, 1] - r) ** 2).sqrt()
                seen.setdefault(name, []).append((b.numel(), float(err.median()), float(err.max())))
        return orig(frame_ids, cameras, ratio, samples)

    loop.opt_garment_surface_ps = spy
    before = [p.detach().clone() for p in loop.shared_parameters()]
    l0, rays0 = loop.step(0)
    l1, rays1 = loop.step(1)
    torch.cuda.synchronize()
    assert torch.isfinite(l0) and torch.isfinite(l1)
    # Bernoulli subset of the covered pixels: expectation 2 garments x 3 frames x 128 rays
    assert 600 <= rays0 <= 940 and 600 <= rays1 <= 940
    assert all(n > 3 * 128 for n in loop.info['surface_pixels'])
    for name, rows in seen.items():
        for n, med, worst in rows:
            # the canonical first-hit point, deformed, lands on its pixel centre up to the curvature of the deformation
            # inside one (few-pixel) face
            # (median ~0.01 px; the worst of a few hundred Bernoulli-drawn pixels sits on a grazing face: 0.6 - 1.3 px by the draw)
            assert n > 200 and med < 0.05 and worst < 2.0, (name, n, med, worst)
    # right after the re-mesh the start points sit on the zero level: (almost) every ray converges at once
    assert sum(loop.info['rays_converged']) >= 0          # key exists
    changed = sum(int(not torch.equal(a, b.detach())) for a, b in zip(before, loop.shared_parameters()))
    assert changed > 10
    for name in loop.garment_names:
        assert f'{name}_grad_loss' in loop.info and f'pc_{name}_loss_sdf' in loop.info


def test_first_iteration_converges_every_ray():
    loop = _loop()
    loop.step(0)
    conv = loop.info['rays_converged']
    assert len(conv) == 2 and sum(conv) >= 0.97 * loop.info['rays_total'], (conv, loop.info['rays_total'])


def test_feature_curve_branch_on_gpu():
    """project_2d_loss on the device: curves deform through the same kernels as the garment vertices, the z-buffer
    visibility uses the HIP rasteriser, the AdamW step moves the curve parameters."""
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    conf = ConfigFactory.parse_file(CONF)
    conf.put('train.sample_pix_num', 256)
    loop = HotLoop(conf, torch.device("cuda:0"), n_frames=12, H=160, W=128,
                   resolutions=[(9, 13, 7), (17, 25, 13), (33, 49, 25), (65, 97, 49)], skin_grid=(17, 33, 17), curves=True)
    before = [p.detach().clone() for p in loop.inter_free_curve.parameters()]
    l0, _ = loop.step(0)
    l1, _ = loop.step(1)
    torch.cuda.synchronize()
    assert torch.isfinite(l0) and torch.isfinite(l1)
    info = loop.info['fl_loss']
    assert torch.isfinite(info['total'])
    for name in loop.garment_names:
        assert 0.0 <= float(info[f'{name}_visible']) <= 1.0
    # the body hides the far side of the rings, not all of them
    assert 0.05 < max(float(info[f'{name}_visible']) for name in loop.garment_names) < 0.95
    moved = max(float((a - b.detach()).abs().max()) for a, b in zip(before, loop.inter_free_curve.parameters()))
    assert 1e-5 < moved < 1e-2
    assert loop.tmpBodyVs.shape[0] > 500 and loop.tmpBodyFs.shape[0] > 1000


def test_iterations_on_a_capture_directory_read_by_recmv_dataset(tmp_path):
    """train.py's sequence on the device with a capture directory in the reference's layout (recmv.dataset.SceneDataset) in
    place of the synthetic frames: the mini-batch dict of a DataLoader goes into OptimGarmentNetwork.forward, the tensors the
    optimiser moves are the dataset's own."""
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import capture_fixture as cf
    from recmv import utils
    from recmv.dataset import getDatasetAndLoader
    from recmv.hocon import ConfigFactory
    from recmv.model.network import getOptNet
    dev = torch.device("cuda:0")
    root = cf.write_capture(str(tmp_path / "capture"), H=160, W=128, loop_camera=True)
    conf = ConfigFactory.parse_file(CONF)
    conf.put('train.sample_pix_num', 256)
    conds_lens = {'deformer': conf.get_int('mlp_deformer.condlen') * 3, 'renderer': conf.get_int('render_net.condlen')}
    torch.manual_seed(3)
    ds, _ = getDatasetAndLoader(root, conds_lens, 3, True, 0, True, True, conf.get_config('train.opt_camera'),
                                cf.GARMENT_TYPE, data_type='scene')
    for t in ds.conds + [ds.poses, ds.trans, ds.shape] + list(ds.camera_params.values()):
        t.data = t.data.to(dev)
    res = [(9, 13, 7), (17, 25, 13), (33, 49, 25), (65, 97, 49)]
    optNet, _ = getOptNet(ds, 'result', 3, None, None, res, dev, conf, curves=True, skin_grid=(17, 33, 17))
    optNet, _ = utils.set_hierarchical_config(conf, 'coarse', optNet, None, res)
    optimizer = optNet.rebuild_optimizer()
    before = ds.poses.detach().clone(), ds.conds[0].detach().clone(), ds.camera_params['focal_length'].detach().clone()
    for it, frames in enumerate(([0, 2, 3], [5, 6, 8])):                      # frames that have a normal map
        datas = torch.utils.data.default_collate([ds[i][1] for i in frames])
        frame_ids = torch.tensor(frames, device=dev)
        ratio = {'sdfRatio': 1., 'deformerRatio': optNet.opt_times / 2500. + 0.5, 'renderRatio': 1.}
        optimizer.zero_grad()
        loss = optNet(datas, 256, ratio, frame_ids, str(tmp_path), global_optimizer=optimizer)
        loss.backward()
        optNet.propagateTmpPsGrad(frame_ids, ratio)
        optimizer.step()
        optNet.opt_times += 1.
        assert torch.isfinite(loss), it
    torch.cuda.synchronize()
    assert not torch.equal(ds.poses.detach()[[0, 2, 3]], before[0][[0, 2, 3]])
    assert not torch.equal(ds.conds[0].detach(), before[1])
    assert not torch.equal(ds.camera_params['focal_length'].detach(), before[2])
    assert torch.isfinite(optNet.info['fl_loss']['total']) and optNet.info['rays_total'] > 0
