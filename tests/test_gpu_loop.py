"""The hot loop itself on the GPU (small frames, small pyramid): every phase runs on the HIP kernels, the surface
points found through the rasteriser are consistent with the rays they seed, and the optimiser moves the parameters."""
import sys
from pathlib import Path

import numpy as np
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
            assert n > 200 and med < 0.05 and worst < 1.0, (name, n, med, worst)
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
