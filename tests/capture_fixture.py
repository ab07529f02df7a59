"""A small synthetic capture directory in the reference's on-disk layout (dataset/dataset.py:181-239), written the same way by
the golden generator (tests/golden/make_golden_dataset.py, which reads it with the REFERENCE's dataset classes) and by the
test (which reads it with recmv.dataset): seeded, so both see identical files."""
import json
import os

import numpy as np

FRAMES, H, W = 12, 24, 20
GARMENT_TYPE = 'female-3-casual'
LINES = ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'left_pant', 'right_pant']
ANNOTATED = {'featurelines': [0, 3, 4, 9], 'mask2fl': [2, 3, 5, 6]}


def _polyline(rng, n, closed_gap):
    t = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = 3.0 + rng.uniform(-0.3, 0.3, n)
    c = rng.uniform(6, 14, 2)
    pts = np.stack([c[0] + r * np.cos(t), c[1] + 1.4 * r * np.sin(t)], -1)
    if closed_gap:                       # an annotation that wraps around: its longest gap is in the middle of the list
        pts = np.roll(pts, n // 3, axis=0)
    return pts


def write_capture(root, seed=7, H=H, W=W, loop_camera=False):
    """`loop_camera`: the pinhole the synthetic frames of recmv.loop use (object at z = 3 in view), for runs of the loop."""
    from PIL import Image
    import joblib
    rng = np.random.RandomState(seed)
    for d in ('imgs', 'masks', 'parsing_SCH_ATR', 'normals', 'featurelines', 'mask2fl'):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for i in range(FRAMES):
        img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, 'imgs', '%d.png' % i))
        yy, xx = np.mgrid[0:H, 0:W]
        fg = ((yy - H / 2 - np.sin(i)) ** 2 / (H / 2.4) ** 2 + (xx - W / 2) ** 2 / (W / 3.0) ** 2) < 1
        m = np.zeros((H, W, 3), np.uint8)
        m[fg] = (255, 0, 0) if i % 2 else (255, 255, 255)
        Image.fromarray(m).save(os.path.join(root, 'masks', '%d.png' % i))
        labels = rng.choice([0, 2, 4, 6, 11, 12, 14, 5], size=(H, W), p=[.3, .1, .15, .15, .05, .1, .1, .05])
        labels = (labels * fg).astype(np.int64)
        np.save(os.path.join(root, 'parsing_SCH_ATR', '%d.npy' % i), labels)
        # the pre-processed label map the sample reader expects beside it (reference: SceneDataset.parsing_mask)
        filled = np.where(fg, np.where(labels > 0, labels, 4), 0).astype(np.uint8)
        np.save(os.path.join(root, 'parsing_SCH_ATR', 'mask_parsing_%d.npy' % i), filled)
        if i % 3 != 1:
            nrm = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
            Image.fromarray(nrm).save(os.path.join(root, 'normals', '%d.png' % i))
    for sub, frames in ANNOTATED.items():
        for k, i in enumerate(frames):
            shapes = []
            for j, name in enumerate(LINES):
                if (i + j) % 5 == 4:
                    continue                                                  # a line missing from this frame
                n = [9, 40, 130, 17, 60, 25][j] + (3 if sub == 'mask2fl' else 0)   # fewer and more points than fl_sampling
                shapes.append({'label': name, 'points': _polyline(rng, n, (i + j) % 2 == 0).tolist(), 'shape_type': 'polygon'})
            with open(os.path.join(root, sub, '%d.json' % i), 'w') as fh:
                json.dump({'shapes': shapes}, fh)
    np.savez(os.path.join(root, 'smpl_rec.npz'), poses=rng.randn(FRAMES, 72).astype(np.float32) * 0.2,
             trans=rng.randn(FRAMES, 3).astype(np.float32) * 0.05, shape=rng.randn(10).astype(np.float32), gender='female')
    q = rng.randn(4)
    if loop_camera:
        np.savez(os.path.join(root, 'camera.npz'), fx=np.float32(1000. * W / 512), fy=np.float32(1000. * H / 512),
                 cx=np.float32(W / 2), cy=np.float32(H / 2), quat=np.array([0., 0., 0., 1.], np.float32),
                 T=np.array([0., 0., 3.], np.float32))
    else:
        np.savez(os.path.join(root, 'camera.npz'), fx=np.float32(900.), fy=np.float32(910.), cx=np.float32(W / 2),
                 cy=np.float32(H / 2), quat=(q / np.linalg.norm(q)).astype(np.float32), T=np.array([0.02, -0.01, 2.5], np.float32))
    joblib.dump([None, {'gt_joints2d': rng.rand(FRAMES, 49, 3).astype(np.float32), 'frame_ids': np.arange(FRAMES),
                        'pose': rng.randn(FRAMES, 72).astype(np.float32), 'betas': rng.randn(FRAMES, 10).astype(np.float32)}],
                os.path.join(root, '%s_tcmr_output.pkl' % GARMENT_TYPE))
    return root


def collect(ds, kind, samples=(0, 1, 5, 11)):
    """What the golden pins of a dataset object (reference's or recmv's): samples, per-line weights, windows, camera."""
    import torch
    out = {}
    for idx in samples:
        i, s = ds[idx]
        assert i == idx + getattr(ds, 'start_idx', 0) * int(type(ds).__name__.startswith('Large_Pose'))
        out['%s_s%d_index' % (kind, idx)] = torch.tensor([float(i)])
        for k, v in s.items():
            out['%s_s%d_%s' % (kind, idx, k)] = torch.as_tensor(np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v).float()
    out[kind + '_fl_weights'] = torch.tensor([ds.fl_weights[n] for n in ds.fl_names]).float()
    out[kind + '_fl_supervised'] = torch.tensor([float(b) for b in ds.fl_supervised])
    fids = torch.tensor([0, 1, len(ds) // 2, len(ds) - 2, len(ds) - 1])   # (len = the sub-range of a PeopleSnapshot capture)
    win, pos = ds.get_batchframe_data('poses', fids, 3)
    out[kind + '_window'], out[kind + '_window_pos'] = win.detach().float(), pos.float()
    cam = ds.get_camera_parameters(2, 'cpu')
    for j, t in enumerate(cam[:4]):
        out['%s_cam%d' % (kind, j)] = t.detach().float()
    out[kind + '_hw'] = torch.tensor([float(cam[4]), float(cam[5])])
    p, t, c0, c1 = ds.get_grad_parameters(torch.tensor([3, 7]), 'cpu')
    out[kind + '_poses'], out[kind + '_trans'] = p.detach(), t.detach()
    out[kind + '_cond0'], out[kind + '_cond1'] = c0.detach(), c1.detach()
    out[kind + '_len'] = torch.tensor([float(len(ds)), float(ds.all_size()), float(getattr(ds, 'start_idx', -1))])
    return out
