"""Golden vectors for the data path, from the REAL reference dataset classes (dataset/dataset.py) reading the synthetic capture
of tests/capture_fixture.py:

  dataset.npz   `SceneDataset`, `People_Snapshot_SceneDataset`, `Large_Pose_SceneDataset` (a_pose False / True) and `Synthe_SceneDataset`: four samples each (image, mask,
                feature-line points and flags, normal map, garment regions, 2-D joints), the per-line projection weights
                (`area_size_statistic`), which frames carry an annotation, temporal windows (`get_batchframe_data`), camera
                tuple, per-frame tensors incl. the DCT-initialised codes (seeded), lengths; the samplers' index streams; the
                nearest-label fill of `load_parsing_mask`.

The reference reads images with OpenCV, which this image does not have: `cv2.imread` is served by Pillow in OpenCV's channel
order (recmv.dataset.read_image_bgr — the decoder is the one thing this golden does not pin).  `load_parsing_mask` uses
pytorch3d's knn on the GPU; its golden is a brute-force nearest-label fill in float64.

    python tests/golden/make_golden_dataset.py
"""
import random
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import PIL.Image  # noqa: E402,F401  (the real Pillow, before the loader's dummy finder could shadow it)
PIL.Image.init()   # ... and its format plugins
import joblib  # noqa: E402,F401
import ref_loader  # noqa: E402
from recmv.dataset import read_image_bgr  # noqa: E402

ref_loader.install()
cv2 = types.ModuleType("cv2")
cv2.imread = lambda path, *a: read_image_bgr(path)
cv2.IMREAD_UNCHANGED = -1
sys.modules["cv2"] = cv2
import capture_fixture as cf  # noqa: E402
from make_golden import save  # noqa: E402

refds = ref_loader.ref_module("dataset.dataset")
CONDS = {'deformer': 16, 'render': 8}


def main():
    out = {}
    with tempfile.TemporaryDirectory() as root:
        cf.write_capture(root)
        torch.manual_seed(11)
        out.update(cf.collect(refds.SceneDataset(root, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30, curve_sampling=2), 'scene'))
        for a_pose in (False, True):
            torch.manual_seed(12)
            ds = refds.People_Snapshot_SceneDataset(root, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30, curve_sampling=1,
                                                    a_pose=a_pose)
            ds.gt_joints2d = None if False else ds.gt_joints2d
            out.update(cf.collect(ds, 'ps%d' % int(a_pose)))
            out['ps%d_apose' % int(a_pose)] = torch.tensor([float(ds.a_pose_start), float(ds.a_pose_end)])
        for a_pose in (False, True):
            torch.manual_seed(13)
            ds = refds.Large_Pose_SceneDataset(root, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30, curve_sampling=1,
                                               a_pose=a_pose)
            kind = 'lp%d' % int(a_pose)
            out.update(cf.collect(ds, kind, samples=(0, 1, len(ds) - 1)))
            out[kind + '_apose'] = torch.tensor([float(ds.a_pose_start), float(ds.a_pose_end)])
            out[kind + '_all_trans'], out[kind + '_all_poses'], out[kind + '_shape'] = ds.trans.clone(), ds.poses.clone(), ds.shape.clone()
        torch.manual_seed(14)
        syn = refds.Synthe_SceneDataset(root, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30, curve_sampling=2)
        for idx in (1, 5):                                   # odd frames: a SceneDataset with curve_sampling=2 would blank their lines
            i, smp = syn[idx]
            out['synthe_s%d_fl_masks' % idx], out['synthe_s%d_fl_pts' % idx] = smp['fl_masks'].float(), smp['fl_pts'].float()
            out['synthe_s%d_keys' % idx] = torch.tensor([float(k in smp) for k in ('gt_joints2d', 'normal', 'upper', 'bottom')])
        # the smoother on an axis-angle sequence with a sign flip of one joint (the branch that re-expresses the rotation)
        smooth = ref_loader.ref_module("engineer.utils.smooth_poses").smooth_poses
        g = torch.Generator().manual_seed(9)
        seq = 0.3 * torch.randn(1, 5, 3, generator=g) + 0.02 * torch.randn(14, 5, 3, generator=g).cumsum(0)
        seq[6:, 2] = -seq[6:, 2] * (1 + (2 * np.pi - 2 * seq[6:, 2].norm(dim=-1, keepdim=True)) / seq[6:, 2].norm(dim=-1, keepdim=True))
        out['smooth_in'] = seq.clone()
        out['smooth_out'] = smooth(seq, min_cutoff=0.004, beta=0.7, d_cutoff=1.)
        out['smooth_out_default'] = smooth(seq)
        ds = refds.SceneDataset(root, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30)
        for name, cls, arg in (('random', refds.RandomSampler, 3), ('clip', refds.ClipSampler, 5)):
            for shuffle in (False, True):
                random.seed(5)
                torch.manual_seed(5)
                out['sampler_%s_%d' % (name, int(shuffle))] = torch.tensor(list(iter(cls(ds, arg, shuffle))), dtype=torch.float32)
                out['sampler_%s_%d_len' % (name, int(shuffle))] = torch.tensor([float(len(cls(ds, arg, shuffle)))])
        # nearest-label fill, brute force in float64
        g = torch.Generator().manual_seed(3)
        mask = (torch.rand(18, 14, generator=g) < 0.7).float()
        logits = (torch.randint(0, 6, (18, 14), generator=g) * (torch.rand(18, 14, generator=g) < 0.25)).long() * mask.long()
        li, lj = torch.nonzero(logits, as_tuple=True)
        mi, mj = torch.nonzero(mask, as_tuple=True)
        d = ((torch.stack([mi, mj], -1)[:, None].double() - torch.stack([li, lj], -1)[None].double()) ** 2).sum(-1)
        filled = torch.zeros_like(logits)
        filled[mi, mj] = logits[li, lj][d.argmin(1)]
        out['fill_mask'], out['fill_logits'], out['fill_out'] = mask, logits.float(), filled.float()
    save("dataset", **out)


if __name__ == "__main__":
    main()
