"""Diagnostic: how the root finder's rays end (|sdf|, ray angle) on the bench workload.  GPU only."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
os.environ.setdefault("RECMV_ROOT_TRACE", "1")

from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
dev = torch.device("cuda:0")
loop = HotLoop(conf, dev, n_frames=64, H=512, W=512, stage="coarse")
orig = loop.opt_garment_surface_ps


def spy(frame_ids, cameras, ratio, samples):
    pts, checks = orig(frame_ids, cameras, ratio, samples)
    d_cond_list, poses, trans, _ = loop.get_grad_parameters(frame_ids, dev)
    with torch.no_grad():
        for g_i, name in enumerate(loop.garment_names):
            b, r, c, p0, rays = samples[g_i]
            for tag, p in (("start", p0), ("end", pts[g_i])):
                f = loop.garment_nets[g_i](p, ratio, features=False).view(-1)
                d = loop.deformer(p, [d_cond_list[g_i + 1], [poses, trans]], b, ratio=ratio, offset_type=name)
                v = d - cameras.cam_pos().view(1, 3)
                ang = torch.arcsin(torch.linalg.cross(v, rays, dim=1).norm(dim=1) / v.norm(dim=1)) * 180 / np.pi
                q = torch.tensor([0.1, 0.5, 0.9], device=dev)
                print(f"{name} {tag}: |sdf| q10/50/90 = {f.abs().quantile(q).tolist()}  angle = "
                      f"{ang.quantile(q).tolist()}  (thr {5e-5}, {loop.angThred:.5f}); ok={int(checks[g_i].sum())}/"
                      f"{checks[g_i].numel()}; both<thr: {int(((f.abs() < 5e-5) & (ang < loop.angThred)).sum())}"
                      f" sdf<thr: {int((f.abs() < 5e-5).sum())} ang<thr: {int((ang < loop.angThred).sum())}",
                      flush=True)
    return pts, checks


loop.opt_garment_surface_ps = spy
for it in range(3):
    loop.step(it)
