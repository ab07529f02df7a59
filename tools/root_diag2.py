"""Diagnostic: where do the first-hit surface points land when pushed through the deformer?  GPU only."""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]

from recmv import raster, utils  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
dev = torch.device("cuda:0")
loop = HotLoop(conf, dev, n_frames=64, H=512, W=512, stage="coarse")
orig = loop.sample_train_ray


def spy(N, frame_ids, cameras):
    def_vs, tmp_vs = loop._surface_inputs
    d_cond_list, poses, trans, _ = loop.get_grad_parameters(frame_ids, dev)
    rast = raster.MeshRasterizer(cameras, (loop.dataset.H, loop.dataset.W))
    torch.cuda.synchronize()
    with torch.no_grad():
        for g_i, name in enumerate(loop.garment_names):
            gf = loop.garment_fs[g_i]
            frags = rast(def_vs[g_i], gf)
            b, r, c, p0, finds = utils.FindSurfacePs(tmp_vs[g_i], gf, frags)
            w = frags.bary_coords[b, r, c, 0]
            tri_d = def_vs[g_i][b[:, None], gf[finds]]
            lin = (w[:, :, None] * tri_d).sum(1)
            conds = [d_cond_list[g_i + 1], [poses, trans]]
            d = loop.deformer(p0, conds, b, ratio=loop._r, offset_type=name)
            pl, pd = cameras.project(lin), cameras.project(d)
            el = ((pl[:, 0] - c) ** 2 + (pl[:, 1] - r) ** 2).sqrt()
            ed = ((pd[:, 0] - c) ** 2 + (pd[:, 1] - r) ** 2).sqrt()
            epx = (cameras.project(tri_d[:, 0]) - cameras.project(tri_d[:, 1])).norm(dim=1)
            print(name, 'n', b.numel(), 'lin err max', el.max().item(), 'def err med/max', ed.median().item(),
                  ed.max().item(), '|d-lin| med', (d - lin).norm(dim=1).median().item(), 'edge px med',
                  epx.median().item(), 'edge m', (tri_d[:, 0] - tri_d[:, 1]).norm(dim=1).median().item(), flush=True)
            cc = tmp_vs[g_i][gf[finds]].reshape(-1, 3)
            bb = b.repeat_interleave(3)
            dc = loop.deformer(cc, conds, bb, ratio=loop._r, offset_type=name).view(-1, 3, 3)
            print('  corner deform (flat) vs def_vs (batched)', (dc - tri_d).abs().max().item(), flush=True)
            # split the deformer: offset MLP alone and skinning alone, linear-interp residuals
            q = loop.deformer.defs[0](p0, conds[0], b, ratio=loop._r, offset_type=name)
            qc = loop.deformer.defs[0](cc, conds[0], bb, ratio=loop._r, offset_type=name).view(-1, 3, 3)
            print('  offset-MLP residual vs its linear interp med', ((w[:, :, None] * qc).sum(1) - q).norm(dim=1).median().item(),
                  ' |offset| med', (q - p0).norm(dim=1).median().item(), flush=True)
            s = loop.deformer.defs[1](q, conds[1], b)
            sc = loop.deformer.defs[1](qc.reshape(-1, 3), conds[1], bb).view(-1, 3, 3)
            print('  skinning residual vs its linear interp med', ((w[:, :, None] * sc).sum(1) - s).norm(dim=1).median().item(),
                  flush=True)
    torch.cuda.synchronize()
    return orig(N, frame_ids, cameras)


loop.sample_train_ray = spy
of = loop.forward


def fw(frame_ids, ratio):
    loop._r = ratio
    return of(frame_ids, ratio)


loop.forward = fw
loop.step(0)
