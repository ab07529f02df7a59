"""GPU timing of the two rasterisers at the loop's shapes (3 x 512 x 512, ~150k points / ~300k faces per frame)."""
import sys
import time
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
loop.step(0)
loop.step(1)
frame_ids = loop.frame_batch(2)
ratio = {'sdfRatio': 1., 'deformerRatio': 0.5, 'renderRatio': 1.}
cams = loop._cameras()
N = frame_ids.numel()
d_cond_list, poses, trans, _ = loop.get_grad_parameters(frame_ids, dev)
with torch.no_grad():
    def_vs = [loop.deformer(gv[None].expand(N, -1, 3), [d_cond_list[g + 1], [poses, trans]], ratio=ratio, offset_type=nm)
              for g, (gv, nm) in enumerate(zip(loop.garment_vs, loop.garment_names))]


def timed(name, fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print("%-60s %8.3f ms" % (name, (time.perf_counter() - t0) / n * 1e3), flush=True)


rast = raster.MeshRasterizer(cams, (512, 512))
for g in range(2):
    gf, gv = loop.garment_fs[g], loop.garment_vs[g].detach()
    timed(f"mesh rasteriser, garment {g} ({gf.shape[0]} faces x {N})", lambda: rast(def_vs[g], gf))
    frags = rast(def_vs[g], gf)
    timed(f"FindSurfacePs, garment {g}", lambda: utils.FindSurfacePs(gv, gf, frags))
whole = torch.cat(def_vs, 1).requires_grad_(True)
rend = raster.PointsRendererWithFrags_Split(cams, (512, 512), radius=loop.pc_radius, points_per_pixel=50)
timed(f"point renderer forward ({whole.shape[1]} points x {N})", lambda: rend(whole, loop.garment_vs[0].shape[0]))


def fb():
    imgs, _ = rend(whole, loop.garment_vs[0].shape[0])
    (imgs[0].sum() + imgs[1].mean()).backward()


timed("point renderer forward + backward", fb)
loop.sample_train_ray  # noqa: B018
loop._surface_inputs = (def_vs, [v.detach().clone() for v in loop.garment_vs])
loop._surface_ready = torch.cuda.Event()
loop._surface_ready.record()


def srays():
    loop._frag_cache = {}
    loop._surface_inputs = (def_vs, [v.detach().clone() for v in loop.garment_vs])
    loop.sample_train_ray(N, frame_ids, cams)


timed("sample_train_ray (2 rasterisations + FindSurfacePs + sampling)", srays)

# ---- sample_train_ray, section by section (synchronised timers)
import contextlib  # noqa: E402


class T:
    def __init__(self):
        self.acc = {}

    @contextlib.contextmanager
    def __call__(self, name):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        yield
        torch.cuda.synchronize()
        self.acc[name] = self.acc.get(name, 0.0) + time.perf_counter() - t0


tm = T()
REP = 5
for _ in range(REP):
    tmp_vs = [v.detach().clone() for v in loop.garment_vs]
    loop._frag_cache = {}
    with tm("find_surface_ps"):
        found = loop.find_surface_ps(def_vs, tmp_vs, cams)
    for g_i, (b, r, c, p0, _f) in enumerate(found):
        with tm("gt mask gather + nonzero + 4 index"):
            gt = loop.dataset.garment_masks(g_i, frame_ids)
            keep = (gt[b, r, c] > 0.).nonzero(as_tuple=True)[0]
            b, r, c, p0 = (t[keep] for t in (b, r, c, p0))
        pnum = b.shape[0]
        with tm("host rand + nonzero"):
            import numpy as np
            sel = torch.rand(pnum).numpy() < float(1024 * N) / float(pnum)
            idx = torch.from_numpy(np.flatnonzero(sel))
        with tm("H2D of the index"):
            idx = idx.to(b.device)
        with tm("4 index + rays"):
            b, r, c, p0 = (t[idx] for t in (b, r, c, p0))
            rays = cams.view_rays(torch.cat([c.view(-1, 1), r.view(-1, 1), torch.ones_like(c.view(-1, 1))], -1).float())
for k, v in tm.acc.items():
    print("  %-44s %8.3f ms / call" % (k, v / REP * 1e3))
print("  pnum", pnum, "threads", torch.get_num_threads())
