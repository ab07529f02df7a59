"""Where does a re-mesh spend its time?  Loads the bench's frozen scene, runs `marching_cube_update` on the loop's own pyramid
(configs[1]) and on the 33^3 -> 257^3 pyramid of configs[2], and prints every Seg3dLossless query (points, GPU ms, TFLOP/s) plus the
split of the whole re-mesh.      python tools/remesh_breakdown.py"""
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402

import bench  # noqa: E402
from recmv.MCAcc import Seg3dLossless  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import RESOLUTIONS, HotLoop  # noqa: E402

dev = torch.device("cuda", 0)
conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
loop = HotLoop(conf, dev, stage="coarse", curves=True, **bench.HOTLOOP_KW)
it = bench.load_scene(loop, bench.SCENE_FILE)
ratio = {'sdfRatio': 1., 'deformerRatio': loop.opt_times / 2500. + 0.5, 'renderRatio': 1.}
# (round 6: a net whose parameters have not moved is not extracted again — HotLoop.discretizeSDF's cache — so the split below is the
# garments' two pyramids; RECMV_REMESH_CACHE=0 gives all three)
FLOP_PER_POINT = 2 * 1966592 - 2 * 256 * 512          # SDF value only: the last layer's 256 feature rows are skipped


def run(tag):
    sizes = []
    real = loop.sdf.__class__.forward

    def spy(self, input, *a, **k):
        sizes.append(int(input.shape[0]))
        return real(self, input, *a, **k)
    loop.sdf.__class__.forward = spy
    try:
        for rep in range(3):
            sizes.clear()
            loop.remesh_trace = []
            # the loop's case: the garment nets have moved since the last re-mesh, the body net has not (its entry stays cached)
            for i in list(getattr(loop, '_remesh_cache', {})):
                if i > 0:
                    del loop._remesh_cache[i]
            torch.cuda.synchronize()
            loop.marching_cube_update(ratio)
            torch.cuda.synchronize()
            tr = loop.remesh_trace
    finally:
        loop.sdf.__class__.forward = real
        loop.remesh_trace = None
    q = [(e0.elapsed_time(e1)) for n, e0, e1, _, _ in tr if n == 'query']
    tot = {n: e0.elapsed_time(e1) for n, e0, e1, _, _ in tr if n != 'query'}
    print("== %s: re-mesh %.2f ms = pyramid %.2f (queries %.2f in %d calls, bookkeeping %.2f) + MC %.2f + hand-over %.2f; vertices %s"
          % (tag, tot['remesh'], tot['pyramid'], sum(q), len(q), tot['pyramid'] - sum(q), tot['mc'],
             tot['remesh'] - tot['pyramid'] - tot['mc'], [int(v.shape[0]) for v in [loop.body_vs] + list(loop.garment_vs)]))
    for n, ms in zip(sizes, q):
        print("   query %8d points  %7.3f ms  %6.1f TFLOP/s" % (n, ms, n * FLOP_PER_POINT / ms / 1e9))
    print("   total %d points, %.2f TFLOP" % (sum(sizes), sum(sizes) * FLOP_PER_POINT / 1e12))


run("configs[1] pyramid %s" % (tuple(int(v) for v in loop.engine.resolutions[-1]),))
old = loop.engine
loop.engine = Seg3dLossless(query_func=None, b_min=old.b_min.view(-1).tolist(), b_max=old.b_max.view(-1).tolist(),
                            resolutions=RESOLUTIONS['higher256'], align_corners=False, balance_value=0.0, use_cuda_impl=True,
                            faster=False).to(dev)
run("configs[2] pyramid 257^3")
