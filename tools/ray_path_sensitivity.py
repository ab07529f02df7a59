"""How much of the step does the root finder's device work hold?  An upper bound for what faster ray-path kernels could buy: the bench
scene's iteration timed with the root finder cut to 20 (the product), 10, 5 and 1 steps (a WRONG root finder — fewer rays converge, so
the render phases shrink too; the plain step with 20 steps but every ray declared converged is not reachable this way).  Timing only.

    python tools/ray_path_sensitivity.py"""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402

import bench  # noqa: E402
from recmv import utils  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

dev = torch.device("cuda", 0)
conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
real = utils.OptimizeGarmentSurfacePs
import recmv.utils.FindSurfacePs as F  # noqa: E402

for times in (20, 10, 5, 1):
    def patched(*a, _t=times, **k):
        k["times"] = _t
        return real(*a, **k)
    utils.OptimizeGarmentSurfacePs = patched
    loop = HotLoop(conf, dev, stage="coarse", curves=True, **bench.HOTLOOP_KW)
    it = bench.load_scene(loop, bench.SCENE_FILE)
    torch.manual_seed(1)
    for _ in range(4):
        loop.step(it)
        it += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    conv = 0
    n = 12
    for _ in range(n):
        loop.step(it)
        it += 1
        conv += sum(loop.info.get("rays_converged", []))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("root finder capped at %2d steps: %.2f ms per iteration, %.0f rays converged per iteration" % (times, dt * 1e3, conv / n), flush=True)
utils.OptimizeGarmentSurfacePs = real
