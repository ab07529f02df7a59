"""Why do blocks of iterations on the same scene take 96 or 104 ms?  Per-iteration wall time after the settle phase next to what
could flip between iterations: rays per garment, converged rays, whether the root finder compacted its rows (and to how many), device
allocations (hipMalloc calls of the caching allocator), re-mesh.        python tools/iter_jitter.py [iters]"""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    import importlib
    FS = importlib.import_module('recmv.utils.FindSurfacePs')
    dev = torch.device("cuda", 0)
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    loop = HotLoop(conf, dev, stage="coarse", curves=True, **bench.HOTLOOP_KW)
    log = []
    real = FS._RootState._compact

    def compact(self):
        real(self)
        log.append((int(self.p.shape[0]), getattr(self, "live", None) if self.perm is not None else None))
    FS._RootState._compact = compact
    it = 0
    for _ in range(240):
        loop.step(it)
        it += 1
    torch.cuda.synchronize()
    print("# it   ms    rays/garment   converged   root finder rows -> rows kept after the first update (None: not compacted)   hipMallocs   remesh")
    for _ in range(n):
        del log[:]
        a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        remesh = loop.forward_time % loop.remesh_intersect == 0
        t0 = time.perf_counter()
        loop.step(it)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        a1 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        print("%4d %6.1f   %-14s %-12s %-40s %3d  %s" % (it, ms, loop.info.get("rays_total"), loop.info.get("rays_converged"), log, a1 - a0,
                                                          "re-mesh" if remesh else ""), flush=True)
        it += 1


if __name__ == "__main__":
    main()
