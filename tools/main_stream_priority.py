"""Does a LOW-priority main stream (the large products) let the ray chain's short kernels through?  The iteration timed with the
loop's main stream = the legacy default stream (the product), a normal-priority created stream, and the lowest-priority stream the
device offers, each with the side streams at normal and at high priority.      python tools/main_stream_priority.py"""
import os
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def child(kind):
    sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
    import torch
    import bench
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    dev = torch.device("cuda", 0)
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    st = None if kind == "default" else torch.cuda.Stream(device=dev, priority=(0 if kind == "normal" else lo))
    ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        loop = HotLoop(conf, dev, stage="coarse", curves=True, **bench.HOTLOOP_KW)
        it = bench.load_scene(loop, bench.SCENE_FILE)
        torch.manual_seed(1)
        for _ in range(4):
            loop.step(it)
            it += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 14
        for _ in range(n):
            loop.step(it)
            it += 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print("RESULT main stream %-8s (priority range of the device: lowest %d .. highest %d), side streams %s: %.2f ms per iteration" % (
        kind, lo, hi, "high" if os.environ.get("RECMV_SIDE_PRIORITY") == "1" else "normal", dt * 1e3), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for side in ("0", "1"):
            for kind in ("default", "normal", "lowest"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind],
                                   env=dict(os.environ, RECMV_SIDE_PRIORITY=side), capture_output=True, text=True)
                out = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
                print(out[0][7:] if out else "FAILED (%s, side %s): %s" % (kind, side, r.stderr.strip().splitlines()[-1][:200]), flush=True)
