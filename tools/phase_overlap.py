"""Host and device intervals of the phases of one iteration WITHOUT synchronising (RECMV_HOST_TRACE=1: HotLoop._phase records the
host's enter / exit times and HIP events on the stream each phase is queued on).  Prints, per phase, when the host queued it and
when the device ran it, relative to the start of the iteration — who waits for whom.

    python tools/phase_overlap.py [iterations]
"""
import os
import sys
import time
from pathlib import Path

os.environ["RECMV_HOST_TRACE"] = "1"
REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "rec-mv_amd", REPO):
    sys.path.insert(0, str(p))
import torch  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
loop = HotLoop(conf, torch.device("cuda", 0), n_frames=64, H=512, W=512, curves=True)
import bench  # noqa: E402
it0 = bench.load_scene(loop, bench.SCENE_FILE)          # the bench's frozen scene, right after its re-mesh
for it in range(it0, it0 + 4):
    loop.step(it)
torch.cuda.synchronize()
acc = {}
for it in range(it0 + 4, it0 + 4 + steps):
    loop.phase_trace = []
    start = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    start.record()
    loop.step(it)
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    h2 = time.perf_counter()
    for name, t0, t1, e0, e1 in loop.phase_trace:
        a = acc.setdefault(name, [0.0] * 4 + [0])
        a[0] += (t0 - h0) * 1e3
        a[1] += (t1 - h0) * 1e3
        a[2] += start.elapsed_time(e0)
        a[3] += start.elapsed_time(e1)
        a[4] += 1
    t = acc.setdefault("(step)", [0.0] * 4 + [0])
    t[1] += (h1 - h0) * 1e3
    t[3] += (h2 - h0) * 1e3
    t[4] += 1
print("# phase: host queues it from..to ms | device runs it from..to ms (means over %d iterations, no synchronisation inside)" % steps)
for name, (a, b, c, d, n) in acc.items():
    k = max(n, 1) / (n / steps if n >= steps else 1) if False else n
    print("%-18s host %7.2f .. %7.2f   device %7.2f .. %7.2f   (x%d per iteration)" % (name, a / n, b / n, c / n, d / n, n // steps))
