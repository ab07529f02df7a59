"""The optimiser loop alone, for a kernel trace: warm-up, a 1 s pause (the marker tools/idle_gaps.py looks for), N steps.

    rocprofv3 --kernel-trace -d /tmp/prof -o run -- python tools/loop_trace.py 10
    python tools/idle_gaps.py /tmp/prof 10 > profiles/<name>.txt
"""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "rec-mv_amd", REPO):
    sys.path.insert(0, str(p))
import torch  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
import os  # noqa: E402
loop = HotLoop(conf, torch.device("cuda", 0), n_frames=64, H=512, W=512, curves=os.environ.get("RECMV_TRACE_NO_CURVES") != "1")
for it in range(3):
    loop.step(it)
torch.cuda.synchronize()
time.sleep(1.0)
t0 = time.perf_counter()
for it in range(3, 3 + steps):
    loop.step(it)
torch.cuda.synchronize()
print("ms_per_step %.2f" % ((time.perf_counter() - t0) * 1e3 / steps))
