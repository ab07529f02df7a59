"""Allocator behaviour of the loop: device (hipMalloc) calls per iteration, reserved bytes, cost of torch.empty."""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "rec-mv_amd", REPO):
    sys.path.insert(0, str(p))
import torch  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
loop = HotLoop(conf, torch.device("cuda", 0), n_frames=64, H=512, W=512, curves=True)
dev = torch.device("cuda", 0)


def t_empty(n=2000, shape=(3000, 512)):
    t0 = time.perf_counter()
    for _ in range(n):
        torch.empty(shape, device=dev)
    return (time.perf_counter() - t0) / n * 1e6


print("empty before any step: %.2f us" % t_empty())
prev = torch.cuda.memory_stats()
for it in range(8):
    t0 = time.perf_counter()
    loop.step(it)
    host = (time.perf_counter() - t0) * 1e3
    s = torch.cuda.memory_stats()
    print("step %d host %.1f ms  device_alloc +%d  device_free +%d  allocs +%d  reserved %.2f GB  active %.2f GB  empty now %.2f us"
          % (it, host, s["num_device_alloc"] - prev["num_device_alloc"], s["num_device_free"] - prev["num_device_free"],
             s["allocation.all.allocated"] - prev["allocation.all.allocated"], s["reserved_bytes.all.current"] / 2**30,
             s["active_bytes.all.current"] / 2**30, t_empty(300)), flush=True)
    prev = s
torch.cuda.synchronize()
print("empty after sync: %.2f us" % t_empty())
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    print("empty on a fresh side stream: %.2f us" % t_empty())
