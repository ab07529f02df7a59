"""aten-op census of one optimiser iteration (host side): which torch ops the iteration issues and how often.
python tools/op_count.py [steps]"""
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "rec-mv_amd", REPO):
    sys.path.insert(0, str(p))
import torch  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
loop = HotLoop(conf, torch.device("cuda", 0), n_frames=64, H=512, W=512, curves=True)
for it in range(3):
    loop.step(it)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    for it in range(3, 3 + steps):
        loop.step(it)
    torch.cuda.synchronize()
rows = prof.key_averages()
rows = sorted(rows, key=lambda r: -r.count)
print(f"{'op':<60} {'calls/step':>10} {'self cpu ms/step':>18}")
for r in rows[:70]:
    print(f"{r.key[:60]:<60} {r.count / steps:>10.1f} {r.self_cpu_time_total / steps / 1e3:>18.3f}")

print()
print("# by self CPU time")
for r in sorted(prof.key_averages(), key=lambda r: -r.self_cpu_time_total)[:25]:
    print(f"{r.key[:60]:<60} {r.count / steps:>10.1f} {r.self_cpu_time_total / steps / 1e3:>18.3f}")
print()
print("# slow host-blocking calls (> 100 us): duration us, shapes, innermost python frames")
n = 0
for e in prof.events():
    if e.name in ("aten::empty", "aten::_local_scalar_dense", "aten::nonzero", "aten::copy_") and e.self_cpu_time_total > 100:
        n += 1
        if n <= 60:
            st = [f for f in (e.stack or []) if "recmv" in f or "tools" in f][:3]
            print(f"{e.name[6:]:<20} {e.self_cpu_time_total:9.0f} {e.input_shapes} {st}")
print("slow calls per step:", n / steps)
