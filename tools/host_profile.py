"""Where the HOST spends an iteration: cProfile over a few steps of the loop on the bench's frozen scene (no synchronisation added;
the loop's own waits — ray counts, root-finder marks — are in).  Functions by own time and by cumulative time.
    python tools/host_profile.py [steps]"""
import cProfile
import pstats
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402

import bench  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
loop = HotLoop(conf, dev, stage="coarse", curves=True, **bench.HOTLOOP_KW)
it = bench.load_scene(loop, bench.SCENE_FILE)
for n in range(4):
    loop.step(it + n)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for n in range(4, 4 + steps):
    loop.step(it + n)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
print("# %d steps; times below are totals over them" % steps)
st.sort_stats("tottime").print_stats(60)
st.sort_stats("cumulative").print_stats(70)
