"""cProfile of a few optimiser iterations (host-side cost per python function).  python tools/host_profile.py [steps]"""
import cProfile
import pstats
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "rec-mv_amd", REPO):
    sys.path.insert(0, str(p))
import torch  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
loop = HotLoop(conf, torch.device("cuda", 0), n_frames=64, H=512, W=512, curves=True)
for it in range(3):
    loop.step(it)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(3, 3 + steps):
    loop.step(it)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
