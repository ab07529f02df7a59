"""Test entry: `bench.main` — launcher, rank set-up, timed region, JSON line — on the CPU port of the kernels.

TEST INFRASTRUCTURE (tests/test_bench_launcher.py runs it as a script).  bench.py itself needs a GPU; here the handful of entry
points through which recmv reaches librecmv_hip.so are swapped from the outside (oracle/cpu_port.py) and the scene is cut down to
what host cores take in seconds, so that `--gpus 2` exercises the very code the driver's SCALE step calls: bench.launch_ranks
re-executes THIS file under torch.distributed.run with two gloo ranks.
"""
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "rec-mv_amd", REPO):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

from oracle import cpu_port  # noqa: E402

import bench  # noqa: E402

TINY = dict(n_frames=12, H=64, W=64, resolutions=[(9, 11, 7), (17, 21, 13)], skin_grid=(5, 9, 7),
            bbox=((-0.9, -1.2, -0.6), (0.9, 1.2, 0.6)))

if __name__ == "__main__":
    cpu_port.install()
    bench.main(device_type="cpu", hotloop_kw=TINY, conf_overrides={
        "train.sample_pix_num": 32,
        "train.learning_rate": 1e-6})      # Adam's first steps at 1e-4 move the SDF by more than this tiny box holds (test_loop_cpu)
