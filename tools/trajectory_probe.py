"""The trajectory drivers (tests/forward_case.py run_trajectory) on a device, printing what was measured without asserting."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO), str(REPO / "tests"), str(REPO / "tests" / "golden")]
import torch  # noqa: E402
import composite_cases as cc  # noqa: E402
import forward_case as fwc  # noqa: E402

dev = sys.argv[1] if len(sys.argv) > 1 else "cuda:0"
for name in ("trajectory_short", "trajectory"):
    with cc.host_draws():
        g = cc.load(name)
        out = fwc.run_trajectory(g, cc.load("forward"), dev)
    print(name, "loss dev:", ["%.1e" % d for d in out['loss_rel_dev']])
    print(name, "rays:", out['rays'], "ref:", out['rays_ref'])
    print(name, {k: v for k, v in out.items() if k.startswith(('canon', 'explicit', 'remesh'))}, flush=True)
