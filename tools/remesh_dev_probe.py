"""Deviation table of the whole-iteration drivers (tests/forward_case.py) on the GPU with the row-tile MLP passes on / off."""
import os
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO), str(REPO / "tests"), str(REPO / "tests" / "golden")]
import torch  # noqa: E402
import composite_cases as cc  # noqa: E402
import forward_case as fwc  # noqa: E402

for rows in ("0", "1"):
    os.environ["RECMV_MLP_ROWS"] = rows
    for name, kw in (("forward", {}), ("forward_remesh", dict(remesh=True, inputs=cc.load("forward")))):
        with cc.host_draws():
            w = fwc.run(cc.load(name), "cuda:0", rtol=1.0, rtol_loss=1.0, rtol_grad=1.0, rtol_cam=1.0, **kw)
        big = {k: "%.2e" % v for k, v in w.items() if v > 2e-5}
        print("rows=%s %s: %s" % (rows, name, big), flush=True)
