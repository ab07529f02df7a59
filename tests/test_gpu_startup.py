"""The start-up stage on the MI355X: the same drivers as tests/test_startup_cpu.py (tests/startup_case.py), through
librecmv_hip.so — the fused skinner, the HIP mesh rasteriser behind the body visibility test, the SDF net's jet pass — against
what the REFERENCE's functions produced on the same inputs (tests/golden/startup.npz)."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent
for p in (str(REPO / "rec-mv_amd"), str(REPO), str(HERE), str(HERE / "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
import startup_case as sc  # noqa: E402

pytestmark = pytest.mark.gpu


def load():
    return {k: torch.from_numpy(v) for k, v in np.load(HERE / "golden" / "startup.npz").items()}


def test_feature_line_registration_on_the_gpu_matches_the_reference(tmp_path):
    """scale_rigid_optimizer (110 epochs of Adam, body z-buffer visibility from the HIP rasteriser) and rigid_optimizer: the
    stored transforms and the registered line vertices of the reference run."""
    worst = sc.run_registration(load(), sc.write_capture(str(tmp_path)), "cuda:0", rtol=2e-3)
    print("registration, largest relative deviations:", {k: "%.1e" % v for k, v in worst.items()})


def test_sdf_prefit_on_the_gpu_matches_the_reference_method():
    """HotLoop.initializeSDF with the jet pass of the HIP MLP vs the reference's double backward: same parameters after
    three epochs."""
    sc.run_prefit(load(), "cuda:0")
