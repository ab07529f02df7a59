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


def test_skinner_with_the_references_single_extent_runs_the_fused_kernels():
    """A skinner baked by the reference normalises canonical points with ONE extent (model/Deformer.py:609, 342-352); the
    fused HIP forward / VJP take it per axis: same result as the module path (sampler + blend), values and gradients."""
    import common_setup as cs
    from recmv.model import LBSkinner
    sk = LBSkinner(**dict(cs.skinner_args(), bbox_extend=torch.tensor(2.4))).to("cuda:0")
    g = torch.Generator().manual_seed(2)
    pts = (0.35 * torch.randn(3, 500, 3, generator=g)).to("cuda:0")
    poses, trans = (t.to("cuda:0") for t in cs.poses_trans(3, seed=5))
    a = pts.clone().requires_grad_(True)
    b = pts.clone().requires_grad_(True)
    fused = sk(a, [poses, trans])
    plain = sk(b, [poses, trans], jet=True)                       # (the jet flag keeps the call on the module path)
    torch.testing.assert_close(fused, plain, rtol=1e-5, atol=1e-6)
    w = torch.randn(fused.shape, generator=g).to("cuda:0")
    (fused * w).sum().backward()
    (plain * w).sum().backward()
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-5)


def test_skinner_baking_on_the_gpu_and_the_fused_kernels_on_the_baked_skinner():
    """compute_lbswField / initialLBSkinner on the device against the reference functions, then the baked skinner (one
    normalisation extent) through the fused HIP forward against the points the REFERENCE's skinner posed."""
    g = load()
    sk = sc.run_skinner_baking(g, "cuda:0")
    pts = g['bake_pts'].to("cuda:0").requires_grad_(True)                   # (grad enabled: the fused path)
    got = sk(pts, [g['bake_poses'].to("cuda:0"), g['bake_trans'].to("cuda:0")])
    torch.testing.assert_close(got.detach().cpu(), g['bake_posed'], rtol=1e-4, atol=2e-6)


def test_smpl_shape_fit_on_the_gpu_matches_the_reference(tmp_path):
    worst = sc.run_beta_fit(load(), sc.write_joint_capture(str(tmp_path)), "cuda:0")
    print("shape fit, largest relative deviations:", {k: "%.1e" % v for k, v in worst.items()})
