"""The composite methods of the iteration ON THE GPU against the fixtures produced by the reference's own methods
(tests/golden/make_golden_{render_loss,propagate,misc,curves,curve_aware}.py): `HotLoop.surface_render_loss`,
`propagateTmpPsGrad`, `compute_garment_pc_loss`, `sample_train_ray`, `compute_fl_proj_loss`,
`fl_visible_by_body_zbuff`, `curve_aware_loss`.  On `cuda:0` these methods take the product's fused branches (MFMA
layers, jet pass, launch chains, fused LBS, HIP rasteriser) — the branches the CPU port never sees.  Same drivers and
tolerances as the CPU tests (tests/composite_cases.py); f32 tolerances are written next to each driver's defaults.
"""
import pytest

import composite_cases as cc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_surface_render_loss_on_gpu_matches_the_reference_method():
    cc.run_render_loss(DEV)


def test_propagate_tmp_ps_grad_on_gpu_matches_the_reference_method():
    cc.run_propagate(DEV)


def test_large_pose_propagate_on_gpu_matches_the_reference_method():
    cc.run_propagate(DEV, large_pose=True)


def test_compute_garment_pc_loss_on_gpu_matches_the_reference_method():
    cc.run_pc_loss(DEV)


def test_sample_train_ray_on_gpu_matches_the_reference_method():
    cc.run_sample_rays(DEV)


def test_compute_fl_proj_loss_on_gpu_matches_the_reference_method():
    cc.run_fl_proj(DEV)


def test_fl_visibility_on_gpu_matches_the_reference_method():
    cc.run_fl_visibility(DEV)


def test_curve_aware_loss_on_gpu_matches_the_reference_method():
    cc.run_curve_aware(DEV)


def test_mask_loss_on_gpu_matches_the_reference_method():
    """OptimGarmentNetwork.mask_loss (:841-981) as a whole (tests/golden/make_golden_mask_loss.py) through the HIP point
    rasteriser + compositor, the fused skinner and the MFMA layers: value, info, moved vertices, gradients."""
    import mask_loss_case as mlc
    worst = mlc.run(cc.load("mask_loss"), DEV, rtol=1e-3, rtol_grad=1e-2)
    print("mask_loss on the GPU, largest relative deviations:", {k: "%.1e" % v for k, v in worst.items() if v > 1e-6})


# The whole-method drivers added after this round's GPU budget was spent (CPU-port parity: tests/test_golden_cpu.py).  They have
# not met the device yet, so the driver's run skips them; RECMV_UNVALIDATED_GPU_TESTS=1 runs them (next round's first step).
import os  # noqa: E402

unvalidated = pytest.mark.skipif(os.environ.get("RECMV_UNVALIDATED_GPU_TESTS") != "1",
                                 reason="not yet run on an MI355X (set RECMV_UNVALIDATED_GPU_TESTS=1)")


@unvalidated
def test_project_2d_loss_on_gpu_matches_the_reference_method():
    import project2d_case as p2c
    print(p2c.run(cc.load("project2d"), DEV, rtol=1e-3, rtol_grad=1e-2))


@unvalidated
def test_one_whole_iteration_on_gpu_matches_the_reference():
    import forward_case as fwc
    with cc.host_draws():
        print(fwc.run(cc.load("forward"), DEV, rtol=1e-3, rtol_grad=2e-2))


@unvalidated
def test_one_whole_large_pose_iteration_on_gpu_matches_the_reference():
    import forward_case as fwc
    with cc.host_draws():
        print(fwc.run(cc.load("forward_large"), DEV, rtol=1e-3, rtol_grad=2e-2, large_pose=True, inputs=cc.load("forward")))


@unvalidated
def test_one_whole_iteration_with_the_remesh_inside_on_gpu_matches_the_reference():
    import forward_case as fwc
    with cc.host_draws():
        print(fwc.run(cc.load("forward_remesh"), DEV, rtol=2e-3, rtol_grad=5e-2, inputs=cc.load("forward"), remesh=True))
