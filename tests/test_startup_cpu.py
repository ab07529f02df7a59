"""The start-up stage that precedes the loop — feature-line template registration and the SDF pre-fit — against the
reference's own functions (tests/golden/make_golden_startup.py -> tests/golden/startup.npz).  CPU only: oracle/cpu_port
stands in for librecmv_hip.so; tests/test_gpu_startup.py runs the same drivers through the HIP library."""
import random
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
import capture_fixture as cf  # noqa: E402
import startup_case as sc  # noqa: E402


def load():
    return {k: torch.from_numpy(v) for k, v in np.load(HERE / "golden" / "startup.npz").items()}


def test_feature_line_transforms_match_the_reference():
    """engineer/utils/matrix_transform.py: 6-D rotation (incl. nearly parallel axes) and the five per-line transforms
    (incl. a negative scale, clamped to zero)."""
    from recmv.engineer.utils import matrix_transform as mt
    g = load()
    lines = list(torch.split(g["mt_lines"], [int(n) for n in g["mt_split"]]))
    R = mt.compute_rotation_matrix_from_ortho6d(g["mt_poses"])
    torch.testing.assert_close(R, g["mt_R"], rtol=1e-6, atol=1e-7)
    T, S = g["mt_T"], g["mt_S"]
    cat = lambda lst: torch.cat(lst, 0)
    for name, got in (("icp", mt.icp_rotate_transfrom(lines, R, T)), ("scale_icp", mt.scale_icp_rotate_transfrom(lines, R, T, S)),
                      ("center", mt.center_transform(lines, R, T)), ("icp_center", mt.icp_rotate_center_transform(lines, R, T)),
                      ("scale_icp_center", mt.scale_icp_rotate_center_transform(lines, R, T, S))):
        torch.testing.assert_close(cat(got), g["mt_" + name], rtol=1e-5, atol=1e-6, msg=lambda m: name + ": " + m)
    # meshes and bare vertex tensors are interchangeable
    meshes = [mt.FeatureLineMesh(v, torch.zeros(1, 3, dtype=torch.long)) for v in lines]
    assert torch.equal(cat(mt.center_transform(meshes, R, T)), cat(mt.center_transform(lines, R, T)))
    moved = meshes[0].update_padded(lines[0][None] + 1.)
    assert torch.equal(moved.verts_packed(), lines[0] + 1.) and moved.faces_packed() is meshes[0].faces_packed()


def test_init_fl_dataset_matches_the_reference(tmp_path):
    """dataset/dataset.py `get_init_fl_datasets` / `Init_Fl_SceneDataset`: the supervising frames, their samples, the
    loader's order under the same seeds; PeopleSnapshot captures hand over the annotated frames only."""
    from recmv.dataset import Init_Fl_SceneDataset, People_Snapshot_SceneDataset, SceneDataset
    g = load()
    root = sc.write_capture(str(tmp_path))
    torch.manual_seed(31)
    ds = SceneDataset(root, dict(sc.CONDS), cf.GARMENT_TYPE, fl_sampling=sc.FL_SAMPLING, curve_sampling=1)
    random.seed(32)
    torch.manual_seed(32)
    loader = ds.get_init_fl_datasets(3, None, 0)
    init_ds = loader.dataset
    assert isinstance(init_ds, Init_Fl_SceneDataset)
    assert [float(i) for i in init_ds.idx] == g["init_idx"].tolist()
    assert [float(len(init_ds)), float(len(loader))] == g["init_len"].tolist()
    for k in (0, 4, 7):
        fid, sample = init_ds[k]
        assert float(fid) == float(g["init_%d_fid" % k])
        # (the reference's tensor is float64 when a line is missing from the frame — its zero filler is; the loop reads float32)
        torch.testing.assert_close(sample["fl_pts"], g["init_%d_fl_pts" % k].float(), rtol=1e-6, atol=1e-5)
        assert torch.equal(sample["fl_masks"].float(), g["init_%d_fl_masks" % k])
        assert torch.equal(sample["mask"], g["init_%d_mask" % k])
    assert [float(f) for fids, _ in loader for f in fids] == g["init_order"].tolist()
    ps = People_Snapshot_SceneDataset(root, dict(sc.CONDS), cf.GARMENT_TYPE, fl_sampling=sc.FL_SAMPLING, curve_sampling=1,
                                      a_pose=False)
    ps_init = ps.get_init_fl_datasets(2, None, 0).dataset
    assert [float(i) for i in ps_init.idx] == g["init_ps_idx"].tolist()
    assert torch.equal(torch.stack([ps_init[k][1]["fl_masks"].float() for k in range(len(ps_init))]), g["init_ps_masks"])


def test_feature_line_registration_matches_the_reference(tmp_path):
    """engineer/core/fl_optimizer.py `scale_rigid_optimizer` (:111-519: 50 + 10 + 50 epochs of Adam with the body z-buffer
    visibility test) and `rigid_optimizer` (:520-677) run for real vs recmv's on the same capture: stored transform,
    registered vertices, and the re-application of a stored file."""
    from oracle import cpu_port
    cpu_port.install()
    try:
        sc.run_registration(load(), sc.write_capture(str(tmp_path)), "cpu")
    finally:
        cpu_port.uninstall()


def test_sdf_prefit_matches_the_reference_method():
    """OptimGarmentNetwork.initializeSDF (:387-443) run for real on the reference's SDF net vs HotLoop.initializeSDF:
    parameters after three epochs, with and without the normal term."""
    from oracle import cpu_port
    cpu_port.install()
    try:
        sc.run_prefit(load(), "cpu")
    finally:
        cpu_port.uninstall()
