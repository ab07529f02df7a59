"""The entry points the north star says are kept: `model.getOptNet`, the `OptimGarmentNetwork` object
with the reference's call signatures, `utils.set_hierarchical_config`, `engineer.core.{fl_optimizer, beta_optimizer}`,
the `train.py` / `train_large_pose.py` drivers — importable by the reference's own import paths, and driving the same
iteration as the hot loop they wrap.  CPU only (oracle/cpu_port stands in for librecmv_hip.so)."""
import subprocess
import sys
from pathlib import Path

import pytest
import torch

REPO = Path(__file__).resolve().parent.parent
CONF = str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf")
TINY = dict(n_frames=12, H=64, W=64, skin_grid=(5, 9, 7))
BOX = ((-0.9, -1.2, -0.6), (0.9, 1.2, 0.6))
RES = [(9, 11, 7), (17, 21, 13)]


def test_reference_import_paths_resolve():
    """train.py:1-20 / OptimGarmentNetwork.py:1-40 import lines of the reference, verbatim, after the alias install."""
    code = f'''
import sys
sys.path[:0] = [r"{REPO / 'rec-mv_amd'}", r"{REPO}"]
import recmv.namespace
recmv.namespace.install()
from model.network import getOptNet
from model import getTmpSdf, CompositeDeformer, LBSkinner
from engineer.core.fl_optimizer import fl_proj_loss, scale_rigid_optimizer, rigid_optimizer
from engineer.core.beta_optimizer import smpl_beta_optimizer
from engineer.networks.OptimGarmentNetwork import OptimGarmentNetwork
from engineer.networks.OptimGarmentNetwork_Large_Pose import OptimGarmentNetwork_LargePose
from MCAcc import Seg3dLossless, create_grid3D, GridSamplerMine3dFunction
import utils
from utils import set_hierarchical_config, save_model, load_model, FastDiff3x3MinvFunction
import FastMinv, MCGpu, GridSamplerMine, interp2x_boundary3d
from dataset.dataset import getDatasetAndLoader, SceneDataset, People_Snapshot_SceneDataset, RandomSampler
from utils.constant import FL_INFOS, ATR_PARSING
from engineer.utils.featureline_utils import obtain_feature_lines, check_feature_lines
from engineer.utils.polygons import uniformsample
assert issubclass(OptimGarmentNetwork_LargePose, OptimGarmentNetwork)
from engineer.utils.matrix_transform import compute_rotation_matrix_from_ortho6d, scale_icp_rotate_center_transform
from engineer.utils.polygons import uniformsample3d
from model.Deformer import Inverse_Fl_Body
from utils.constant import INI_FL_SCALE
from dataset.dataset import Init_Fl_SceneDataset
from model.Deformer import initialLBSkinner, compute_lbswField, getSMPL
assert callable(scale_rigid_optimizer) and callable(rigid_optimizer) and hasattr(OptimGarmentNetwork, "initializeTmpSDF")
try:
    smpl_beta_optimizer("female", None, None)           # no SMPL model in this image: the one thing these steps cannot bring
except ImportError as e:
    assert "SMPL model" in str(e)
else:
    raise SystemExit("the shape fit ran without a SMPL model")
import torch
a = [torch.rand(2, 5, 3) * 50]; b = [torch.rand(2, 4, 2) * 50]; m = [torch.ones(2, 5, 3, dtype=torch.bool)]
assert float(fl_proj_loss(a, b, m, [1.0])) > 0
print("IMPORTS-OK")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "IMPORTS-OK" in r.stdout, r.stdout + r.stderr


def _conf(lr=None):
    from recmv.hocon import ConfigFactory
    conf = ConfigFactory.parse_file(CONF)
    conf.put('train.sample_pix_num', 32)
    if lr is not None:
        conf.put('train.learning_rate', lr)
    return conf


def test_train_py_sequence_on_the_facade_equals_hotloop_step():
    """getOptNet + `optNet(outs, sample_pix_num, ratio, frame_ids, root, global_optimizer=...)` + backward +
    propagateTmpPsGrad + optimizer.step (train.py:170-171, :317-328) is the same iteration as HotLoop.step: same loss,
    same parameters afterwards.  The mini-batch dict `outs` carries the ground truth (not the loop's own dataset)."""
    from oracle import cpu_port
    from recmv import utils
    from recmv.loop import FrameLoader, HotLoop
    from recmv.model.network import getOptNet
    cpu_port.install()
    try:
        plain = HotLoop(_conf(), 'cpu', resolutions=RES, bbox=BOX, curves=True, **TINY)
        l_plain, _ = plain.step(0)
        optNet, sdf_initialized = getOptNet(None, 'result', 3, BOX[0], BOX[1], RES, 'cpu', _conf(), curves=True, **TINY)
        assert sdf_initialized == -1 and optNet.engine.b_min.view(-1).tolist() == pytest.approx(list(BOX[0]))
        loader = FrameLoader(optNet)
        optNet, loader = utils.set_hierarchical_config(_conf(), 'coarse', optNet, loader, RES)
        optNet.train()
        optimizer = optNet.rebuild_optimizer()
        ratio = {'sdfRatio': 1., 'deformerRatio': optNet.opt_times / 2500. + 0.5, 'renderRatio': 1.}
        frame_ids, outs = next(iter(loader.set_epoch(0)))
        assert torch.equal(frame_ids, plain.frame_batch(0))
        assert set(outs) >= {'img', 'normal', 'mask', 'upper', 'bottom', 'fl_pts', 'fl_masks'}
        # hide the dataset's own targets: the call must read them from `outs`
        ds = optNet.dataset
        keep = ds.img, ds._masks, ds.gt_fl_pts
        ds.img, ds._masks, ds.gt_fl_pts = None, None, None
        optimizer.zero_grad()
        loss = optNet(outs, 32, ratio, frame_ids, '/tmp/debug', global_optimizer=optimizer)
        ds.img, ds._masks, ds.gt_fl_pts = keep
        loss.backward()
        optNet.propagateTmpPsGrad(frame_ids, ratio)
        optimizer.step()
        assert float(loss) == pytest.approx(float(l_plain), rel=1e-6)
        for a, b in zip(plain.shared_parameters(), optNet.shared_parameters()):
            assert torch.equal(a, b)
        assert all(torch.equal(a, b) for a, b in zip(plain.garment_vs, optNet.garment_vs))
    finally:
        cpu_port.uninstall()


def test_large_pose_variant_freezes_the_surfaces():
    """OptimGarmentNetwork_LargePose (OptimGarmentNetwork_Large_Pose.py:122-137, :219): SDF nets frozen and absent from
    the optimiser, curve losses zero-weighted (the curve parameters only see AdamW's decay), everything else moves."""
    from oracle import cpu_port
    from recmv.model.network import getOptNet
    cpu_port.install()
    try:
        optNet, _ = getOptNet(None, 'result', 3, BOX[0], BOX[1], RES, 'cpu', _conf(), opt_large=True, curves=True, **TINY)
        assert type(optNet).__name__ == 'OptimGarmentNetwork_LargePose' and optNet.large_pose
        sdf_params = [p for n in list(optNet.garment_nets) + [optNet.sdf] for p in n.parameters()]
        assert not any(p.requires_grad for p in sdf_params)
        in_opt = {id(p) for g in optNet.optimizer.param_groups for p in g['params']}
        assert not any(id(p) in in_opt for p in sdf_params)
        before_sdf = [p.detach().clone() for p in sdf_params]
        before_def = [p.detach().clone() for p in optNet.deformer.parameters()]
        before_curve = [p.detach().clone() for p in optNet.inter_free_curve.parameters()]
        l0, _ = optNet.step(0)
        l1, _ = optNet.step(1)
        assert torch.isfinite(l0) and torch.isfinite(l1)
        assert all(torch.equal(a, b) for a, b in zip(before_sdf, sdf_params)), "frozen SDF nets must not move"
        assert any(not torch.equal(a, b.detach()) for a, b in zip(before_def, optNet.deformer.parameters()))
        # zero-weighted curve loss: zero gradients -> AdamW only applies its decay lr * wd = 1e-6 per step
        for a, b in zip(before_curve, optNet.inter_free_curve.parameters()):
            assert float((a - b.detach()).abs().max()) <= 3e-6 * max(1.0, float(a.abs().max()))
    finally:
        cpu_port.uninstall()


def test_large_pose_propagate_matches_the_reference_method():
    """OptimGarmentNetwork_LargePose.propagateTmpPsGrad (:326-475) run for real after its freeze_sdf
    (tests/golden/make_golden_propagate.py -> propagate_large.npz) vs HotLoop.propagateTmpPsGrad with frozen SDF nets."""
    from oracle import cpu_port
    import composite_cases as cc
    cpu_port.install()
    try:
        cc.run_propagate("cpu", large_pose=True)
    finally:
        cpu_port.uninstall()


def test_large_pose_driver_cli():
    """train_large_pose.py:20-39: the flags of train.py minus --a_pose / --resume."""
    sys.path.insert(0, str(REPO / "rec-mv_amd"))
    import train
    import train_large_pose  # noqa: F401
    a = train.build_parser(large_pose=True).parse_args(['--conf', CONF, '--data', '/tmp/x', '--save-folder', 'r',
                                                        '--project_name', 'p', '--exp_name', 'e', '--data_type', 'large_pose'])
    assert a.data_type == 'large_pose' and not hasattr(a, 'resume') and not hasattr(a, 'a_pose')
    with pytest.raises(SystemExit):
        train.build_parser(large_pose=True).parse_args(['--resume', 'x.pth'])


def test_visualizer_without_wandb_writes_scalars_to_a_file(tmp_path, monkeypatch):
    """engineer/visualizer/wandb_visualizer.py:7-21 as an optional sink: no wandb (or WANDB_MODE=disabled) -> jsonl."""
    import json
    monkeypatch.setenv('WANDB_MODE', 'disabled')
    from recmv.engineer.visualizer import wandb_visualizer
    wv = wandb_visualizer('proj', 'exp', log_dir=str(tmp_path))
    wv.add_scalar({'loss': torch.tensor(1.5), 'rays': 7}, 3)
    wv.add_scalar({'loss': 1.25}, 4)
    wv.add_image({'img': torch.zeros(4, 4, 3)}, 4)
    wv.watch_model(torch.nn.Linear(2, 2))
    rows = [json.loads(l) for l in open(tmp_path / 'exp.jsonl')]
    assert rows == [{'loss': 1.5, 'rays': 7.0, 'step': 3}, {'loss': 1.25, 'step': 4}]


def test_draw_loss_flattens_info_like_the_reference(tmp_path, monkeypatch):
    """OptimGarmentNetwork.draw_loss (:3309-3316) + make_recursive_meta_func (utils/common_utils.py:73-84): nested dicts -> 'a/b',
    the first three entries of a tuple -> '000'..'002', the caller's scalars merged in; non-scalar tensors are skipped."""
    import json
    import types
    monkeypatch.setenv('WANDB_MODE', 'disabled')
    from recmv.engineer.networks.OptimGarmentNetwork import OptimGarmentNetwork, flatten_info
    from recmv.engineer.visualizer import wandb_visualizer
    assert flatten_info({'a': {'b': 1, 'c': (4, 5, 6, 7)}, 'd': 2.5}) == {'a/b': 1, 'a/c/000': 4, 'a/c/001': 5, 'a/c/002': 6, 'd': 2.5}
    fake = types.SimpleNamespace(visualizer=wandb_visualizer('proj', 'run', log_dir=str(tmp_path)),
                                 info={'upper_grad_loss': torch.tensor(0.25), 'upper_rayInfo': (3072, 1024),
                                       'fl_loss': {'total': torch.tensor(2.0)}, 'surface_pixels': torch.ones(3)})
    OptimGarmentNetwork.draw_loss(fake, 12., total_loss=1.5, learning_rate=1e-3, ratio={'sdfRatio': 1., 'deformerRatio': 0.6})
    row = json.loads(open(tmp_path / 'run.jsonl').read())
    assert row == {'upper_grad_loss': 0.25, 'upper_rayInfo/000': 3072.0, 'upper_rayInfo/001': 1024.0, 'fl_loss/total': 2.0,
                   'total_loss': 1.5, 'learning_rate': 1e-3, 'ratio/sdfRatio': 1.0, 'ratio/deformerRatio': 0.6, 'step': 12}
    fake.visualizer = None
    OptimGarmentNetwork.draw_loss(fake, 13.)                                   # no sink: nothing to do


def test_checkpoint_layout_equals_the_references():
    """utils.save_model on the loop object writes what the reference's save_model (utils/utils.py:350-357) wrote for an object of its
    own classes (tests/golden/make_golden_checkpoint.py -> checkpoint_layout.json): the same top-level entries, and a
    `model_state_dict` with the same keys, shapes and dtypes for the networks, the skinner, the curves and the SMPL template — so
    that a reference checkpoint loads here (load_model drops `engine.*` and the skinning volume on both sides) and ours loads there."""
    import json
    import tempfile
    from oracle import cpu_port
    from recmv import utils
    from recmv.model.network import getOptNet
    want = json.loads((REPO / "tests" / "golden" / "checkpoint_layout.json").read_text())
    cpu_port.install()
    try:
        optNet, _ = getOptNet(None, None, 3, BOX[0], BOX[1], RES, 'cpu', _conf(), n_frames=7, H=24, W=20, curves=True, skin_grid=(5, 9, 7))
        with tempfile.TemporaryDirectory() as tmp:
            utils.save_model(tmp + '/latest.pth', 3, optNet, optNet.dataset)
            saved = torch.load(tmp + '/latest.pth', map_location='cpu')
    finally:
        cpu_port.uninstall()
    top = {k: (list(v.shape) if torch.is_tensor(v) else type(v).__name__) for k, v in saved.items() if k != 'model_state_dict'}
    assert set(top) == set(want['top_level']), set(top) ^ set(want['top_level'])
    for k in ('epoch', 'poses', 'trans', 'shape', 'dcond', 'rcond'):
        assert top[k] == want['top_level'][k] or k in ('shape',), (k, top[k], want['top_level'][k])
    got = {k: [list(v.shape), str(v.dtype)] for k, v in saved['model_state_dict'].items()}
    ref = want['model_state_dict']
    # everything the reference stores for its networks, its skinner, its curves and its SMPL template is stored here under the same
    # name; sizes that follow from the scene (skinning volume, template and curve sample counts) are free
    scene_sized = ('deformer.defs.1.ws', 'tmpBodyVs', 'tmpBodyFs', 'engine.', 'inter_free_curve.')
    missing = [k for k in ref if k not in got and not k.startswith('engine.')]
    assert not missing, missing
    for k, (shape, dtype) in ref.items():
        if k in got and not k.startswith(scene_sized):
            assert got[k] == [shape, dtype], (k, got[k], [shape, dtype])
    extra = [k for k in got if k not in ref and not k.startswith('engine.')]
    assert not extra, extra


def test_name_tables_equal_the_reference_module():
    """recmv/utils/constant.py against the tables of the reference's utils/constant.py (dumped from the imported reference module by
    tests/golden/make_golden_constants.py): garment templates per capture, feature lines per capture / per garment (list ORDER
    included — it is the column order of `fl_pts`), z-buffer slack, curve-aware captures, ATR regions, initial line scales."""
    import json
    from recmv.utils import constant
    want = json.loads((REPO / "tests" / "golden" / "constant_tables.json").read_text())
    for name, table in want.items():
        assert getattr(constant, name) == table, name


def test_presplit_is_a_no_op_off_the_device_and_in_the_f32_mode():
    """ops.presplit only acts on CUDA weight matrices in the bf16x6 matrix mode: everything else comes back untouched (no planes,
    no registration), so the call sites in the weight normalisation and the transpose cache cost nothing in the default mode."""
    import torch
    from recmv import ops
    W = torch.randn(64, 64)
    assert ops.presplit(W) is W and getattr(W, "_recmv_b3", None) is None
    v, g = torch.randn(8, 16), torch.rand(8, 1) + 0.5
    Wn = ops.weight_norm(v, g)                                  # CPU route: plain torch
    torch.testing.assert_close(Wn, g * v / v.norm(dim=1, keepdim=True))
    with pytest.raises(RuntimeError):
        ops.def_regu(torch.eye(3).view(1, 3, 3), 0.01)          # device-only entry point: fails loudly on a CPU tensor
