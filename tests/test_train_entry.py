"""train.py entry point and checkpoint layout (host logic only: no GPU, no compute)."""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "rec-mv_amd"))

import train  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402

CONF = str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf")


def test_cli_flags_of_the_reference_are_accepted():
    """train.py:21-39 of the reference: every flag keeps its name and arity."""
    a = train.build_parser().parse_args([
        '--gpu-ids', '0', '1', '--conf', CONF, '--data', '/tmp/x', '--model-rm-prefix', 'sdf.', 'deformer.',
        '--sdf-model', 'sdf.pth', '--save-folder', 'result', '--project_name', 'p', '--exp_name', 'e',
        '--data_type', 'people_snapshot', '--a_pose', '--curve_sampling', '2', '--resume', 'latest.pth'])
    assert a.gpu_ids == [0, 1] and a.model_rm_prefix == ['sdf.', 'deformer.'] and a.a_pose and a.curve_sampling == 2
    assert a.conf == CONF and a.save_folder == 'result' and a.resume == 'latest.pth' and a.sdf_model == 'sdf.pth'


def test_stage_schedule_and_resumed_opt_times():
    conf = ConfigFactory.parse_file(CONF)
    med, fine = conf.get_int('train.medium.start_epoch'), conf.get_int('train.fine.start_epoch')
    assert train.stage_of_epoch(conf, 0) == 'coarse' and train.stage_of_epoch(conf, med - 1) == 'coarse'
    assert train.stage_of_epoch(conf, med) == 'medium' and train.stage_of_epoch(conf, fine) == 'fine'
    # train.py:250-260 with F=64, batch 3/2/1, resume after epoch `fine`
    t = train.resumed_opt_times(conf, 64, fine)
    assert t == 22 * (med - 0) + 32 * (fine - med) + 64 * (fine - med + 1)


class _FakeDataset:
    def __init__(self):
        self.frame_num = 4
        self.poses = torch.zeros(4, 24, 3, requires_grad=True)
        self.trans = torch.zeros(4, 3, requires_grad=True)
        self.shape = torch.zeros(1, 10)
        self.conds = [torch.zeros(4, 384, requires_grad=True), torch.zeros(4, 256, requires_grad=True)]
        self.camera_params = {'focal_length': torch.ones(1, 2, requires_grad=True), 'princeple_points': torch.ones(1, 2),
                              'cam2world_coord_quat': torch.tensor([[0., 0., 0., 1.]]),
                              'world2cam_coord_trans': torch.zeros(1, 3, requires_grad=True)}


class _FakeNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.sdf = torch.nn.Linear(3, 2)
        self.engine = torch.nn.Linear(1, 1)
        self.deformer = torch.nn.ModuleDict({'defs': torch.nn.ModuleList([torch.nn.Linear(2, 2), torch.nn.Module()])})
        self.deformer['defs'][1].register_buffer('ws', torch.ones(3))


def test_checkpoint_layout_and_load_rules(tmp_path):
    """utils/utils.py:350-420: key layout of the file; load drops engine.* and the skinner volume, removes prefixes,
    substitutes the SDF net, restores the per-frame tensors with their requires_grad."""
    from recmv.utils import load_model, save_model
    net, ds = _FakeNet(), _FakeDataset()
    ds.poses.data.fill_(0.25)
    f = tmp_path / "latest.pth"
    save_model(str(f), 7, net, ds)
    saved = torch.load(str(f))
    assert set(saved) == {"epoch", "model_state_dict", "focal_length", "princeple_points", "cam2world_coord_quat",
                          "world2cam_coord_trans", "poses", "trans", "shape", "dcond", "rcond"}
    assert saved["epoch"] == 7 and "sdf.weight" in saved["model_state_dict"]
    net2, ds2 = _FakeNet(), _FakeDataset()
    with torch.no_grad():
        net2.engine.weight.fill_(5.0)
        net2.deformer['defs'][1].ws.fill_(9.0)
        net2.deformer['defs'][0].weight.fill_(3.0)
    sub = tmp_path / "sdf.pth"
    torch.save({'weight': torch.full((2, 3), 0.5), 'bias': torch.zeros(2)}, str(sub))
    net2, ds2, epoch = load_model(str(f), net2, ds2, 'cpu', subsdfmodel=str(sub), model_rm_prefix=['deformer.defs.0'])
    assert epoch == 7
    assert torch.all(net2.engine.weight == 5.0), "engine.* is never loaded"
    assert torch.all(net2.deformer['defs'][1].ws == 9.0), "the skinner's ws volume is never loaded"
    assert torch.all(net2.deformer['defs'][0].weight == 3.0), "removed prefix stays untouched"
    assert torch.all(net2.sdf.weight == 0.5), "sdf.* substituted from the separate file"
    assert torch.all(ds2.poses == 0.25) and ds2.poses.requires_grad and not ds2.shape.requires_grad
    assert ds2.camera_params['focal_length'].requires_grad and not ds2.camera_params['princeple_points'].requires_grad
