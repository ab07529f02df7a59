"""The checkpoint layout of the reference, from the REAL `utils.save_model` (utils/utils.py:350-357) on an object assembled from the
reference's own classes under the attribute names `OptimGarmentNetwork` gives them (engineer/networks/OptimNetwork.py:58-64,
OptimGarmentNetwork.py:122-131, :3537): `sdf`, `deformer` (CompositeDeformer[MLPTranslator, LBSkinner]), `netRender`, `engine`
(Seg3dLossless), `garment_nets` (ModuleList), buffers `tmpBodyVs` / `tmpBodyFs`, `inter_free_curve`; and a dataset stand-in with the
tensors `save_model` reads.  Only the LAYOUT is kept (top-level keys, state-dict keys with shapes and dtypes): the file itself is
tens of megabytes of initial weights.

    python tests/golden/make_golden_checkpoint.py     ->  tests/golden/checkpoint_layout.json
"""
import json
import sys
import tempfile
import types
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import ref_loader  # noqa: E402

ref_loader.install()
import common_setup as cs  # noqa: E402
import project2d_case as pc  # noqa: E402


def main():
    Nref = ref_loader.ref_module("model.network")
    Dref = ref_loader.ref_module("model.Deformer")
    Rref = ref_loader.ref_module("model.RenderNet")
    Uref = ref_loader.ref_module("utils.utils")
    G = ref_loader.ref_module("engineer.utils.garment_structure")
    Sref = ref_loader.ref_module("MCAcc.seg3d_lossless")
    st = pc.state()

    class Net(torch.nn.Module):
        pass

    net = Net()
    net.sdf = Nref.getTmpSdf("cpu", 6, 0.6, 256)
    net.deformer = Dref.CompositeDeformer([cs.build_translator(Dref.MLPTranslator), cs.build_skinner(Dref.LBSkinner, Dref.batch_rodrigues)])
    net.netRender = cs.build_render(Rref.RenderingNetwork_view_norm)
    net.engine = Sref.Seg3dLossless(query_func=None, b_min=[-1., -1., -1.], b_max=[1., 1., 1.], resolutions=[(9, 11, 7), (17, 21, 13)],
                                    align_corners=False, balance_value=0.0, use_cuda_impl=False, faster=False)
    net.garment_nets = torch.nn.ModuleList([Nref.getTmpSdf("cpu", 6, 0.6, 256) for _ in range(2)])
    net.register_buffer('tmpBodyVs', st['body_v'])
    net.register_buffer('tmpBodyFs', st['body_f'])
    curve = object.__new__(G.Intersect_Free_Curve)
    torch.nn.Module.__init__(curve)
    curve.cano2canosmpl = lambda lst, nm: [0.9 * c for c in lst]
    curve.fl_names, curve.sample_num = list(pc.NAMES), pc.S
    curve.initialize_parameters([c.clone() for c in st['curves']])
    net.inter_free_curve = curve
    F = 7
    dataset = types.SimpleNamespace(
        camera_params={'focal_length': torch.ones(2), 'princeple_points': torch.ones(2), 'cam2world_coord_quat': torch.tensor([0., 0., 0., 1.]),
                       'world2cam_coord_trans': torch.zeros(3)},
        poses=torch.zeros(F, 24, 3), trans=torch.zeros(F, 3), shape=torch.zeros(10), conds=[torch.zeros(F, 384), torch.zeros(F, 256)])
    with tempfile.TemporaryDirectory() as tmp:
        Uref.save_model(str(Path(tmp) / 'latest.pth'), 3, net, dataset)
        saved = torch.load(str(Path(tmp) / 'latest.pth'), map_location='cpu')
    layout = {'top_level': {k: (list(v.shape) if torch.is_tensor(v) else type(v).__name__) for k, v in saved.items() if k != 'model_state_dict'},
              'model_state_dict': {k: [list(v.shape), str(v.dtype)] for k, v in saved['model_state_dict'].items()}}
    (HERE / 'checkpoint_layout.json').write_text(json.dumps(layout, indent=1, sort_keys=True))
    print("wrote checkpoint_layout.json: %d state-dict keys, top level %s" % (len(layout['model_state_dict']), sorted(layout['top_level'])))


if __name__ == "__main__":
    main()
