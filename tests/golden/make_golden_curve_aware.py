"""Golden vector for `OptimGarmentNetwork.curve_aware_loss` (engineer/networks/OptimGarmentNetwork.py:787-839), from the
REAL reference method called on a stand-in `self`:

  curve_aware.npz   loss value, the per-term info entry, and the gradients the term leaves on the last garment net —
                    reference SDF net (model/network.py), reference Intersect_Free_Curve, the config's
                    pc_weight.curve_aware_weight.  trimesh (3.10.5, third party, absent) is replaced by
                    common_setup.TrimeshStandIn (its published sampling algorithm on a seeded numpy RandomState) and
                    `Tensor.cuda()` by the identity (no GPU in this container) — everything else that runs is the
                    reference's code.

    python tests/golden/make_golden_curve_aware.py
"""
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import ref_loader  # noqa: E402

ref_loader.install()
import common_setup as cs  # noqa: E402
from make_golden import save  # noqa: E402

SEED = 77


def main():
    Nref = ref_loader.ref_module("model.network")
    G = ref_loader.ref_module("engineer.utils.garment_structure")
    OGN = ref_loader.ref_module("engineer.networks.OptimGarmentNetwork")
    from recmv.hocon import ConfigFactory
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf")).get_config('loss_coarse')
    names = ['neck', 'upper_bottom', 'left_pant']
    ring = cs.curve_aware_ring()
    others = [ring * 0.5 + torch.tensor([0., 0.6, 0.]), ring * 0.4 + torch.tensor([0.1, -0.3, 0.])]
    curves = [others[0], ring, others[1]]
    ref = object.__new__(G.Intersect_Free_Curve)
    torch.nn.Module.__init__(ref)
    ref.cano2canosmpl = lambda lst, nm: [0.9 * c for c in lst]
    ref.fl_names, ref.sample_num = names, ring.shape[0]
    ref.initialize_parameters([c.clone() for c in curves])
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        ref.scale.add_(0.05 * torch.randn(ref.scale.shape, generator=g))
        ref.nx_scale.add_(0.01 * torch.randn(ref.nx_scale.shape, generator=g))
    net = cs.build_sdf(Nref.getTmpSdf)
    other_net = cs.build_sdf(Nref.getTmpSdf)
    cs.TrimeshStandIn.rng = np.random.RandomState(SEED)
    OGN.trimesh = types.SimpleNamespace(Trimesh=cs.TrimeshStandIn)
    fake = types.SimpleNamespace(conf=conf, fl_names=names, inter_free_curve=ref, garment_nets=[other_net, net],
                                 sdfShrinkRadius=0.0, info={}, garment_type='female-3-casual', isfine=False)
    ratio = {"sdfRatio": 1.0, "deformerRatio": 0.7, "renderRatio": 1.0}
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        loss = OGN.OptimGarmentNetwork.curve_aware_loss(fake, ratio)
    finally:
        torch.Tensor.cuda = real_cuda
    loss.backward()
    assert all(p.grad is None for p in other_net.parameters()) and ref.scale.grad is None
    grads = {k.replace('.', '_'): p.grad for k, p in net.named_parameters() if k in cs.SDF_GRAD_KEYS}
    print("curve_aware_loss = %.6f (info %.6f)" % (float(loss), fake.info['pc_upper_bottom_circle_loss_sdf']))
    save("curve_aware", scale=ref.scale, nx_scale=ref.nx_scale, loss=loss,
         info=np.float32(fake.info['pc_upper_bottom_circle_loss_sdf']), seed=np.int64(SEED),
         fingerprint=cs.fingerprint(net), **{"g_" + k: v for k, v in grads.items()})


if __name__ == "__main__":
    main()
