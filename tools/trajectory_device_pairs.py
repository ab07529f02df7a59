"""Row (g): how far apart do runs of THIS implementation's own 35-iteration loop end when only the rounding of its matrix products
changes?  The device-side twin of tests/golden/make_golden_envelope.py (the reference against itself under 1..8 sgemm threads).

The loop is bit-reproducible in every configuration (tests/test_gpu_loop.py), so a second run of the same configuration says nothing;
what differs between the configurations below is only WHICH kernels form the products and therefore how they round:
  f32            f32-input MFMA everywhere (the default, `value`'s mode)
  bf16x6         every eligible product as six bf16 MFMA products of a three-way split (RECMV_GEMM_MODE=1)
  b3 mask 1/2/4  the bf16x6 kernels for one product family only (recmv_set_b3_families: 128x128 NT / 64-wide NT / TN), f32 elsewhere
Each run is compared with the reference's run (tests/golden/trajectory.npz) AND with every other device run: if the device runs end
as far from EACH OTHER as from the reference, the distance to the reference at this horizon is the map's own sensitivity to rounding
and not a difference of algorithm.

    python tools/trajectory_device_pairs.py [fixture=trajectory]         # on the GPU box; ~20 s per run
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rec-mv_amd"), os.path.join(ROOT, "tests"), ROOT]
import composite_cases as cc  # noqa: E402
import forward_case as fwc  # noqa: E402
from recmv import _lib as L  # noqa: E402

CONFIGS = [("f32", 0, 7), ("bf16x6", 1, 7), ("b3 mask 1", 1, 1), ("b3 mask 2", 1, 2), ("b3 mask 4", 1, 4)]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "trajectory"
    lib = L.lib()
    g = cc.load(name)
    inputs = cc.load("forward")
    runs = []
    for label, mode, mask in CONFIGS:
        prev = lib.recmv_set_gemm_mode(mode)
        lib.recmv_set_b3_families(mask)
        try:
            with cc.host_draws():
                out = fwc.run_trajectory(g, inputs, "cuda:0")
        finally:
            lib.recmv_set_gemm_mode(prev)
            lib.recmv_set_b3_families(7)
        runs.append(out)
        print("%-10s loss deviation from the reference at iterations 1/5/10/20/%d: %s;  canonical Chamfer to the REFERENCE: body %.2e  upper "
              "%.2e  bottom %.2e" % (label, len(out["losses"]), ["%.1e" % out["loss_rel_dev"][min(i, len(out["losses"]) - 1)]
                                                                for i in (0, 4, 9, 19, len(out["losses"]) - 1)],
                                     out["canon_body"]["chamfer_sq"], out["canon_u"]["chamfer_sq"], out["canon_b"]["chamfer_sq"]), flush=True)
    K = len(runs)
    for tag, title in (("u", "upper garment"), ("b", "bottom garment"), ("body", "body")):
        d = np.zeros((K, K))
        for i in range(K):
            for j in range(i + 1, K):
                d[i, j] = d[j, i] = fwc.chamfer_vertices(runs[i]["canon_verts"][tag], runs[j]["canon_verts"][tag])[0]
        iu = d[np.triu_indices(K, 1)]
        to_ref = [runs[i]["canon_" + tag]["chamfer_sq"] for i in range(K)]
        print("%-14s device runs AGAINST EACH OTHER (%d pairs): min %.2e median %.2e max %.2e;  against the reference: min %.2e median "
              "%.2e max %.2e" % (title, len(iu), iu.min(), np.median(iu), iu.max(), min(to_ref), float(np.median(to_ref)), max(to_ref)))
        for i in range(K):
            print("    %-10s %s" % (CONFIGS[i][0], " ".join("%.2e" % v for v in d[i])))
    env = fwc.reference_envelope()
    if env is not None and name == "trajectory":
        for tag in ("u", "b"):
            m = env["canon_chamfer_" + tag]
            iu = m[np.triu_indices(m.shape[0], 1)]
            print("reference against itself, 1..8 sgemm threads (%s): min %.2e median %.2e max %.2e" % (tag, iu.min(), np.median(iu), iu.max()))


if __name__ == "__main__":
    main()
