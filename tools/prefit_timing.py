"""Time the SDF pre-fit (HotLoop.initializeSDF, OptimGarmentNetwork.py:387-443 of the reference) on the GPU at the
reference's sizes: a 6890-vertex SMPL template, 5000-point batches (so 5000 + 1890 surface points per epoch, each with its
7/6 x scattered points for the eikonal term), Adam 5e-3.  Prints ms per epoch, the matrix FLOP rate of the value + ∇ₓ jets with
their backward, and the projected time of the reference's default 1200 epochs x 3 nets.

    python tools/prefit_timing.py [--epochs 40] [--verts 6890]
"""
import argparse
import sys
import tempfile
import time
import types
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "rec-mv_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=40)
    ap.add_argument('--verts', type=int, default=6890)
    args = ap.parse_args()
    from recmv.loop import HotLoop
    from recmv.model import getTmpSdf
    dev = 'cuda:0'
    torch.manual_seed(0)
    net = getTmpSdf(dev, 6)
    d = torch.nn.functional.normalize(torch.randn(args.verts, 3, device=dev), dim=1)
    vs, ns = 0.5 * d, d
    opt = torch.optim.Adam([{"params": net.parameters(), "lr": 0.005, "weight_decay": 0}])
    sche = torch.optim.lr_scheduler.StepLR(opt, 500, 0.5)
    fake = types.SimpleNamespace()
    with tempfile.TemporaryDirectory() as tmp:
        name = str(Path(tmp) / 'initial_sdf_idr_6_0.pth')
        HotLoop.initializeSDF(fake, net, opt, sche, 5000, 3, dev, vs, ns, True, name, log=None)             # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        HotLoop.initializeSDF(fake, net, opt, sche, 5000, args.epochs, dev, vs, ns, True, name, log=None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    rows = args.verts + sum(n + n // 6 for n in (min(5000, args.verts), max(args.verts - 5000, 0)))   # surface + scattered
    # per row: value pass (1x) + tangent pass for the three input directions (3x) forward, about twice that backward
    flop = rows * 3.93e6 * 4 * 3
    with torch.no_grad():
        f = net(vs, -1).abs().mean().item()
    print("pre-fit: %d surface points, %d rows per epoch: %.2f ms / epoch  (~%.1f TFLOP/s of MLP work)"
          % (args.verts, rows, 1e3 * dt / args.epochs, flop * args.epochs / dt / 1e12))
    print("mean |f| on the surface after %d epochs: %.4f" % (args.epochs + 3, f))
    print("reference default, 1200 epochs x 3 nets at this rate: %.1f s" % (1200 * 3 * dt / args.epochs))


if __name__ == '__main__':
    main()
