"""Diagnostic: how many rays the root finder converges per iteration as the synthetic optimisation settles."""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]

from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
dev = torch.device("cuda:0")
loop = HotLoop(conf, dev, n_frames=64, H=512, W=512, stage="coarse")
t0 = time.time()
for it in range(n):
    try:
        loss, rays = loop.step(it)
    except Exception as e:
        print('FAILED at', it, repr(e)[:200], 'V', [v.shape[0] for v in loop.garment_vs], flush=True)
        raise
    if it % 10 in (0, 1, 5) or it == n - 1 or it < 8:
        sdf = [float(loop.info.get('pc_%s_loss_sdf' % k, -1)) for k in loop.garment_names]
        print(it, 'loss %.4f' % float(loss), 'rays', rays, 'converged', loop.info.get('rays_converged'),
              'mean|sdf(verts)|', ['%.4f' % s for s in sdf], 'V', [v.shape[0] for v in loop.garment_vs], 'mask', ['%.3f' % float(loop.info.get('pc_%s_mask_loss' % k, -1)) for k in loop.garment_names], 'vmax', ['%.2f' % float(v.detach().abs().max()) for v in loop.garment_vs], '%.1fs' % (time.time() - t0), flush=True)
