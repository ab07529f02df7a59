"""A/B of the sampler forward kernels on the skinning grid (24 channels, channels-last): graph-timed launches at the
surface-coherent and random point sets of tools/kernel_only.py, outputs kept for a bit-for-bit comparison between runs.

    python tools/sampler_ab.py save          # with the library built from one variant
    python tools/sampler_ab.py check         # ... from the other: timings + bit-for-bit comparison with the saved outputs
    python tools/sampler_ab.py time

(The A/B runs of profiles/r02_sampler_forward_ab.txt switched variants inside one build through a RECMV_GS_REC environment
variable that the committed library no longer has.)
"""
import os
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd")]


def main():
    from recmv import GridSamplerMine
    mode = sys.argv[1] if len(sys.argv) > 1 else "time"
    dev = "cuda:0"
    torch.manual_seed(0)
    C, D, H, W = 24, 65, 225, 129
    vol = torch.softmax(2 * torch.randn(1, C, D, H, W, device=dev), dim=1).contiguous(memory_format=torch.channels_last_3d)
    print("RECMV_GS_REC =", os.environ.get("RECMV_GS_REC", "(default)"))          # (ignored by the committed library)
    for P in (4099, 105038, 153600, 460800, 1 << 20):
        n = int(round(P ** 0.5))
        u, v = torch.meshgrid(torch.linspace(-0.9, 0.9, n, device=dev), torch.linspace(-0.9, 0.9, n, device=dev), indexing="ij")
        surf = torch.stack([u, v, 0.3 * torch.sin(3 * u) * torch.cos(2 * v)], -1).view(1, 1, 1, -1, 3).contiguous()
        rnd = (torch.rand(1, 1, 1, P, 3, device=dev) - 0.5) * 2.2
        multi = rnd.view(3, 1, 1, -1, 3)[:, :, :, : P // 3 - 1].contiguous() if P % 3 == 0 else None
        for tag, pts, volume in (("surface", surf, vol), ("random", rnd, vol), ("3 batch items", multi, vol.expand(3, -1, -1, -1, -1))):
            if pts is None:
                continue
            out = GridSamplerMine.forward(volume, pts, 0, 1)
            torch.cuda.synchronize()
            name = "/tmp/gs_%s_%d.pt" % (tag.replace(' ', '_'), P)
            if mode == "save":
                torch.save(out.cpu(), name)
            elif mode == "check":
                assert torch.equal(out.cpu(), torch.load(name)), (tag, P)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    out = GridSamplerMine.forward(volume, pts, 0, 1)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            print("  P=%8d %-14s %7.2f us / launch%s" % (pts.shape[0] * pts.shape[3], tag, e0.elapsed_time(e1) * 1e3 / 100,
                                                          "   (bit-identical to the saved run)" if mode == "check" else ""))


if __name__ == "__main__":
    main()
