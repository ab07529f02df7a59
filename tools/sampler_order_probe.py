"""Does the ORDER of the points matter to the sampler-path kernels at the loop's sizes?  The explicit vertices arrive in marching-cubes
key order (x-major: consecutive points walk along z inside one voxel row, ~2 per skinning cell); a Morton (Z-curve) order over the
skinning grid's cells would put a wave's 64 points into a compact block of cells that share corner records.

    python tools/sampler_order_probe.py

Times GridSamplerMine.forward / backward and the fused lbs_forward / lbs_vjp kernels (hipGraph of 20 launches, bench.py's timer) on the
bench scene's vertices (1 and 3 frames) in: MC order, Morton order, random order."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402
import bench  # noqa: E402
from recmv import GridSamplerMine, chains  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402


def morton_key(ijk):
    def spread(v):
        v = v.long() & 0x3ff
        v = (v | (v << 16)) & 0x30000ff
        v = (v | (v << 8)) & 0x300f00f
        v = (v | (v << 4)) & 0x30c30c3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(ijk[:, 0]) | (spread(ijk[:, 1]) << 1) | (spread(ijk[:, 2]) << 2)


def main():
    dev = torch.device("cuda", 0)
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    loop = HotLoop(conf, dev, **bench.HOTLOOP_KW)
    loop.step(0)
    torch.cuda.synchronize()
    sk = loop.deformer.defs[1]
    vol = sk.ws
    C = vol.shape[1]
    D, H, W = vol.shape[2:]
    verts = loop.garment_vs[0].detach()
    nps = (verts - sk.bbox_center.to(dev)) / sk.bbox_extend.to(dev) * 2.
    cell = torch.stack([((nps[:, 0] + 1) * W - 1) / 2, ((nps[:, 1] + 1) * H - 1) / 2, ((nps[:, 2] + 1) * D - 1) / 2], 1).floor().clamp(min=0)
    orders = {"MC key order": torch.arange(verts.shape[0], device=dev), "Morton order of the skinning cells": torch.argsort(morton_key(cell)),
              "random order": torch.randperm(verts.shape[0], device=dev)}
    poses, trans = 0.15 * torch.randn(3, 24, 3, device=dev), 0.01 * torch.randn(3, 3, device=dev)
    with torch.no_grad():
        A_pose, t_pose = sk._posed(poses, trans)
    grid = sk._lbs_grid()
    for frames in (1, 3):
        for name, perm in orders.items():
            v = verts[perm].repeat(frames, 1).contiguous()
            g = ((v - sk.bbox_center.to(dev)) / sk.bbox_extend.to(dev) * 2.).view(1, 1, 1, -1, 3).contiguous()
            P = v.shape[0]
            frame = (torch.arange(P, device=dev) // verts.shape[0]).contiguous()
            go = torch.randn(1, C, 1, 1, P, device=dev)
            gd = torch.randn(P, 3, device=dev)
            uniq = torch.unique(morton_key(cell[perm][:4096]).view(-1, 64), dim=1)      # (not used: cells per wave below)
            cells_per_wave = float(torch.tensor([torch.unique(morton_key(cell[perm][i:i + 64])).numel()
                                                  for i in range(0, min(P, verts.shape[0]) - 64, 6400)]).float().mean())
            t_f = bench._graph_time(lambda: GridSamplerMine.forward(vol, g, 0, 1))[0] * 1e6
            t_b = bench._graph_time(lambda: GridSamplerMine.backward(vol, g, go, 0, 1, need_grad_input=False))[0] * 1e6
            t_lf = bench._graph_time(lambda: chains.lbs_forward(v, frame, A_pose, t_pose, grid))[0] * 1e6
            t_lv = bench._graph_time(lambda: chains.lbs_vjp_input(v, frame, A_pose, grid, gd))[0] * 1e6
            print("P=%7d (%d frame%s) %-36s cells per 64 points %5.1f | sampler fwd %6.2f us (%.3f of 8 TB/s)  bwd %6.2f us (%.3f) | "
                  "lbs_forward %6.2f us  lbs_vjp %6.2f us" % (P, frames, "s" if frames > 1 else " ", name, cells_per_wave, t_f,
                                                               P * (12 + 4 * C) / t_f / 8e6, t_b, P * (24 + 4 * C) / t_b / 8e6, t_lf, t_lv),
                  flush=True)


if __name__ == "__main__":
    main()
