"""Does a 128-byte record (24 channels padded to 32 floats, every corner exactly one aligned cache line) feed the sampler faster than the
dense 96-byte one?  Forward / backward / double backward on surface-coherent points, hipGraph-timed.   python tools/sampler_pad_probe.py"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402
import bench  # noqa: E402
from recmv import GridSamplerMine  # noqa: E402

dev = "cuda:0"
C, D, H, W = 24, 65, 225, 129
base = torch.softmax(2 * torch.randn(1, C, D, H, W, device=dev), dim=1)
dense = base.contiguous(memory_format=torch.channels_last_3d)
for pad in (24, 32):
    buf = torch.zeros(1, D, H, W, pad, device=dev)
    buf[..., :C] = base.permute(0, 2, 3, 4, 1)
    vol = buf[..., :C].permute(0, 4, 1, 2, 3)              # [1,C,D,H,W], stride[1] == 1, record stride `pad`
    assert vol.stride(1) == 1 and vol.stride(4) == pad
    for P in (95154, 285462, 1 << 20, 1 << 22):
        n = int(round(P ** 0.5))
        u, v = torch.meshgrid(torch.linspace(-0.9, 0.9, n, device=dev), torch.linspace(-0.9, 0.9, n, device=dev), indexing="ij")
        surf = torch.stack([u, v, 0.3 * torch.sin(3 * u) * torch.cos(2 * v)], -1).view(1, 1, 1, -1, 3).contiguous()
        Pc = surf.shape[3]
        go = torch.randn(1, C, 1, 1, Pc, device=dev)
        gg = torch.randn(1, 1, 1, Pc, 3, device=dev)
        ref = GridSamplerMine.forward(dense, surf, 0, 1)
        assert torch.equal(GridSamplerMine.forward(vol, surf, 0, 1), ref)
        t_f = bench._graph_time(lambda: GridSamplerMine.forward(vol, surf, 0, 1))[0]
        t_b = bench._graph_time(lambda: GridSamplerMine.backward(vol, surf, go, 0, 1, need_grad_input=False))[0]
        t_d = bench._graph_time(lambda: GridSamplerMine.dbackward(None, gg, vol, surf, go, 0, 1, need_grad_input=False))[0]
        f = lambda by, t: by * Pc / t / 8e12
        print("record %3d B  P=%8d   fwd %7.1f us (%.3f)   bwd %7.1f us (%.3f)   dbwd %7.1f us (%.3f)" % (
            4 * pad, Pc, t_f * 1e6, f(12 + 4 * C, t_f), t_b * 1e6, f(24 + 4 * C, t_b), t_d * 1e6, f(36 + 8 * C, t_d)))
