"""257^3 extraction timing split (Seg3dLossless query + MC) with a per-phase breakdown — run on the GPU box."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch
from recmv.MCAcc import Seg3dLossless
from recmv.model import getTmpSdf
dev = "cuda:0"
torch.manual_seed(0)
sdf = getTmpSdf(dev, 6)
npts = []
def query(points):
    npts.append(points.numel() // 3)
    with torch.no_grad():
        return sdf.forward(points.reshape(-1, 3), 1.0, features=False).reshape(1, 1, -1)
for use_hip in (True, False):
    eng = Seg3dLossless(query_func=query, b_min=[-1, -1, -1], b_max=[1, 1, 1], resolutions=[(33,)*3, (65,)*3, (129,)*3, (257,)*3],
                        align_corners=False, balance_value=0.0, use_cuda_impl=use_hip, faster=False).to(dev)
    vol = eng.forward(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        npts.clear()
        t0 = time.perf_counter(); v = eng.forward(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("device route" if use_hip else "volume route", "median %.2f ms" % (ts[2] * 1e3), "queries", npts, "sum", sum(npts))
    if use_hip: ref = v
    else: print("routes agree:", torch.equal(ref, v), float((ref - v).abs().max()))
# query-only cost: the same point counts through the net
pts = [torch.randn(1, n, 3, device=dev) * 0.5 for n in npts]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    for p in pts: query(p)
torch.cuda.synchronize(); print("MLP queries alone: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
