"""Histogram of the GEMM shapes one optimiser iteration launches from Python (ops.gemm_nt / ops.gemm_tn; the C chains
of the root finder are not visible here).  python tools/gemm_shapes.py"""
import collections
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "rec-mv_amd", REPO):
    sys.path.insert(0, str(p))
import torch  # noqa: E402
from recmv import ops, _lib as L  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
loop = HotLoop(conf, torch.device("cuda", 0), n_frames=64, H=512, W=512)
for it in range(2):
    loop.step(it)
hist = collections.Counter()
o_nt, o_tn = ops.gemm_nt, ops.gemm_tn
lib = L.lib()
o_lb = lib.recmv_linear_backward


def nt(A, B, *a, **k):
    hist[("nt", A.shape[0], B.shape[0], A.shape[1])] += 1
    return o_nt(A, B, *a, **k)


def tn(A, B):
    hist[("tn", A.shape[1], B.shape[1], A.shape[0])] += 1
    return o_tn(A, B)


class LB:
    argtypes = o_lb.argtypes
    restype = o_lb.restype

    def __call__(self, *args):
        M, N, K = args[8], args[9], args[10]
        gx, gW = args[13], args[15]
        if gx is not None and getattr(gx, "value", gx):
            hist[("nt", M, K, N)] += 1
        if gW is not None and getattr(gW, "value", gW):
            hist[("tn", N, K, M)] += 1
        return o_lb(*args)


ops.gemm_nt, ops.gemm_tn = nt, tn
lib.recmv_linear_backward = LB()
steps = 3
for it in range(2, 2 + steps):
    loop.step(it)
torch.cuda.synchronize()
tot = 0.0
rows = []
for (kind, M, N, K), c in hist.items():
    fl = 2.0 * M * N * K * c / steps
    tot += fl
    rows.append((fl, kind, M, N, K, c / steps))
rows.sort(reverse=True)
print("total GFLOP/step from python-level GEMMs: %.1f" % (tot / 1e9))
for fl, kind, M, N, K, c in rows[:60]:
    print("%-3s M=%-7d N=%-5d K=%-7d  calls/step=%-6.1f GFLOP/step=%.1f" % (kind, M, N, K, c, fl / 1e9))
