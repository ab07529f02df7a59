"""bf16x6 matrix mode: is every product kernel bit-reproducible from launch to launch with other work in flight on a side stream?
(The LOOP in that mode is not reproducible run to run — DESIGN.md §9 — and this rules the products themselves out.)   python tools/bf16x6_products_repro.py"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch
from recmv import _lib as L, ops
L.lib().recmv_set_gemm_mode(1)
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(dev)
cases = []
for (M, N, K) in [(3000, 512, 512), (6000, 512, 512), (20000, 512, 512), (90000, 512, 512), (3000, 473, 512), (3000, 512, 39), (24000, 3, 512), (150000, 512, 168)]:
    A, B, b = rnd(M, K), rnd(N, K) / K ** 0.5, rnd(N)
    cases.append(("gemm_nt %dx%dx%d" % (M, N, K), lambda A=A, B=B, b=b: ops.gemm_nt(A, B, b, ops.ACT_SOFTPLUS, 100.0)))
for (M, N, K) in [(512, 512, 3000), (512, 512, 20000), (512, 512, 90000), (473, 512, 6000), (512, 39, 20000), (512, 168, 150000), (3, 512, 24000)]:
    A, B = rnd(K, M), rnd(K, N)
    cases.append(("gemm_tn %dx%dx%d" % (M, N, K), lambda A=A, B=B: ops.gemm_tn(A, B)))
for name, fn in cases:
    ref = fn()
    bad = 0
    for _ in range(6):
        # other work in flight on a side stream, to vary the timing
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            junk = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)
        out = fn()
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
    print("%-32s %s" % (name, "reproducible" if bad == 0 else "DIFFERS in %d of 6 repeats" % bad), flush=True)
