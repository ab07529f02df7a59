"""bf16x6 matrix mode: the large layer product with the weight matrix split in the loop vs split once (recmv_b3_split), hipGraph-timed.
TFLOP/s-equivalent = 2 M N K / time (the f32 product it replaces).      python tools/b3_presplit_bench.py"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402
import bench  # noqa: E402
from recmv import _lib as L, ops  # noqa: E402

dev = "cuda:0"
for mode in (0, 1):
    L.lib().recmv_set_gemm_mode(mode)
    for M, N, K in ((460800, 512, 512), (153600, 512, 512), (85000, 512, 512), (460800, 473, 512)):
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        t0 = bench._graph_time(lambda: ops.gemm_nt(A, W, b, ops.ACT_SOFTPLUS, 100.0), reps=10)[0]
        line = "mode %s  %7d x %3d x %3d   %8.1f us  %6.1f TFLOP/s-eq" % ("f32   " if mode == 0 else "bf16x6", M, N, K, t0 * 1e6, 2.0 * M * N * K / t0 / 1e12)
        if mode == 1:
            ref = ops.gemm_nt(A, W, b, ops.ACT_SOFTPLUS, 100.0)
            ops.presplit(W)
            assert torch.equal(ops.gemm_nt(A, W, b, ops.ACT_SOFTPLUS, 100.0), ref)
            t1 = bench._graph_time(lambda: ops.gemm_nt(A, W, b, ops.ACT_SOFTPLUS, 100.0), reps=10)[0]
            line += "   weights split once: %8.1f us  %6.1f TFLOP/s-eq  (x%.2f)" % (t1 * 1e6, 2.0 * M * N * K / t1 / 1e12, t0 / t1)
        print(line, flush=True)
L.lib().recmv_set_gemm_mode(0)
