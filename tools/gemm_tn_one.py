"""dW-shaped product (C = A^T B, A [K, M], B [K, N]) timed over a hipGraph of launches: python tools/gemm_tn_one.py K [M N]"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402
import bench  # noqa: E402
from recmv import ops  # noqa: E402

for K in [int(v) for v in sys.argv[1].split(",")]:
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    A = torch.randn(K, M, device="cuda:0")
    B = torch.randn(K, N, device="cuda:0")
    ref = (A.double().t() @ B.double())
    out = ops.gemm_tn(A, B)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    sec, how = bench._graph_time(lambda: ops.gemm_tn(A, B), reps=10, trips=4)
    print("K=%7d %dx%d  %8.1f us  %6.1f TFLOP/s  %.3f of peak  (incl. reduction pass; %s)  rel err %.1e  checksum %.6e" % (
        K, M, N, sec * 1e6, 2.0 * M * N * K / sec / 1e12, 2.0 * M * N * K / sec / 157.3e12, how, err, float(out.double().sum())))
