"""One GEMM shape, a few launches (a target for rocprofv3 --pmc).  python tools/gemm_one.py M N K mode [act] [reps]"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "rec-mv_amd"))
from recmv import _lib as L  # noqa: E402
from recmv import ops  # noqa: E402

M, N, K, mode = (int(v) for v in sys.argv[1:5])
act = int(sys.argv[5]) if len(sys.argv) > 5 else ops.ACT_SOFTPLUS
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda", 0)
A = torch.randn(M, K, device=dev)
B = torch.randn(N, K, device=dev) / K ** 0.5
bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
L.lib().recmv_set_gemm_mode(mode)
for _ in range(reps):
    ops.gemm_nt(A, B, bias, act, 100.0, 1.0, out=out)
torch.cuda.synchronize()
