"""Both matrix modes of the MFMA contractions side by side: error against fp64 (the same bound for both) and time.

    python tools/gemm_modes.py [--quick]
"""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "rec-mv_amd"))
from recmv import _lib as L  # noqa: E402
from recmv import ops  # noqa: E402

DEV = torch.device("cuda", 0)
quick = "--quick" in sys.argv


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def check(out, ref, absprod, what):
    bound = 4e-7 * absprod + 1e-6
    err = (out.double() - ref).abs()
    worst = float((err / bound).max())
    print(f"    {what}: max err/bound {worst:.3f}  {'OK' if worst <= 1 else 'FAIL'}", flush=True)
    return worst <= 1


ok = True
nt_shapes = [(20000, 512, 512), (153600, 512, 512), (460800, 512, 512), (20000, 473, 512), (70000, 512, 40), (70000, 512, 168),
             (33000, 257, 512), (6144, 512, 512), (3072, 512, 512), (300, 473, 512)]
if quick:
    nt_shapes = nt_shapes[:2]
for M, N, K in nt_shapes:
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, K, generator=g) * torch.logspace(-3, 2, M).view(-1, 1)).to(DEV)
    B = (torch.randn(N, K, generator=g) / np.sqrt(K)).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    rows = torch.randint(0, M, (2048,), generator=g).to(DEV)          # fp64 reference on a row sample
    ref = A[rows].double() @ B.double().t() + bias.double()
    absprod = A[rows].abs().double() @ B.abs().double().t()
    out = torch.empty(M, N, device=DEV)
    print(f"gemm_nt M={M} N={N} K={K}")
    for mode in (0, 1):
        L.lib().recmv_set_gemm_mode(mode)
        ops.gemm_nt(A, B, bias, out=out)
        ok &= check(out[rows], ref, absprod, f"mode {mode}")
        t = timeit(lambda: ops.gemm_nt(A, B, bias, ops.ACT_SOFTPLUS, 100.0, 1.0, out=out))
        print(f"    mode {mode}: +bias+softplus {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s", flush=True)

# activation-gradient variant: C = (G . softplus'(Y)) B^T
lib = L.lib()
for M, N, K in ([(20000, 512, 512)] if quick else [(20000, 512, 512), (153600, 512, 512), (6144, 512, 512), (70000, 39, 512)]):
    g = torch.Generator().manual_seed(M)
    G = torch.randn(M, K, generator=g).to(DEV)
    Y = (torch.rand(M, K, generator=g) * 0.05).to(DEV)
    B = (torch.randn(N, K, generator=g) / np.sqrt(K)).to(DEV)
    rows = torch.randint(0, M, (2048,), generator=g).to(DEV)
    d = -torch.expm1(-100.0 * Y[rows].double())
    ref = (G[rows].double() * d * 0.5) @ B.double().t()
    absprod = (G[rows].double() * d * 0.5).abs() @ B.abs().double().t()
    out = torch.empty(M, N, device=DEV)
    print(f"gemm_nt_actgrad M={M} N={N} K={K}")
    for mode in (0, 1):
        lib.recmv_set_gemm_mode(mode)
        call = lambda: L.check(lib.recmv_gemm_nt_actgrad(L.ptr(G), K, L.ptr(Y), K, L.ptr(B), K, L.ptr(out), N, M, N, K,
                                                        ops.ACT_SOFTPLUS, 100.0, 1.0, 0.5, L.stream_ptr(DEV)), "actgrad")
        call()
        ok &= check(out[rows], ref, absprod * 1.5 + 1e-3, f"mode {mode}")   # (+ the hardware exp2 of the derivative)
        t = timeit(call)
        print(f"    mode {mode}: {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s", flush=True)

for K, M, N in ([(153600, 512, 512)] if quick else [(460800, 512, 512), (153600, 512, 512), (6144, 512, 512), (153600, 512, 39),
                                                     (150001, 473, 257)]):
    g = torch.Generator().manual_seed(K + M)
    A = torch.randn(K, M, generator=g).to(DEV)
    B = torch.randn(K, N, generator=g).to(DEV)
    ref = A.double().t() @ B.double()
    absprod = A.abs().double().t() @ B.abs().double()
    print(f"gemm_tn K={K} M={M} N={N}")
    for mode in (0, 1):
        lib.recmv_set_gemm_mode(mode)
        out = ops.gemm_tn(A, B)
        ok &= check(out, ref, absprod, f"mode {mode}")
        ok &= bool(torch.equal(out, ops.gemm_tn(A, B)))
        t = timeit(lambda: ops.gemm_tn(A, B))
        print(f"    mode {mode}: {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s", flush=True)
lib.recmv_set_gemm_mode(0)
print("ALL OK" if ok else "FAILURES")
