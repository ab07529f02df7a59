"""A/B of the large f32 products: RECMV_GEMM_OCC=1 (default: the high-occupancy kernels, four / five workgroups per CU) against =0
(the two-per-CU kernels they replaced), one process per setting (the switch is read once): HIP events over `reps` launches, and for
the NT products an order-independent checksum of the raw output bits (the kernels add the same products in the same order).
The round-3 exploration that chose the tiles ran more variants through a temporary switch: profiles/r03_gemm_occupancy_variants.txt.
python tools/gemm_variants.py [nt|tn]"""
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
SHAPES = [(460800, 512, 512), (153600, 512, 512), (90000, 512, 512), (90000, 512, 40), (50000, 256, 512)]
PEAK = 157.3


def child():
    import torch
    sys.path.insert(0, str(REPO / "rec-mv_amd"))
    from recmv import _lib as L
    from recmv import ops
    dev = torch.device("cuda", 0)
    v = os.environ.get("RECMV_GEMM_OCC", "1")
    for (M, N, K) in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g).to(dev)
        B = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        Y = torch.rand(M, N, generator=g).to(dev) * 0.05
        out = torch.empty(M, N, device=dev)
        for kind in ("softplus", "mulgrad", "actgrad"):
            st = L.stream_ptr(dev)
            if kind == "softplus":
                fn = lambda: ops.gemm_nt(A, B, bias, ops.ACT_SOFTPLUS, 100.0, 1.0, out=out)   # noqa: E731
            elif kind == "mulgrad":        # C = (A . B^T) (.) act'(Y): the epilogue transform of the backward chain
                fn = lambda: L.check(L.lib().recmv_gemm_nt_mulgrad(L.ptr(A), K, L.ptr(B), K, L.ptr(out), N, M, N, K, L.ptr(Y), N,   # noqa: E731
                                                                   ops.ACT_SOFTPLUS, 100.0, 1.0, 1.0, st), "mulgrad")
            else:                          # C = (G (.) act'(Y)) . B^T: the operand transform (K == N only)
                if K != N:
                    continue
                fn = lambda: L.check(L.lib().recmv_gemm_nt_actgrad(L.ptr(A), K, L.ptr(Y), N, L.ptr(B), K, L.ptr(out), N, M, N, K,   # noqa: E731
                                                                   ops.ACT_SOFTPLUS, 100.0, 1.0, 1.0, st), "actgrad")
            try:
                fn()
            except Exception as e:                                   # an op this build does not expose under that name
                print(f"occ={v} {kind} M={M} N={N} K={K}: skipped ({type(e).__name__}: {e})")
                continue
            torch.cuda.synchronize()
            reps = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps * 1e3)
            check = int(out.view(torch.int32).to(torch.int64).sum().item())
            tf = 2.0 * M * N * K / best / 1e6
            print(f"occ={v} {kind:8s} M={M} N={N} K={K}: {best:8.1f} us  {tf:6.1f} TFLOP/s  {tf / PEAK:.3f} of peak  bits {check}")


TN_SHAPES = [(460800, 512, 512), (153600, 512, 512), (90000, 512, 512), (60000, 512, 512), (90000, 256, 512), (90000, 512, 40),
             (24000, 512, 512), (6000, 512, 512)]


def child_tn():
    """dW = X^T dZ: K = points, split over the workgroups; timed with its reduction pass (ops.gemm_tn launches both)."""
    import torch
    sys.path.insert(0, str(REPO / "rec-mv_amd"))
    from recmv import ops
    dev = torch.device("cuda", 0)
    v = os.environ.get("RECMV_GEMM_OCC", "1")
    for (K, M, N) in TN_SHAPES:
        g = torch.Generator(device="cpu").manual_seed(M + N + K)
        A = torch.randn(K, M, generator=g).to(dev)
        B = torch.randn(K, N, generator=g).to(dev)
        out = ops.gemm_tn(A, B)
        ref = (A.double().t() @ B.double())
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        torch.cuda.synchronize()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                ops.gemm_tn(A, B)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        tf = 2.0 * M * N * K / best / 1e6
        print(f"tn occ={v} K={K} M={M} N={N}: {best:8.1f} us  {tf:6.1f} TFLOP/s  {tf / PEAK:.3f} of peak  max err / max |ref| {err:.2e}")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "nt"
    if what == "child":
        child()
    elif what == "child_tn":
        child_tn()
    else:
        for v in ("0", "1", "0", "1"):
            subprocess.run([sys.executable, __file__, "child_tn" if what == "tn" else "child"], env=dict(os.environ, RECMV_GEMM_OCC=v),
                           check=False)
