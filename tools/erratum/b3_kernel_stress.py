"""Is a product kernel of the bf16x6 matrix mode bit-reproducible from launch to launch while two other streams keep the device busy
with this library's own kernels?  The shapes are the jets' of the loop (rows = 4 x / 3 x the regulariser's points); the last-layer
tangent product (N = 3 outputs) is where tools/loop_repro_inproc.py sees the offset MLP's Jacobian part.

    python tools/b3_kernel_stress.py [repeats=200]
"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402
from recmv import _lib as L, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = L.lib()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def rnd(*s):
    return torch.randn(*s, generator=g).to(dev)


P = 30720
shapes = [("last-layer tangents  M=3P N=3   K=512", 3 * P, 3, 512, False),
          ("hidden layer         M=4P N=512 K=512", 4 * P, 512, 512, True),
          ("layer 0 (padded)     M=4P N=512 K=168", 4 * P, 512, 168, True),
          ("value rows, last     M=P  N=3   K=512", P, 3, 512, False),
          ("cotangent, last      M=3P N=512 K=3  ", 3 * P, 512, 3, False)]
side = [torch.cuda.Stream(), torch.cuda.Stream()]
noiseA, noiseB = rnd(60000, 512), rnd(512, 512) / 22.0
noiseT = rnd(60000, 512)
for mode in (1, 0):
    lib.recmv_set_gemm_mode(mode)
    for name, M, N, K, bias in shapes:
        A, B = rnd(M, K), rnd(N, K) / K ** 0.5
        b = rnd(N) if bias else None
        out = torch.empty(M, N, device=dev)
        ref = ops.gemm_nt(A, B, b, ops.ACT_RELU if bias else ops.ACT_NONE, 0.0).clone()
        torch.cuda.synchronize()
        for busy in (False, True):
            bad, worst = 0, 0.0
            for r in range(reps):
                if busy:
                    for i, st in enumerate(side):
                        with torch.cuda.stream(st):
                            for _ in range(2):
                                if (i + r) % 2:
                                    ops.gemm_nt(noiseA, noiseB, None, ops.ACT_RELU, 0.0)
                                else:
                                    ops.gemm_tn(noiseT, noiseA)
                ops.gemm_nt(A, B, b, ops.ACT_RELU if bias else ops.ACT_NONE, 0.0, out=out)
                ne = out != ref
                n = int(ne.sum())              # (a host read-back per repetition: the side streams run ahead meanwhile)
                if n:
                    bad += 1
                    worst = max(worst, float((out - ref).abs().max()))
            print("%s  %-40s %-28s %d of %d launches differ from the first%s" % (
                "bf16x6" if mode else "f32   ", name, "two busy side streams" if busy else "alone", bad, reps,
                "  (max |d| %.3e, max |ref| %.3e)" % (worst, float(ref.abs().max())) if bad else ""), flush=True)
lib.recmv_set_gemm_mode(0)
