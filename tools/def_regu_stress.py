"""Is recmv_def_regu bit-reproducible from launch to launch while two other streams run the bf16x6 mode's large products?
    python tools/def_regu_stress.py [repeats=400]"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402
from recmv import _lib as L, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
lib = L.lib()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
P = 30714
J = (torch.eye(3).view(1, 3, 3) + 0.02 * torch.randn(P, 3, 3, generator=g)).to(dev).contiguous()
noiseA, noiseB = torch.randn(120000, 512, generator=g).to(dev), (torch.randn(512, 512, generator=g) / 22.0).to(dev)
side = [torch.cuda.Stream(), torch.cuda.Stream()]
for mode in (1, 0):
    lib.recmv_set_gemm_mode(mode)
    y0, g0 = torch.empty(P, device=dev), torch.empty_like(J)
    L.check(lib.recmv_def_regu(L.ptr(J), P, 0.03, L.ptr(y0), L.ptr(g0), L.stream_ptr(dev)), "def_regu")
    torch.cuda.synchronize()
    for busy in (False, True):
        bad = 0
        for r in range(reps):
            if busy:
                for st in side:
                    with torch.cuda.stream(st):
                        ops.gemm_nt(noiseA, noiseB, None, ops.ACT_RELU, 0.0)
            y, gj = torch.empty(P, device=dev), torch.empty_like(J)
            L.check(lib.recmv_def_regu(L.ptr(J), P, 0.03, L.ptr(y), L.ptr(gj), L.stream_ptr(dev)), "def_regu")
            bad += int(not (torch.equal(y, y0) and torch.equal(gj, g0)))
        print("def_regu, %s products on two side streams %s: %d of %d launches differ from the first" % (
            "bf16x6" if mode else "f32", "busy" if busy else "idle", bad, reps), flush=True)
lib.recmv_set_gemm_mode(0)
