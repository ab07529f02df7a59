"""Is recmv_def_regu bit-reproducible from launch to launch while two other streams run large products?  Which products disturb it,
and what do the differences look like?

    python tools/def_regu_stress.py [repeats=400]
    RECMV_GEMM_OCC=0 python tools/def_regu_stress.py      # the f32 mode's products on the 128 x 128 kernel with 72 KB of LDS

Round 5: with the bf16x6 mode's 128 x 128 kernel (gemm_nt_b3_kernel, 66 KB of dynamic LDS per workgroup) on the side streams, a few
launches in a hundred of this per-thread, atomics-free kernel give different bits for the SAME input (profiles/r05_def_regu_stress*.txt).
"""
import os
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
big_A, big_B = torch.randn(120000, 512, generator=g).to(dev), (torch.randn(512, 512, generator=g) / 22.0).to(dev)
mid_A = torch.randn(12000, 512, generator=g).to(dev)
tn_A, tn_B = torch.randn(120000, 512, generator=g).to(dev), torch.randn(120000, 512, generator=g).to(dev)
side = [torch.cuda.Stream(), torch.cuda.Stream()]
print("# RECMV_GEMM_OCC=%s" % os.environ.get("RECMV_GEMM_OCC", "(unset)"))


def run(label, mode, fam, work):
    lib.recmv_set_gemm_mode(mode)
    lib.recmv_set_b3_families(fam)
    torch.cuda.synchronize()               # (the reference launch runs alone)
    y0, g0 = torch.empty(P, device=dev), torch.empty_like(J)
    L.check(lib.recmv_def_regu(L.ptr(J), P, 0.03, L.ptr(y0), L.ptr(g0), L.stream_ptr(dev)), "def_regu")
    torch.cuda.synchronize()
    bad, shown = 0, 0
    for r in range(reps):
        if work is not None:
            for st in side:
                with torch.cuda.stream(st):
                    work()
        y, gj = torch.empty(P, device=dev), torch.empty_like(J)
        L.check(lib.recmv_def_regu(L.ptr(J), P, 0.03, L.ptr(y), L.ptr(gj), L.stream_ptr(dev)), "def_regu")
        same = torch.equal(y, y0) and torch.equal(gj, g0)
        if not same:
            bad += 1
            if shown < 3:
                shown += 1
                torch.cuda.synchronize()
                ny = (y != y0).nonzero().view(-1)
                ng = (gj != g0).view(P, 9).any(1).nonzero().view(-1)
                idx = torch.unique(torch.cat([ny, ng]))
                blocks = torch.unique(idx // 256)
                lanes = torch.unique(idx % 64)
                dy = (y - y0).abs().max().item()
                dg = (gj - g0).abs().max().item()
                print("    launch %d: %d matrices differ (y %d, dy/dJ %d) in %d workgroup(s) %s, lanes %s; max |dy| %.3e (max |y| %.3e), "
                      "max |d dy/dJ| %.3e (max %.3e); first matrices %s" % (
                          r, idx.numel(), ny.numel(), ng.numel(), blocks.numel(), blocks[:6].tolist(), lanes[:8].tolist(), dy,
                          y0.abs().max().item(), dg, g0.abs().max().item(), idx[:6].tolist()), flush=True)
    print("def_regu beside %-64s %d of %d launches differ from the first" % (label + ":", bad, reps), flush=True)


if len(sys.argv) > 2 and sys.argv[2] == "ab":
    # the A/B of the f32 -> bf16 conversion inside the product kernels: run once with the product library and once with
    # RECMV_LIB_PATH=tools/bin/librecmv_hip_intsplit.so (the same sources built with -DRECMV_SPLIT_INT)
    A4k, B4k = torch.randn(12000, 4096, generator=g).to(dev), (torch.randn(512, 4096, generator=g) / 64.0).to(dev)
    print("# library: %s" % L.LIB_PATH)
    torch.cuda.synchronize()
    run("bf16x6 64x64 products, K = 4096", 1, 7, lambda: ops.gemm_nt(A4k, B4k, None, ops.ACT_RELU, 0.0))
    torch.cuda.synchronize()
    run("bf16x6 64x64 products, K = 512", 1, 7, lambda: ops.gemm_nt(mid_A, big_B, None, ops.ACT_RELU, 0.0))
    torch.cuda.synchronize()
    run("bf16x6 128x128 products (gemm_nt_b3_kernel)", 1, 7, lambda: ops.gemm_nt(big_A, big_B, None, ops.ACT_RELU, 0.0))
    torch.cuda.synchronize()
    lib.recmv_set_gemm_mode(0)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "shapes":
    # which part of the 64 x 64 product kernel matters: the K loop's length, the epilogue's activation, the loader
    B32, A32 = (torch.randn(512, 32, generator=g) / 6.0).to(dev), torch.randn(12000, 32, generator=g).to(dev)
    A4k, B4k = torch.randn(12000, 4096, generator=g).to(dev), (torch.randn(512, 4096, generator=g) / 64.0).to(dev)
    A510, B510 = torch.randn(12000, 510, generator=g).to(dev), (torch.randn(512, 510, generator=g) / 22.0).to(dev)
    nar_A = torch.randn(3000, 512, generator=g).to(dev)
    for mode in (1, 0):
        tag = "bf16x6" if mode else "f32"
        run(tag + " 64x64 K=512 no activation", mode, 7, lambda: ops.gemm_nt(mid_A, big_B, None, ops.ACT_NONE, 0.0))
        run(tag + " 64x64 K=512 softplus(100)", mode, 7, lambda: ops.gemm_nt(mid_A, big_B, None, ops.ACT_SOFTPLUS, 100.0))
        run(tag + " 64x64 K=32 (one K-tile) relu", mode, 7, lambda: ops.gemm_nt(A32, B32, None, ops.ACT_RELU, 0.0))
        run(tag + " 64x64 K=4096 relu", mode, 7, lambda: ops.gemm_nt(A4k, B4k, None, ops.ACT_RELU, 0.0))
        run(tag + " 64x64 K=510 (element-guarded loader) relu", mode, 7, lambda: ops.gemm_nt(A510, B510, None, ops.ACT_RELU, 0.0))
        run(tag + " 64x32 narrow tiles, 3000 rows, relu", mode, 7, lambda: ops.gemm_nt(nar_A, big_B, None, ops.ACT_RELU, 0.0))
    lib.recmv_set_gemm_mode(0)
    sys.exit(0)
run("idle side streams", 1, 7, None)
run("bf16x6 128x128 products (gemm_nt_b3_kernel, 66 KB LDS)", 1, 7, lambda: ops.gemm_nt(big_A, big_B, None, ops.ACT_RELU, 0.0))
run("bf16x6 64x64 products (gemm_nt_kernel<1,..,BF3>, 36 KB LDS)", 1, 7, lambda: ops.gemm_nt(mid_A, big_B, None, ops.ACT_RELU, 0.0))
run("bf16x6 TN products (gemm_tn_kernel<BF3>, 66 KB LDS)", 1, 7, lambda: ops.gemm_tn(tn_A, tn_B))
run("f32 128x128 products", 0, 7, lambda: ops.gemm_nt(big_A, big_B, None, ops.ACT_RELU, 0.0))
run("f32 TN products", 0, 7, lambda: ops.gemm_tn(tn_A, tn_B))
lib.recmv_set_gemm_mode(0)
