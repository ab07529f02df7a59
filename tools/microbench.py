"""Per-kernel micro-benchmarks on the MI355X (HIP-event timed on torch's current stream, which is the
stream every recmv kernel is launched on).  Prints one JSON line per kernel with the algorithmic
bytes/FLOPs of SURVEY.md §8d and the achieved fraction of the roofline.

    python tools/microbench.py [--quick] [--only mc,gemm]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "rec-mv_amd", REPO):
    sys.path.insert(0, str(p))

HBM_PEAK = 8.0e12        # B/s   (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 achievable)
MFMA_F32_PEAK = 157.3e12  # FLOP/s (f32-in MFMA == f32 vector peak)
DEV = "cuda:0"


def timeit(fn, warmup=3, iters=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def report(name, seconds, best, nbytes=None, flops=None, **extra):
    rec = {"kernel": name, "ms": round(seconds * 1e3, 4), "ms_best": round(best * 1e3, 4)}
    if nbytes is not None:
        rec.update(bound="hbm", GBps=round(nbytes / seconds / 1e9, 1), frac=round(nbytes / seconds / HBM_PEAK, 4),
                   alg_bytes=int(nbytes))
    if flops is not None:
        rec.update(bound="mfma", TFLOPs=round(flops / seconds / 1e12, 2), frac=round(flops / seconds / MFMA_F32_PEAK, 4),
                   alg_flops=int(flops))
    rec.update(extra)
    print(json.dumps(rec), flush=True)
    return rec


def bench_inv(quick):
    from recmv import FastMinv
    for n in ([1 << 20] if quick else [10000, 35937, 1 << 20, 1 << 24]):
        ms = torch.randn(n, 3, 3, device=DEV)
        t, b = timeit(lambda: FastMinv.Fast3x3Minv(ms))
        report(f"inv3x3_fwd n={n}", t, b, nbytes=73 * n)
        invs, _ = FastMinv.Fast3x3Minv(ms)
        g = torch.randn_like(ms)
        t, b = timeit(lambda: FastMinv.Fast3x3Minv_backward(g, invs))
        report(f"inv3x3_bwd n={n}", t, b, nbytes=108 * n)


def bench_sampler(quick):
    from recmv import GridSamplerMine
    C, D, H, W = 24, 65, 225, 129        # reference fallback skinning-grid resolution (model/network.py:267)
    vol = torch.softmax(2 * torch.randn(1, C, D, H, W, device=DEV), dim=1)
    vol_cl = vol.contiguous(memory_format=torch.channels_last_3d)
    for P in ([460800] if quick else [6144, 153600, 460800, 1 << 20]):
        grid = (torch.rand(1, 1, 1, P, 3, device=DEV) - 0.5) * 2.2
        touched = min(4 * C * D * H * W, 32 * C * P)
        for nm, v in (("channels_last", vol_cl), ("NCDHW", vol)):
            t, b = timeit(lambda: GridSamplerMine.forward(v, grid, 0, 1))
            report(f"grid_sample_fwd {nm} P={P}", t, b, nbytes=P * (12 + 4 * C) + touched)
        go = torch.randn(1, C, 1, 1, P, device=DEV)
        t, b = timeit(lambda: GridSamplerMine.backward(vol_cl, grid, go, 0, 1, need_grad_input=False))
        report(f"grid_sample_bwd(grid only) P={P}", t, b, nbytes=P * (12 + 4 * C + 12) + touched)
        ggG = torch.randn(1, 1, 1, P, 3, device=DEV)
        t, b = timeit(lambda: GridSamplerMine.dbackward(None, ggG, vol_cl, grid, go, 0, 1, need_grad_input=False))
        report(f"grid_sample_dbwd(ggI none) P={P}", t, b, nbytes=P * (12 + 12 + 4 * C + 12 + 4 * C) + touched)
        # surface-coherent points (what the loop samples: MC vertices in lattice order): neighbours share corner records
        n = int(round(P ** 0.5))
        u, v = torch.meshgrid(torch.linspace(-0.9, 0.9, n, device=DEV), torch.linspace(-0.9, 0.9, n, device=DEV),
                              indexing="ij")
        surf = torch.stack([u, v, 0.3 * torch.sin(3 * u) * torch.cos(2 * v)], -1).view(1, 1, 1, -1, 3).contiguous()
        Pc = surf.shape[3]
        touched_c = min(4 * C * D * H * W, 32 * C * Pc)
        t, b = timeit(lambda: GridSamplerMine.forward(vol_cl, surf, 0, 1))
        report(f"grid_sample_fwd channels_last, surface-coherent P={Pc}", t, b, nbytes=Pc * (12 + 4 * C),
               note="bytes = coordinates + output only (the touched records are shared between neighbours)")
        if P <= 153600:
            t, b = timeit(lambda: GridSamplerMine.backward(vol, grid, go, 0, 1, need_grad_input=True), iters=5)
            report(f"grid_sample_bwd(full, reference behaviour) P={P}", t, b,
                   nbytes=P * (12 + 4 * C + 12) + 2 * 4 * C * D * H * W)


def bench_interp(quick):
    from recmv import interp2x_boundary3d
    for n in ([129] if quick else [33, 65, 129]):
        x = torch.randn(1, 1, n, n, n, device=DEV)
        t, b = timeit(lambda: interp2x_boundary3d.forward(x, 0.0))
        m = 2 * n - 1
        report(f"interp2x_fwd {n}^3->{m}^3", t, b, nbytes=4 * n ** 3 + 5 * m ** 3)
        go = torch.randn(1, 1, m, m, m, device=DEV)
        t, b = timeit(lambda: interp2x_boundary3d.backward(go))
        report(f"interp2x_bwd {m}^3->{n}^3", t, b, nbytes=4 * m ** 3 + 4 * n ** 3)


def body_like_volume(n):
    ax = torch.linspace(-1, 1, n, device=DEV)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    return (torch.sqrt(X ** 2 + (Y * 0.8) ** 2 + Z ** 2) - 0.6 + 0.03 * torch.sin(9 * X) * torch.cos(7 * Z)).contiguous()


def bench_mc(quick):
    from recmv import MCGpu, _lib as L
    import ctypes as C
    for n in ([257] if quick else [129, 257, 385]):
        vol = body_like_volume(n)
        step = 2.0 / (n - 1)
        v, f = MCGpu.mc_gpu(vol, step, step, step, -1.0, -1.0, -1.0, 0.0)
        V, Fc = v.shape[0], f.shape[0]
        t, b = timeit(lambda: MCGpu.mc_gpu(vol, step, step, step, -1.0, -1.0, -1.0, 0.0))
        report(f"mc_gpu total (count+sync+alloc+emit) {n}^3", t, b, nbytes=4 * n ** 3 + 12 * V + 24 * Fc, V=V, F=Fc)
        # split: count phase (classify + scan + D2H) and emit phase
        lib = L.lib()
        ws = torch.empty(int(lib.recmv_mc_workspace_bytes(n, n, n)), dtype=torch.uint8, device=DEV)
        counts = (C.c_int32 * 3)(0, 0, 0)
        st = L.stream_ptr(vol.device)
        t, b = timeit(lambda: lib.recmv_mc_count(L.ptr(vol), n, n, n, 0.0, L.ptr(ws), ws.numel(),
                                                 C.cast(counts, C.c_void_p), st))
        report(f"mc_count (classify+scan+readback) {n}^3", t, b, nbytes=4 * n ** 3)
        verts = torch.empty(V, 3, device=DEV)
        faces = torch.empty(Fc, 3, dtype=torch.int64, device=DEV)
        t, b = timeit(lambda: lib.recmv_mc_emit(L.ptr(vol), n, n, n, 0.0, step, step, step, -1.0, -1.0, -1.0,
                                                L.ptr(ws), ws.numel(), int(counts[2]), L.ptr(verts), V, L.ptr(faces), Fc,
                                                st))
        report(f"mc_emit {n}^3", t, b, nbytes=12 * V + 24 * Fc)
        cdev = torch.empty(3, dtype=torch.int32, device=DEV)
        t, b = timeit(lambda: lib.recmv_mc_run(L.ptr(vol), n, n, n, 0.0, step, step, step, -1.0, -1.0, -1.0, L.ptr(ws),
                                               ws.numel(), L.ptr(verts), V, L.ptr(faces), Fc, L.ptr(cdev), st))
        report(f"mc_run (classify+scan+emit, no host round trip) {n}^3", t, b, nbytes=4 * n ** 3 + 12 * V + 24 * Fc)


def bench_raster(quick):
    """First-hit rasteriser on an MC mesh, 3 frames of 512x512 (the loop's find_surface_ps shape)."""
    from recmv import MCGpu, raster
    from recmv.model import RectifiedPerspectiveCameras
    for n in ([193] if quick else [129, 193, 257]):
        vol = body_like_volume(n)
        step = 2.0 / (n - 1)
        v, f = MCGpu.mc_gpu(vol, step, step, step, -1.0, -1.0, -1.0, 0.0)
        H = W = 512
        cam = RectifiedPerspectiveCameras(torch.tensor([[1000., 1000.]], device=DEV),
                                          torch.tensor([[256., 256.]], device=DEV),
                                          torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3).to(DEV),
                                          torch.tensor([[0., 0., 3.]], device=DEV), image_size=[(W, H)])
        N = 3
        def_vs = (0.55 * v)[None].repeat(N, 1, 1) + 0.02 * torch.arange(N, device=DEV).view(N, 1, 1)
        rast = raster.MeshRasterizer(cam, (H, W))
        frags = rast(def_vs, f)
        covered = int((frags.pix_to_face >= 0).sum())
        ndc = cam.transform_points_ndc(def_vs.reshape(-1, 3)).view(N, -1, 3)
        fv = ndc[:, f.reshape(-1)].reshape(-1, 3, 3).contiguous()
        F = f.shape[0]
        first = torch.arange(N, device=DEV) * F
        num = torch.full((N,), F, device=DEV, dtype=torch.int64)
        t, b = timeit(lambda: raster.rasterize_meshes(fv, first, num, (H, W), max_faces_per_mesh=F))
        # algorithmic bytes: 36 B per face + per pixel 8 B key write/read + 32 B of outputs
        report(f"rasterize_meshes {N}x{H}x{W}, {F} faces/mesh", t, b, nbytes=36 * N * F + N * H * W * (16 + 32),
               faces=N * F, covered_pixels=covered)
        t, b = timeit(lambda: rast(def_vs, f))
        report(f"MeshRasterizer (project + gather + rasterise) {N}x{H}x{W}", t, b, faces=N * F)


def bench_gemm(quick):
    from recmv import ops
    shapes = [(460800, 512, 512), (153600, 512, 512), (6144, 512, 512), (3072, 512, 512), (153600, 512, 39), (153600, 257, 512),
              (153600, 473, 512), (460800, 512, 167)]
    if quick:
        shapes = shapes[:4]
    for M, N, K in shapes:
        A = torch.randn(M, K, device=DEV)
        B = torch.randn(N, K, device=DEV) / K ** 0.5
        bias = torch.randn(N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        t, b = timeit(lambda: ops.gemm_nt(A, B, bias, ops.ACT_SOFTPLUS, 100.0, 1.0, out=out))
        report(f"gemm_nt+bias+softplus M={M} N={N} K={K}", t, b, flops=2.0 * M * N * K)
        t, b = timeit(lambda: ops.gemm_nt(A, B, None, ops.ACT_NONE, 0.0, 1.0, out=out))
        report(f"gemm_nt (no bias/act) M={M} N={N} K={K}", t, b, flops=2.0 * M * N * K)
        t, b = timeit(lambda: torch.nn.functional.softplus(torch.addmm(bias, A, B.t()), beta=100))
        report(f"  (torch addmm+softplus / rocBLAS, for scale) M={M} N={N} K={K}", t, b, flops=2.0 * M * N * K)
    for K, M, N in ([(153600, 512, 512)] if quick else [(460800, 512, 512), (153600, 512, 512), (6144, 512, 512),
                                                         (153600, 512, 39)]):
        A = torch.randn(K, M, device=DEV)
        B = torch.randn(K, N, device=DEV)
        t, b = timeit(lambda: ops.gemm_tn(A, B))
        report(f"gemm_tn (dW) K={K} M={M} N={N}", t, b, flops=2.0 * M * N * K)


def bench_sdf(quick):
    from recmv.model import getTmpSdf
    sdf = getTmpSdf(DEV, 6)
    flop_pt = 2 * 1966592
    for P in ([1 << 20] if quick else [35937, 153600, 1 << 20, 1 << 22]):
        x = torch.randn(P, 3, device=DEV) * 0.5
        with torch.no_grad():
            t, b = timeit(lambda: sdf(x, 1.0), iters=10)
        report(f"sdf_mlp forward (no grad, 9 fused layers + PE) P={P}", t, b, flops=flop_pt * P,
               Mpts_per_s=round(P / t / 1e6, 2))
    P = 153600
    x = torch.randn(P, 3, device=DEV) * 0.5

    def train_step():
        xs = x.clone().requires_grad_(True)
        y = sdf(xs, 1.0)
        y.abs().mean().backward()

    t, b = timeit(train_step, iters=5)
    report(f"sdf_mlp fwd+bwd(theta,x) autograd path P={P}", t, b, flops=3 * flop_pt * P)


ALL = {"inv": bench_inv, "sampler": bench_sampler, "interp": bench_interp, "mc": bench_mc, "raster": bench_raster, "gemm": bench_gemm,
       "sdf": bench_sdf}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    names = [n for n in a.only.split(",") if n] or list(ALL)
    print(json.dumps({"device": torch.cuda.get_device_name(0), "torch": torch.__version__}))
    for n in names:
        ALL[n](a.quick)
