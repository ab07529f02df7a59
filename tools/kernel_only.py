"""Kernel-only durations of the HBM-bound kernels at named shapes, from a rocprofv3 kernel trace.

The event-based micro-benchmark (tools/microbench.py) cannot see below ~20 us: the GPU waits for the python wrapper
between the two events.  Here every case is a run of identical launches bracketed by a separator kernel, the trace is
read back from the rocpd database in start order, and each segment is reported per kernel name.

    # on the GPU box
    cd /tmp && rocprofv3 --kernel-trace -d /tmp/ko -o run -- python $REPO/tools/kernel_only.py run /tmp/ko_cases.json
    python tools/kernel_only.py report /tmp/ko /tmp/ko_cases.json > profiles/<name>.txt
"""
import glob
import json
import sqlite3
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO), str(REPO / "tools")]
REPS = 10
HBM_PEAK = 8.0e12


def run(out_json):
    import torch
    from recmv import FastMinv, GridSamplerMine, MCGpu, interp2x_boundary3d, raster
    dev = "cuda:0"
    cases = []
    sep = torch.zeros(257, device=dev)

    def case(name, nbytes, fn):
        torch.sort(sep)                          # (the warm-up call gets a segment of its own, skipped by the report)
        fn()
        torch.cuda.synchronize()
        torch.sort(sep)                          # separator launch: a (rocprim) sort kernel nothing else here uses
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        cases.append({"name": name, "alg_bytes": int(nbytes)})

    # ---- sampler on the skinning grid (channels-last), random and surface-coherent points
    C, D, H, W = 24, 65, 225, 129
    vol = torch.softmax(2 * torch.randn(1, C, D, H, W, device=dev), dim=1).contiguous(memory_format=torch.channels_last_3d)
    for P in (153600, 460800, 1 << 20, 1 << 22):
        n = int(round(P ** 0.5))
        u, v = torch.meshgrid(torch.linspace(-0.9, 0.9, n, device=dev), torch.linspace(-0.9, 0.9, n, device=dev),
                              indexing="ij")
        surf = torch.stack([u, v, 0.3 * torch.sin(3 * u) * torch.cos(2 * v)], -1).view(1, 1, 1, -1, 3).contiguous()
        Pc = surf.shape[3]
        go = torch.randn(1, C, 1, 1, Pc, device=dev)
        gg = torch.randn(1, 1, 1, Pc, 3, device=dev)
        case(f"sampler fwd, surface-coherent P={Pc} (coords + output bytes)", Pc * (12 + 4 * C),
             lambda: GridSamplerMine.forward(vol, surf, 0, 1))
        case(f"sampler bwd (grid only), surface-coherent P={Pc}", Pc * (12 + 4 * C + 12),
             lambda: GridSamplerMine.backward(vol, surf, go, 0, 1, need_grad_input=False))
        case(f"sampler dbwd, surface-coherent P={Pc}", Pc * (12 + 12 + 4 * C + 12 + 4 * C),
             lambda: GridSamplerMine.dbackward(None, gg, vol, surf, go, 0, 1, need_grad_input=False))
        if P <= (1 << 20):
            grid = (torch.rand(1, 1, 1, P, 3, device=dev) - 0.5) * 2.2
            case(f"sampler fwd, uniformly random P={P} (+ touched records)", P * (12 + 4 * C) + min(4 * C * D * H * W, 32 * C * P),
                 lambda: GridSamplerMine.forward(vol, grid, 0, 1))
    # ---- 3x3 inverse
    for n in (1 << 20, 1 << 24):
        ms = torch.randn(n, 3, 3, device=dev)
        case(f"inv3x3 fwd n={n}", 73 * n, lambda: FastMinv.Fast3x3Minv(ms))
    # ---- 2x boundary upsampler
    for n in (65, 129):
        x = torch.randn(1, 1, n, n, n, device=dev)
        case(f"interp2x fwd {n}^3 -> {2 * n - 1}^3", 4 * n ** 3 + 5 * (2 * n - 1) ** 3,
             lambda: interp2x_boundary3d.forward(x, 0.0))
    # ---- marching cubes (count + emit: classify / scan / emit kernels reported separately)
    from microbench import body_like_volume
    for n in (257, 385):
        torch.sort(sep)                          # keep the set-up launches out of the previous case's segment
        vol3 = body_like_volume(n)
        step = 2.0 / (n - 1)
        v, f = MCGpu.mc_gpu(vol3, step, step, step, -1.0, -1.0, -1.0, 0.0)
        case(f"mc_gpu {n}^3 (V={v.shape[0]}, F={f.shape[0]}): volume + vertices + faces", 4 * n ** 3 + 12 * v.shape[0] + 24 * f.shape[0],
             lambda: MCGpu.mc_gpu(vol3, step, step, step, -1.0, -1.0, -1.0, 0.0))
    # ---- first-hit mesh rasteriser
    torch.sort(sep)
    vol3 = body_like_volume(193)
    step = 2.0 / 192
    v, f = MCGpu.mc_gpu(vol3, step, step, step, -1.0, -1.0, -1.0, 0.0)
    from recmv.model import RectifiedPerspectiveCameras
    cam = RectifiedPerspectiveCameras(torch.tensor([[1000., 1000.]], device=dev), torch.tensor([[256., 256.]], device=dev),
                                      torch.diag(torch.tensor([-1., -1., 1.])).view(1, 3, 3).to(dev),
                                      torch.tensor([[0., 0., 3.]], device=dev), image_size=[(512, 512)])
    N = 3
    dv = (0.55 * v)[None].repeat(N, 1, 1) + 0.02 * torch.arange(N, device=dev).view(N, 1, 1)
    ndc = cam.transform_points_ndc(dv.reshape(-1, 3)).view(N, -1, 3)
    fv = ndc[:, f.reshape(-1)].reshape(-1, 3, 3).contiguous()
    F = f.shape[0]
    first = torch.arange(N, device=dev) * F
    num = torch.full((N,), F, device=dev, dtype=torch.int64)
    case(f"rasterize_meshes 3x512x512, {F} faces/mesh", 36 * N * F + N * 512 * 512 * 48,
         lambda: raster.rasterize_meshes(fv, first, num, (512, 512), max_faces_per_mesh=F))
    pts = ndc.reshape(-1, 3).contiguous()
    V = v.shape[0]
    pfirst = torch.arange(N, device=dev) * V
    pnum = torch.full((N,), V, device=dev, dtype=torch.int64)
    case(f"rasterize_points 3x512x512, {V} points/cloud, K=50, r=0.006", 12 * N * V + N * 512 * 512 * 50 * 12,
         lambda: raster.rasterize_points(pts, pfirst, pnum, (512, 512), 0.006, 50, max_points_per_cloud=V))
    json.dump(cases, open(out_json, "w"))


def run_sampler(out_json):
    """Sampler only: forward, backward, double backward on surface-coherent points, the exact-order kernels against the
    record-coalesced lane splits (recmv_set_sampler_mode 1 / 0 / 2 / 3 = one lane per point, 8 lanes x 3 channels, 4 x 6, 2 x 12)."""
    import torch
    from recmv import GridSamplerMine, _lib
    dev = "cuda:0"
    cases = []
    sep = torch.zeros(257, device=dev)

    def case(name, nbytes, fn):
        torch.sort(sep)
        fn()
        torch.cuda.synchronize()
        torch.sort(sep)
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        cases.append({"name": name, "alg_bytes": int(nbytes)})

    C, D, H, W = 24, 65, 225, 129
    vol = torch.softmax(2 * torch.randn(1, C, D, H, W, device=dev), dim=1).contiguous(memory_format=torch.channels_last_3d)
    for P in (105038, 153600, 460800, 1 << 20, 1 << 22):
        n = int(round(P ** 0.5))
        u, v = torch.meshgrid(torch.linspace(-0.9, 0.9, n, device=dev), torch.linspace(-0.9, 0.9, n, device=dev), indexing="ij")
        surf = torch.stack([u, v, 0.3 * torch.sin(3 * u) * torch.cos(2 * v)], -1).view(1, 1, 1, -1, 3).contiguous()
        Pc = surf.shape[3]
        go = torch.randn(1, C, 1, 1, Pc, device=dev)
        gg = torch.randn(1, 1, 1, Pc, 3, device=dev)
        for mode, tag in ((1, "exact order, 1 lane/point"), (2, "8 lanes x 3 ch"), (3, "4 lanes x 6 ch"), (4, "2 lanes x 12 ch"),
                          (0, "default")):
            _lib.lib().recmv_set_sampler_mode(mode)
            case(f"sampler fwd [{tag}], surface-coherent P={Pc} (coords + output bytes)", Pc * (12 + 4 * C),
                 lambda: GridSamplerMine.forward(vol, surf, 0, 1))
            case(f"sampler bwd (grid only) [{tag}], surface-coherent P={Pc}", Pc * (12 + 4 * C + 12),
                 lambda: GridSamplerMine.backward(vol, surf, go, 0, 1, need_grad_input=False))
            case(f"sampler dbwd [{tag}], surface-coherent P={Pc}", Pc * (12 + 12 + 4 * C + 12 + 4 * C),
                 lambda: GridSamplerMine.dbackward(None, gg, vol, surf, go, 0, 1, need_grad_input=False))
        if P <= (1 << 20):
            rnd = (torch.rand(1, 1, 1, P, 3, device=dev) - 0.5) * 2.2
            gor = torch.randn(1, C, 1, 1, P, device=dev)
            case(f"sampler fwd [default], uniformly random P={P} (+ touched records)", P * (12 + 4 * C) + min(4 * C * D * H * W, 32 * C * P),
                 lambda: GridSamplerMine.forward(vol, rnd, 0, 1))
            case(f"sampler bwd [default], uniformly random P={P} (+ touched records)", P * (24 + 4 * C) + min(4 * C * D * H * W, 32 * C * P),
                 lambda: GridSamplerMine.backward(vol, rnd, gor, 0, 1, need_grad_input=False))
        _lib.lib().recmv_set_sampler_mode(0)
    json.dump(cases, open(out_json, "w"))


def run_mc(out_json):
    """Marching cubes only: 257^3, 385^3 and the loop's coarse pyramid, through recmv_mc_run (no host round trip)."""
    import ctypes as C
    import torch
    from recmv import _lib as L
    from microbench import body_like_volume
    dev = "cuda:0"
    cases = []
    sep = torch.zeros(257, device=dev)
    lib = L.lib()
    for n in (257, 385):
        torch.sort(sep)
        vol3 = body_like_volume(n)
        ws = torch.empty(int(lib.recmv_mc_workspace_bytes(n, n, n)), dtype=torch.uint8, device=dev)
        cnt = (C.c_int32 * 3)(0, 0, 0)
        L.check(lib.recmv_mc_count(L.ptr(vol3), n, n, n, 0.0, L.ptr(ws), ws.numel(), C.cast(cnt, C.c_void_p), L.stream_ptr(vol3.device)), "mc")
        V, F = int(cnt[0]), int(cnt[1])
        vb = torch.empty(V, 3, device=dev)
        fb = torch.empty(F, 3, dtype=torch.int64, device=dev)
        cdev = torch.empty(3, dtype=torch.int32, device=dev)
        step = 2.0 / (n - 1)

        def fn():
            L.check(lib.recmv_mc_run(L.ptr(vol3), n, n, n, 0.0, step, step, step, -1.0, -1.0, -1.0, L.ptr(ws), ws.numel(), L.ptr(vb), V,
                                     L.ptr(fb), F, L.ptr(cdev), L.stream_ptr(vol3.device)), "mc_run")
        torch.sort(sep)
        fn()
        torch.cuda.synchronize()
        torch.sort(sep)
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        cases.append({"name": f"mc_run {n}^3 (V={V}, F={F}): volume + vertices + faces", "alg_bytes": 4 * n ** 3 + 12 * V + 24 * F})
    json.dump(cases, open(out_json, "w"))


def run_interp(out_json):
    """The 2x boundary upsampler only."""
    import torch
    from recmv import interp2x_boundary3d
    dev = "cuda:0"
    cases = []
    sep = torch.zeros(257, device=dev)
    for n in (65, 129, 193):
        x = torch.randn(1, 1, n, n, n, device=dev)
        torch.sort(sep)
        interp2x_boundary3d.forward(x, 0.0)
        torch.cuda.synchronize()
        torch.sort(sep)
        for _ in range(REPS):
            interp2x_boundary3d.forward(x, 0.0)
        torch.cuda.synchronize()
        cases.append({"name": f"interp2x fwd {n}^3 -> {2 * n - 1}^3", "alg_bytes": 4 * n ** 3 + 5 * (2 * n - 1) ** 3})
    json.dump(cases, open(out_json, "w"))


def run_pmc(out_json):
    """The HBM-bound kernels at the sizes the bench line quotes (default kernels only), for the counter passes of
    tools/pmc_hbm_kernels.sh: sampler forward / backward / double backward at 85 k, 256 k, 2^20 and 2^22 surface-coherent points,
    marching cubes 257^3 and the loop's coarse pyramid, the 2x upsampler 129^3 -> 257^3, the 3x3 inverse at 2^20."""
    import ctypes as C
    import torch
    from recmv import FastMinv, GridSamplerMine, _lib as L, interp2x_boundary3d
    from microbench import body_like_volume
    dev = "cuda:0"
    cases = []
    sep = torch.zeros(257, device=dev)

    def case(name, nbytes, fn):
        torch.sort(sep)
        fn()
        torch.cuda.synchronize()
        torch.sort(sep)
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        cases.append({"name": name, "alg_bytes": int(nbytes)})

    Cc, D, H, W = 24, 65, 225, 129
    vol = torch.softmax(2 * torch.randn(1, Cc, D, H, W, device=dev), dim=1).contiguous(memory_format=torch.channels_last_3d)
    for P in (85000, 256000, 1 << 20, 1 << 22):
        n = int(round(P ** 0.5))
        u, v = torch.meshgrid(torch.linspace(-0.9, 0.9, n, device=dev), torch.linspace(-0.9, 0.9, n, device=dev), indexing="ij")
        surf = torch.stack([u, v, 0.3 * torch.sin(3 * u) * torch.cos(2 * v)], -1).view(1, 1, 1, -1, 3).contiguous()
        Pc = surf.shape[3]
        go = torch.randn(1, Cc, 1, 1, Pc, device=dev)
        gg = torch.randn(1, 1, 1, Pc, 3, device=dev)
        case(f"sampler fwd, surface-coherent P={Pc} (coords + output bytes)", Pc * (12 + 4 * Cc),
             lambda: GridSamplerMine.forward(vol, surf, 0, 1))
        case(f"sampler bwd (grid only), surface-coherent P={Pc}", Pc * (12 + 4 * Cc + 12),
             lambda: GridSamplerMine.backward(vol, surf, go, 0, 1, need_grad_input=False))
        case(f"sampler dbwd, surface-coherent P={Pc}", Pc * (12 + 12 + 4 * Cc + 12 + 4 * Cc),
             lambda: GridSamplerMine.dbackward(None, gg, vol, surf, go, 0, 1, need_grad_input=False))
    lib = L.lib()
    for shape in ((257, 257, 257), (225, 321, 129)):
        nx, ny, nz = shape
        torch.sort(sep)
        if nx == ny == nz:
            vol3 = body_like_volume(nx)
        else:
            ax = [torch.linspace(-1, 1, k, device=dev) for k in shape]
            X, Y, Z = torch.meshgrid(*ax, indexing="ij")
            vol3 = (torch.sqrt(X * X + (0.8 * Y) ** 2 + Z * Z) - 0.6 + 0.03 * torch.sin(9 * X) * torch.cos(7 * Z)).contiguous()
        ws = torch.empty(int(lib.recmv_mc_workspace_bytes(nx, ny, nz)), dtype=torch.uint8, device=dev)
        cnt = (C.c_int32 * 3)(0, 0, 0)
        L.check(lib.recmv_mc_count(L.ptr(vol3), nx, ny, nz, 0.0, L.ptr(ws), ws.numel(), C.cast(cnt, C.c_void_p), L.stream_ptr(vol3.device)), "mc")
        V, F = int(cnt[0]), int(cnt[1])
        vb = torch.empty(V, 3, device=dev)
        fb = torch.empty(F, 3, dtype=torch.int64, device=dev)
        cdev = torch.empty(3, dtype=torch.int32, device=dev)
        case(f"mc_run {nx}x{ny}x{nz} (V={V}, F={F}): volume + vertices + faces", 4 * nx * ny * nz + 12 * V + 24 * F,
             lambda: L.check(lib.recmv_mc_run(L.ptr(vol3), nx, ny, nz, 0.0, 2. / nx, 2. / ny, 2. / nz, -1.0, -1.0, -1.0, L.ptr(ws), ws.numel(),
                                              L.ptr(vb), V, L.ptr(fb), F, L.ptr(cdev), L.stream_ptr(vol3.device)), "mc_run"))
    x = torch.randn(1, 1, 129, 129, 129, device=dev)
    case("interp2x fwd 129^3 -> 257^3", 4 * 129 ** 3 + 5 * 257 ** 3, lambda: interp2x_boundary3d.forward(x, 0.0))
    ms = torch.randn(1 << 20, 3, 3, device=dev)
    case("inv3x3 fwd n=1048576", 73 * (1 << 20), lambda: FastMinv.Fast3x3Minv(ms))
    json.dump(cases, open(out_json, "w"))


def report_pmc(cases_json, *prof_dirs):
    """Counter passes (one directory per `rocprofv3 --pmc` run over `run_pmc`) -> per case and kernel: HBM-side bytes per launch
    (FETCH_SIZE x 2 per MI355X_MICROARCH.md's gfx950 correction for wide reads — an UPPER bound for narrow gathers —, WRITE_SIZE as
    reported; both in KiB units x 1024), L2 hit rate, and traffic / algorithmic bytes."""
    import csv
    cases = json.load(open(cases_json))
    per_case = [dict() for _ in cases]
    for d in prof_dirs:
        rows = []
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            with open(f, newline="") as fh:
                rows += list(csv.DictReader(fh))
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        body, ci = None, 0
        last_id = None
        seq = []                                 # (dispatch id, kernel, {counter: value}) in dispatch order
        for r in rows:
            did = int(r["Dispatch_Id"])
            if did != last_id:
                seq.append((r["Kernel_Name"], {}))
                last_id = did
            seq[-1][1][r["Counter_Name"]] = seq[-1][1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for name, ctr in seq + [("sort (sentinel)", {})]:
            if "recmv::" not in name and "sort" in name.lower():
                if body and max(len(v) for v in body.values()) >= REPS and ci < len(cases):
                    for k, lst in body.items():
                        tgt = per_case[ci].setdefault(k, {})
                        for c in lst[0]:
                            tgt[c] = sum(x[c] for x in lst) / len(lst)
                        tgt["launches_per_call"] = max(len(lst) // REPS, 1)
                    ci += 1
                body = {}
            elif body is not None and "recmv::" in name:
                body.setdefault(name.split("recmv::(anonymous namespace)::")[-1].split("(")[0], []).append(ctr)
        if ci != len(cases):
            print("# WARNING: %s matched %d of %d cases" % (d, ci, len(cases)))
    print("# rocprofv3 --pmc passes over tools/kernel_only.py run_pmc (%d launches per case, means per launch).  fetch = FETCH_SIZE x 1024 x 2"
          " (gfx950 wide-read correction: an upper bound for gathers), write = WRITE_SIZE x 1024; L2 hit = TCC_HIT_sum / (HIT + MISS)" % REPS)
    for c, ks in zip(cases, per_case):
        tot_f = tot_w = 0.0
        parts = []
        for k, v in ks.items():
            f = v.get("FETCH_SIZE", float("nan")) * 1024 * 2 * v["launches_per_call"]
            w = v.get("WRITE_SIZE", float("nan")) * 1024 * v["launches_per_call"]
            hit = v.get("TCC_HIT_sum")
            miss = v.get("TCC_MISS_sum")
            tot_f += f
            tot_w += w
            parts.append("%s: fetch %.2f MB write %.2f MB%s" % (k, f / 1e6, w / 1e6, "" if hit is None or miss is None or hit + miss == 0
                                                                  else " L2 hit %.3f" % (hit / (hit + miss))))
        print("%-74s alg %8.2f MB  fetch %8.2f MB  write %8.2f MB  traffic/alg %.2f   [%s]" % (
            c["name"], c["alg_bytes"] / 1e6, tot_f / 1e6, tot_w / 1e6, (tot_f + tot_w) / c["alg_bytes"], "; ".join(parts)))


def report(prof_dir, cases_json):
    cases = json.load(open(cases_json))
    db = glob.glob(prof_dir + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()

    def is_sep(name):
        return "recmv::" not in name and "sort" in name.lower()

    # a case = the recmv launches between one separator (sort) and the next; cases without REPS launches of some
    # kernel are the single warm-up calls and are skipped
    out, body, ci = [], None, 0
    for name, s, e in rows + [("sort (sentinel)", 0, 0)]:
        if is_sep(name):
            if body and max(len(v) for v in body.values()) >= REPS and ci < len(cases):
                out.append((cases[ci], body))
                ci += 1
            body = {}
        elif body is not None and "recmv::" in name:
            body.setdefault(name, []).append(e - s)
    print("# kernel-only durations (rocprofv3 --kernel-trace), %d launches per case; GB/s = algorithmic bytes / sum of the"
          " case's kernels" % REPS)
    for c, body in out:
        tot, parts = 0.0, []
        for k, d in body.items():
            tot += sum(d) / REPS
            short = k.split("recmv::(anonymous namespace)::")[-1].split("(")[0]
            parts.append("%s %.1f us x%d" % (short, (sum(d) / len(d)) / 1e3, max(len(d) // REPS, 1)))
        gbs = c["alg_bytes"] / (tot * 1e-9) / 1e9
        print("%-78s %9.1f us  %8.1f GB/s  %.3f of 8 TB/s   [%s]" % (c["name"], tot / 1e3, gbs, gbs * 1e9 / HBM_PEAK,
                                                                    "; ".join(parts)))
    if ci != len(cases):
        print("# WARNING: matched %d of %d cases" % (ci, len(cases)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    elif sys.argv[1] == "run_sampler":
        run_sampler(sys.argv[2])
    elif sys.argv[1] == "run_mc":
        run_mc(sys.argv[2])
    elif sys.argv[1] == "run_interp":
        run_interp(sys.argv[2])
    elif sys.argv[1] == "run_pmc":
        run_pmc(sys.argv[2])
    elif sys.argv[1] == "report_pmc":
        report_pmc(sys.argv[2], *sys.argv[3:])
    else:
        report(sys.argv[2], sys.argv[3])
