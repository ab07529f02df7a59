"""Interleaved A/B of runtime switches on ONE evolving scene: after the settle phase the configurations take turns in blocks of a
few iterations (the scene drifts slowly, neighbouring blocks see nearly the same meshes and ray counts), so that the comparison does
not depend on where a chaotic settle phase lands for each build ("the scene of the line moves with the code").

    python tools/ab_interleaved.py rows          # the row-tile MLP passes: off / 16-row / 32-row workgroups
"""
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402

import bench  # noqa: E402


def rows_cfg(on, lo=2304, hi=4096, rt=0):
    def apply():
        import recmv.chains as chains
        from recmv import _lib as L
        os.environ["RECMV_MLP_ROWS"] = "1" if on else "0"
        chains.MLP_ROWS_MIN, chains.MLP_ROWS_MAX = lo, hi
        L.check(L.lib().recmv_set_mlp_rows_tile(rt), "rows tile")
    return apply


def env_cfg(**kw):
    def apply():
        for k, v in kw.items():
            os.environ[k] = v
    return apply


CONFIGS = {
    "render_streams": [("render-loss terms of both garments on the ray stream", env_cfg(RECMV_RENDER_STREAMS="0")),
                       ("second garment's render-loss terms on a side stream", env_cfg(RECMV_RENDER_STREAMS="1"))],
    "prop_joint": [("propagateTmpPsGrad garment by garment", env_cfg(RECMV_PROP_JOINT="0")),
                   ("propagateTmpPsGrad for both garments as one block of rows", env_cfg(RECMV_PROP_JOINT="1"))],
    "fused_regu": [("deformation regulariser in ~90 torch launches (closed-form singular values)", env_cfg(RECMV_FUSED_REGU="0")),
                   ("deformation regulariser: value + gradient from one launch (csrc/def_regu.hip)", env_cfg(RECMV_FUSED_REGU="1"))],
    "jets": [("two jet passes per net (RECMV_MERGE_JETS=0)", env_cfg(RECMV_MERGE_JETS="0")),
             ("one jet pass per net over eikonal points + converged rays", env_cfg(RECMV_MERGE_JETS="1"))],
    "rows": [("per-layer chains", rows_cfg(False)), ("rows 16, 2304..4096", rows_cfg(True, 2304, 4096, 1)),
             ("rows 16, 3328..4096", rows_cfg(True, 3328, 4096, 1)),
             ("rows 16 <= 4096 / 32 <= 8192, from 2304", rows_cfg(True, 2304, 8192, 0))],
}


def main():
    which = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    block = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    torch.set_num_threads(8)
    dev = torch.device("cuda", 0)
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    cfgs = CONFIGS[which]
    cfgs[0][1]()
    loop = HotLoop(conf, dev, stage="coarse", curves=True, **bench.HOTLOOP_KW)
    it = 0
    for _ in range(240):
        loop.step(it)
        it += 1
    torch.cuda.synchronize()
    acc = {name: [] for name, _ in cfgs}
    rays = {name: [] for name, _ in cfgs}
    for r in range(rounds):
        for name, apply in cfgs:
            apply()
            if loop.forward_time % loop.remesh_intersect > loop.remesh_intersect - block - 2:
                loop.forward_time = 1                       # keep the re-mesh out of the blocks (same for every configuration)
            loop.step(it)
            it += 1
            torch.cuda.synchronize()
            for _ in range(block):                           # every iteration timed on its own: an epoch's short last batch (one
                t0 = time.perf_counter()                     # frame instead of three, ~75 ms) is dropped below instead of
                _, nr = loop.step(it)                        # landing in one configuration's mean
                torch.cuda.synchronize()
                acc[name].append((time.perf_counter() - t0) * 1e3)
                rays[name].append(int(nr))
                it += 1
    all_rays = sorted(r for v in rays.values() for r in v)
    full = 0.8 * all_rays[len(all_rays) // 2]
    print("# %d rounds x %d iterations per configuration, interleaved on one scene (settled 240 iterations on the first configuration, "
          "no re-mesh afterwards); ms per iteration (wall, synchronised after every iteration); iterations on an epoch's short last "
          "batch (rays < %.0f) left out; MC vertices at the end %s" % (rounds, block, full, [int(v.shape[0]) for v in loop.garment_vs]))
    kept = {name: [t for t, r in zip(acc[name], rays[name]) if r >= full] for name, _ in cfgs}
    base = sum(kept[cfgs[0][0]]) / len(kept[cfgs[0][0]])
    for name, _ in cfgs:
        v = sorted(kept[name])
        mean = sum(v) / len(v)
        print("%-44s mean %8.2f ms   median %7.2f   min %7.2f   %+5.1f %%   (%d iterations, rays %.0f)" % (
            name, mean, v[len(v) // 2], v[0], (mean / base - 1) * 100, len(v),
            sum(r for r in rays[name] if r >= full) / len(v)))


if __name__ == "__main__":
    main()
