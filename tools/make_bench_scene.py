"""Produce the frozen benchmark scene `configs/synthetic/bench_scene_v1.pt` (bench.load_scene reads it).

    python tools/make_bench_scene.py [--iters 240] [--out gpurun_out/bench_scene_v1.pt]

Builds bench.py's loop from its seed (configs/synthetic/people_snapshot_like.conf, 64 synthetic frames of 512 x 512, curve branch
on), runs `--iters` optimiser iterations — 240 = eight re-mesh periods: past Adam's start-up transient, the next step re-meshes —
and writes everything the optimisation moved (bench.save_scene).  Run ONCE (round 6, on the round-5 kernels); from then on the
timed workload is a function of the committed file, not of the code under test.  The file defines the scene: a later run of this
script with other kernels lands somewhere else (the optimisation is chaotic) and would be a NEW scene (v2), not a refresh.
"""
import argparse
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=240)
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "bench_scene_v1.pt"))
    a = ap.parse_args()
    import bench
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    dev = torch.device("cuda", 0)
    torch.set_num_threads(8)
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    loop = HotLoop(conf, dev, stage="coarse", curves=True, **bench.HOTLOOP_KW)
    for it in range(a.iters):
        loop.step(it)
        if (it + 1) % 30 == 0:
            torch.cuda.synchronize()
            print("iteration %d: rays converged %s of %s, MC vertices %s" % (
                it + 1, loop.info.get("rays_converged"), loop.info.get("rays_total"), [int(v.shape[0]) for v in loop.garment_vs]),
                flush=True)
    assert loop.forward_time % loop.remesh_intersect == 0, "the scene is taken where the next step re-meshes"
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ""
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    bench.save_scene(loop, a.out, a.iters - 1,
                     note="%d iterations from the seeded state of bench.py's loop (tools/make_bench_scene.py, kernels of %s)"
                          % (a.iters, head or "round 5 / start of round 6"))
    print("wrote %s (%.1f MB)" % (a.out, Path(a.out).stat().st_size / 1e6))


if __name__ == "__main__":
    main()
