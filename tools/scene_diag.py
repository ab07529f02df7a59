"""Which part of the frozen-scene recipe (bench.save_scene / load_scene) changes how the optimisation continues?

Runs bench.py's loop 240 iterations from its seed, keeps the exact state in memory, continues 30 iterations (the natural run), then
rebuilds the loop and continues from variants of the saved state: exact everything / first moments primed / f16 matrices / bf16
second moments / the file's recipe.  Prints, per variant, the converged rays per iteration and the MC vertex counts of the re-mesh
15 iterations in.      python tools/scene_diag.py
"""
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import torch  # noqa: E402

import bench  # noqa: E402
from recmv.hocon import ConfigFactory  # noqa: E402
from recmv.loop import HotLoop  # noqa: E402

DEV = torch.device("cuda", 0)


def build():
    conf = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    return HotLoop(conf, DEV, stage="coarse", curves=True, **bench.HOTLOOP_KW)


def snapshot(loop):
    params = [q for g in loop.optimizer.param_groups for q in g['params']]
    return dict(tensors={k: v.detach().clone() for k, v in bench._scene_tensors(loop).items()},
                adam=[{k: (v.clone() if torch.is_tensor(v) else v) for k, v in loop.optimizer.state[q].items()} if q in loop.optimizer.state
                      else None for q in params], opt_times=loop.opt_times)


def restore(loop, snap, f16=False, v_bf16=False, prime=False):
    mine = bench._scene_tensors(loop)

    def val(k):
        v = snap["tensors"][k]
        return v.half().float() if (f16 and v.is_floating_point() and v.numel() >= (1 << 14)) else v
    with torch.no_grad():
        for k, dst in mine.items():
            dst.copy_(val(k))
    params = [q for g in loop.optimizer.param_groups for q in g['params']]
    for q, st in zip(params, snap["adam"]):
        if st is not None:
            v = st['exp_avg_sq'].bfloat16().float() if v_bf16 else st['exp_avg_sq'].clone()
            loop.optimizer.state[q] = {'step': st['step'].clone(), 'exp_avg': torch.zeros_like(q) if prime else st['exp_avg'].clone(),
                                       'exp_avg_sq': v}
    loop.opt_times = snap["opt_times"]
    ratio = {'sdfRatio': 1., 'deformerRatio': loop.opt_times / 2500. + 0.5, 'renderRatio': 1.}
    loop.marching_cube_update(ratio)
    loop.forward_time = 1
    if prime:
        verts0 = [v.detach().clone() for v in loop.garment_vs]
        for k in range(16):
            loop.step(240 + k)
            with torch.no_grad():
                for kk, dst in mine.items():
                    dst.copy_(val(kk))
                for v, v0 in zip(loop.garment_vs, verts0):
                    v.copy_(v0)
            loop.opt_times = snap["opt_times"]
        loop.forward_time = 1


def continue_run(loop, tag, n=30, remesh_at=15):
    conv = []
    loop.forward_time = 1
    for k in range(n):
        if k == remesh_at:
            loop.forward_time = 0
        loop.step(240 + k)
        conv.append(sum(loop.info.get('rays_converged', [])))
        if k == remesh_at:
            torch.cuda.synchronize()
            print("%-34s re-mesh after %d its: MC vertices %s" % (tag, remesh_at, [int(v.shape[0]) for v in loop.garment_vs]), flush=True)
    print("%-34s converged rays per iteration: %s" % (tag, conv), flush=True)


def main():
    torch.set_num_threads(8)
    loop = build()
    for it in range(240):
        loop.step(it)
    torch.cuda.synchronize()
    snap = snapshot(loop)
    print("state at 240: MC vertices of the last re-mesh %s" % [int(v.shape[0]) for v in loop.garment_vs], flush=True)
    loop.forward_time = 0
    loop.step(240)                      # the natural run re-meshes here
    print("natural re-mesh at 240: %s" % [int(v.shape[0]) for v in loop.garment_vs], flush=True)
    continue_run(loop, "natural continuation", n=29, remesh_at=14)
    for tag, kw in (("exact state", {}), ("first moments primed", dict(prime=True)), ("f16 matrices", dict(f16=True)),
                    ("bf16 second moments", dict(v_bf16=True)), ("the file's recipe", dict(f16=True, v_bf16=True, prime=True))):
        loop = build()
        restore(loop, snap, **kw)
        torch.manual_seed(5)
        print("%-34s re-mesh at load: %s" % (tag, [int(v.shape[0]) for v in loop.garment_vs]), flush=True)
        continue_run(loop, tag)


if __name__ == "__main__":
    main()
