"""`python bench.py --gpus N` must start N ranks itself and never measure fewer.

The driver's SCALE step calls `bench.py --gpus N` (and, for N > 1, the same under torch.distributed.run).  Both forms are covered
here without a GPU: the launcher's refusals on the real script, and the whole entry — launcher, two gloo ranks, timed region, JSON
line — through tests/bench_cpu_entry.py on the CPU port with a scene cut down for host cores.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
SMALL = ["--steps", "2", "--warmup", "1", "--settle-iters", "0", "--scene", "none", "--no-mc", "--no-hbm-kernels", "--no-cpu-baseline", "--no-config2",
         "--no-alt-mode", "--no-kernel-events", "--no-curves"]


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="2", **kw)
    return env


def test_gpus_n_without_n_devices_exits_non_zero():
    """No GPU in this container: `--gpus 2` must refuse (exit 2) instead of running one rank and printing n_gpus 1."""
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2"] + SMALL, env=_env(HIP_VISIBLE_DEVICES="",
                       CUDA_VISIBLE_DEVICES=""), capture_output=True, text=True, timeout=600)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "refusing to measure fewer ranks" in r.stderr and not r.stdout.strip()


def test_world_size_that_contradicts_gpus_is_refused():
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "4"] + SMALL,
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr and not r.stdout.strip()


def _line(args, **env):
    r = subprocess.run([sys.executable, str(REPO / "tests" / "bench_cpu_entry.py")] + args + SMALL, env=_env(**env),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line: %r" % r.stdout[-500:]
    return json.loads(lines[0])


def test_gpus_2_launches_two_ranks_with_identical_replicas():
    """The entry with `--gpus 2` and no torchrun environment: two ranks over gloo, one line with n_gpus 2, a step time per rank,
    the shared-gradient exchange timed, replicas bit-identical after the run; twice the frames of the one-rank job per step."""
    two = _line(["--gpus", "2"])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["steps"] == 2
    cfg = two["config"]
    assert len(cfg["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in cfg["per_rank_ms_per_step"])
    assert cfg["shared_grad_allreduce_us"] > 0 and cfg["shared_grad_bytes"] > 1e6
    assert cfg["replicas_bit_identical"] is True
    assert "dp2" in cfg["parallelism"]
    assert abs(two["value"] - 2 * 1e3 / two["ms_per_step"]) < 1e-2 * two["value"]       # whole-job rate: N iterations per step time
    one = _line(["--gpus", "1"])
    assert one["n_gpus"] == 1 and one["config"]["per_rank_ms_per_step"] is None and one["config"]["replicas_bit_identical"] is None


def test_frames_at_counts_the_short_last_batch_of_an_epoch():
    """bench.frames_at: frames a rank takes in iteration `it` — batch_size except at an epoch's last position (64 frames, 3 per step:
    every 22nd iteration has one frame), the same deal HotLoop.frame_batch_at makes, also when frames are sharded over ranks."""
    import types
    sys.path.insert(0, str(REPO))
    import bench
    from recmv.loop import iters_per_epoch

    def fake(F, bs, world, rank):
        loop = types.SimpleNamespace(batch_size=bs, world_size=world, rank=rank, dataset=types.SimpleNamespace(F=F))
        loop.iters_per_epoch = lambda: iters_per_epoch(F, bs, world)
        return loop
    one = fake(64, 3, 1, 0)
    assert one.iters_per_epoch() == 22
    assert [bench.frames_at(one, it) for it in range(22)] == [3] * 21 + [1]
    assert bench.frames_at(one, 22 + 5) == 3 and bench.frames_at(one, 43) == 1
    two = [fake(64, 3, 2, r) for r in range(2)]
    assert two[0].iters_per_epoch() == 11
    assert [bench.frames_at(two[0], it) for it in range(11)] == [3] * 10 + [2]
    assert [bench.frames_at(two[1], it) for it in range(11)] == [3] * 10 + [2]
    # fewer remaining frames than ranks: the permutation wraps so that every rank still has one
    three = [fake(10, 3, 3, r) for r in range(3)]            # 9 per step, 2 positions, the second holds 1 frame for 3 ranks
    assert three[0].iters_per_epoch() == 2
    assert [bench.frames_at(three[r], 1) for r in range(3)] == [1, 1, 1]


def test_frozen_scene_round_trip_on_the_cpu_port(tmp_path):
    """bench.save_scene / load_scene: the tensors the optimisation moves and Adam's two moments go through the file (matrices of
    >= 2^14 elements as f16, moments as bf16), the loaded loop re-meshes from the file's SDF nets and continues right after that
    re-mesh; a file that does not describe the loop is refused."""
    import pytest
    import torch
    for p in (REPO / "rec-mv_amd", REPO):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    import bench
    from oracle import cpu_port
    sys.path.insert(0, str(REPO / "tests"))
    from test_loop_cpu import _tiny_loop
    cpu_port.install()
    try:
        a = _tiny_loop(curves=True)
        for it in range(3):
            a.step(it)
        path = str(tmp_path / "scene.pt")
        bench.save_scene(a, path, 2, note="test")
        b = _tiny_loop(curves=True, seed=5)                       # another seed: everything the file holds must be overwritten
        it0 = bench.load_scene(b, path)
        assert it0 == 3 and b.forward_time == 1 and b.opt_times == a.opt_times
        ta, tb = bench._scene_tensors(a), bench._scene_tensors(b)
        assert set(ta) == set(tb) and any(k.startswith("model.inter_free_curve.") for k in ta) and "dataset.poses" in ta
        for k in ta:
            x, y = ta[k].detach(), tb[k].detach()
            if x.is_floating_point() and x.numel() >= (1 << 14):
                assert torch.equal(y, x.half().float()), k        # the file's rounding, nothing else
            else:
                assert torch.equal(y, x), k
        pa = [q for g in a.optimizer.param_groups for q in g['params']]
        pb = [q for g in b.optimizer.param_groups for q in g['params']]
        seen = 0
        for qa, qb in zip(pa, pb):
            if qa in a.optimizer.state:
                sa, sb = a.optimizer.state[qa], b.optimizer.state[qb]
                assert float(sb['step']) == float(sa['step']) == 3.0
                assert torch.equal(sb['exp_avg'], sa['exp_avg'].bfloat16().float())
                assert torch.equal(sb['exp_avg_sq'], sa['exp_avg_sq'].bfloat16().float())
                seen += 1
        assert seen > 20
        assert [tuple(v.shape) for v in b.garment_vs] and all(v.requires_grad for v in b.garment_vs)
        loss, rays = b.step(it0)                                  # and the loop runs on from there
        assert torch.isfinite(loss) and rays > 0 and b.forward_time == 2
        c = _tiny_loop(curves=False)                              # no curve branch: the file has tensors this loop lacks
        with pytest.raises(SystemExit):
            bench.load_scene(c, path)
    finally:
        cpu_port.uninstall()
