"""Frame-sharded data parallelism over RCCL (torch.distributed backend "nccl" on ROCm): two ranks, two GPUs, two
optimiser iterations — the replicas must hold bit-identical shared parameters, MC vertices and curve parameters after
every step (the same assertion as the gloo/CPU test, tests/test_loop_cpu.py::test_data_parallel_world2_gloo, here with
the HIP kernels and the real collective).  Needs 2 devices: skipped on the 1-GPU boxes."""
import os
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
CONF = str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf")


def _worker(rank, world, port, out_dir, backend="nccl", serial=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0", RECMV_SERIAL="1" if serial else "0")
    if backend == "gloo":                     # several ranks on ONE device: RCCL refuses that, gloo stages through the host
        os.environ.update(RECMV_SHARE_GPU0="1", RECMV_DIST_BACKEND="gloo")
    for p in (REPO / "rec-mv_amd", REPO):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    from recmv import dist as rdist
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    r, lr, w = rdist.init_distributed(backend)
    dev = torch.device("cuda", lr)
    torch.cuda.set_device(dev)
    conf = ConfigFactory.parse_file(CONF)
    conf.put('train.sample_pix_num', 256)
    loop = HotLoop(conf, dev, n_frames=12, H=160, W=128, resolutions=[(9, 13, 7), (17, 25, 13), (33, 49, 25), (65, 97, 49)],
                   skin_grid=(17, 33, 17), world_size=w, rank=r, seed=r, curves=True,    # different seeds: broadcast aligns
                   bbox=((-0.85, -1.2, -0.85), (0.85, 1.2, 0.85)))      # (the default box is sized from the rank's OWN initial nets)
    rdist.broadcast_state([p for p in loop.shared_parameters()] + list(loop.sdf.parameters())
                          + list(loop.inter_free_curve.parameters()) + list(loop.inter_free_curve.buffers()))
    allreduce = rdist.GradAllReduce(w)
    for it in range(2):
        loop.step(it, allreduce)
    torch.cuda.synchronize()
    torch.save({"params": torch.cat([p.detach().reshape(-1) for p in loop.shared_parameters()]).cpu(),
                "verts": torch.cat([v.detach().reshape(-1) for v in loop.garment_vs]).cpu(),
                "curves": torch.cat([p.detach().reshape(-1) for p in loop.inter_free_curve.parameters()]).cpu(),
                "frames": loop.frame_batch(1).cpu()}, os.path.join(out_dir, f"rank{r}.pt"))
    rdist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL refuses two ranks on one device)")
def test_data_parallel_world2_rccl(tmp_path):
    import torch.multiprocessing as mp
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert len(set(a["frames"].tolist()) & set(b["frames"].tolist())) == 0, "ranks take disjoint frames"
    assert torch.equal(a["params"], b["params"]), "shared parameters diverged across ranks"
    assert torch.equal(a["verts"], b["verts"]), "MC vertices diverged across ranks (needs deterministic MC order)"
    assert torch.equal(a["curves"], b["curves"]), "feature-curve parameters diverged across ranks"


def test_data_parallel_world2_three_streams_on_one_gpu(tmp_path):
    """Two frame-sharded ranks on ONE device (gloo collectives on device tensors): the three-stream schedule with its exchanges
    issued from the streams that need them (explicit vertices: main, curve parameters: curve stream, shared gradients: two
    asynchronous buckets around the implicit differentiation) leaves bit-identical replicas — and the same parameters as the
    reference's serial order on one stream (RECMV_SERIAL=1)."""
    import torch.multiprocessing as mp
    out = {}
    for k, serial in enumerate((False, True)):
        d = tmp_path / ("serial" if serial else "streams")
        d.mkdir()
        mp.spawn(_worker, args=(2, 29600 + ((os.getpid() + 11 * (k + 1)) % 300), str(d), "gloo", serial), nprocs=2, join=True)
        a, b = torch.load(d / "rank0.pt"), torch.load(d / "rank1.pt")
        assert len(set(a["frames"].tolist()) & set(b["frames"].tolist())) == 0, "ranks take disjoint frames"
        for key in ("params", "verts", "curves"):
            assert torch.equal(a[key], b[key]), (key, "diverged across ranks", "serial" if serial else "three streams")
        out[serial] = a
    for key in ("params", "verts", "curves"):
        assert torch.equal(out[False][key], out[True][key]), (key, "three-stream order differs from the serial order")
