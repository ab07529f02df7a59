"""Host logic of the hot loop and of the frame-sharded data parallelism, on CPU.

The loop code (recmv/loop.py) is the product's; here it runs on host cores through oracle/cpu_port.py
(torch + C oracle standing in for librecmv_hip.so).  The N>1 path is covered with a world_size-2 gloo job.
"""
import os
import sys
from pathlib import Path

import pytest
import torch

REPO = Path(__file__).resolve().parent.parent
CONF = str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf")


def _tiny_loop(world=1, rank=0, seed=0, curves=False, lr=None):
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    conf = ConfigFactory.parse_file(CONF)
    conf.put('train.sample_pix_num', 32)
    if lr is not None:
        conf.put('train.learning_rate', lr)
    return HotLoop(conf, 'cpu', n_frames=12, H=64, W=64, resolutions=[(9, 11, 7), (17, 21, 13)], skin_grid=(5, 9, 7),
                   bbox=((-0.9, -1.2, -0.6), (0.9, 1.2, 0.6)), world_size=world, rank=rank, seed=seed, curves=curves)


REF_CONFS = {       # the reference's own config files, parsed in place where /root/reference is mounted (this container)
    'anran': '/root/reference/configs/gap-female/config_anran_garment_10-5-1.conf',
    'leyang_jump': '/root/reference/configs/female_large_pose/leyang_jump_large_pose.conf',
    'female-3-casual': '/root/reference/configs/people_snapshot/female-3-casual.conf',
}


@pytest.mark.parametrize("capture,garments,lines,mask_keys,large_pose", [
    ('female-3-casual', ['long_sleeve_upper', 'long_pants'],
     ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'left_pant', 'right_pant'], ['upper', 'bottom'], False),
    ('anran', ['short_sleeve_upper', 'skirt'], ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'bottom_curve'],
     ['upper', 'bottom'], False),                                                     # BASELINE config C4
    ('leyang_jump', ['dress'], ['neck', 'left_cuff', 'right_cuff', 'bottom_curve'], ['upper_bottom'], True),   # C5
])
def test_garment_set_follows_the_capture_name(capture, garments, lines, mask_keys, large_pose):
    """`train.garment_type` -> TEMPLATE_GARMENT / FL_INFOS / FL_EXTRACT, `train.is_upper_bottom` -> the union region
    (utils/constant.py:53-131; OptimGarmentNetwork.py:141-164, :670-676, :1894-1905): HotLoop builds one- and two-garment loops
    under the reference's names and takes an iteration with the feature-curve branch on.  The reference's own config file of the
    capture is the input where the reference tree is mounted (sizes cut down for the host); the synthetic config with the
    capture's name and switch otherwise."""
    from oracle import cpu_port
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    ref_conf = REF_CONFS[capture]
    if os.path.isfile(ref_conf):
        conf = ConfigFactory.parse_file(ref_conf)
        assert conf.get_string('train.garment_type') == capture
    else:
        conf = ConfigFactory.parse_file(CONF)
        conf.put('train.garment_type', capture)
        conf.put('train.is_upper_bottom', mask_keys == ['upper_bottom'])
    conf.put('train.sample_pix_num', 32)
    cpu_port.install()
    try:
        loop = HotLoop(conf, 'cpu', n_frames=12, H=64, W=64, resolutions=[(9, 11, 7), (17, 21, 13)], skin_grid=(5, 9, 7),
                       bbox=((-0.9, -1.2, -0.6), (0.9, 1.2, 0.6)), curves=True, large_pose=large_pose)
        assert loop.garment_names == garments and loop.garment_size == len(garments) == len(loop.garment_nets)
        assert loop.fl_names == lines and loop.mask_keys == mask_keys
        assert [n for g in garments for n in loop.fl_extract[g]] == lines
        assert loop.dataset.d_cond.shape[1] == 128 * (1 + len(garments))                    # train.py:107
        frame_ids = loop.frame_batch(0)
        datas = loop.dataset.get_batch(frame_ids, loop.mask_keys)
        assert set(mask_keys) <= set(datas) and not ({'upper', 'bottom', 'upper_bottom'} - set(mask_keys)) & set(datas)
        assert datas['fl_masks'].shape == (frame_ids.numel(), len(lines))
        l0, rays = loop.step(0)
        assert torch.isfinite(l0) and rays > 0
        for g in garments:
            assert f'{g}_grad_loss' in loop.info and f'pc_{g}_loss_sdf' in loop.info and f'pc_{g}_mask_loss' in loop.info
            assert f'{g}_project loss' in loop.info['fl_loss']
        # the waist disc exists only where the capture has a waist line (:794); no CURVE_AWARE capture here
        assert ('pc_upper_bottom_circle_loss_sdf' in loop.info) == ('upper_bottom' in lines)
        if large_pose:
            assert all(not p.requires_grad for net in loop.garment_nets for p in net.parameters())
        assert set(loop.state_dict()) >= {'garment_nets.%d.lin0.bias' % i for i in range(len(garments))}
    finally:
        cpu_port.uninstall()


def test_loop_two_steps_on_cpu_port():
    from oracle import cpu_port
    cpu_port.install()
    try:
        loop = _tiny_loop()
        before = [p.detach().clone() for p in loop.shared_parameters()]
        l0, rays = loop.step(0)
        # rays: a Bernoulli(sample_pix * N / pixels) subset of the rasterised surface pixels of each garment
        # (OptimGarmentNetwork.py:1019-1027): expectation 2 garments x 3 frames x 16, binomial spread
        assert torch.isfinite(l0) and 40 <= rays <= 160
        assert all(n > 3 * 16 for n in loop.info['surface_pixels']), "each garment covers more pixels than it samples"
        assert loop.body_vs.shape[0] > 0 and all(v.shape[0] > 0 for v in loop.garment_vs)
        for name in loop.garment_names:
            assert f'{name}_grad_loss' in loop.info and f'pc_{name}_loss_sdf' in loop.info
        l1, _ = loop.step(1)
        assert torch.isfinite(l1)
        changed = sum(int(not torch.equal(a, b.detach())) for a, b in zip(before, loop.shared_parameters()))
        assert changed > 10, "Adam updated the shared parameters"
        assert loop.forward_time == 2 and loop.opt_times == 2.0
    finally:
        cpu_port.uninstall()


def test_feature_curve_branch_on_cpu_port():
    """project_2d_loss (OptimGarmentNetwork.py:1772-1883) inside the iteration: curves deform, part of their samples
    is visible, the AdamW step moves the curve parameters, and the rest of the iteration is unaffected by the
    gradients the branch leaves behind (they are cleared by the optimiser's zero_grad)."""
    from oracle import cpu_port
    cpu_port.install()
    try:
        loop = _tiny_loop(curves=True)
        assert loop.fl_names == ['neck', 'left_cuff', 'right_cuff', 'upper_bottom', 'left_pant', 'right_pant']
        before = [p.detach().clone() for p in loop.inter_free_curve.parameters()]
        l0, _ = loop.step(0)
        circle0 = float(loop.info['pc_upper_bottom_circle_loss_sdf'])     # curve_aware_loss (:787-813) ran
        l1, _ = loop.step(1)
        assert torch.isfinite(l0) and torch.isfinite(l1) and circle0 > 0
        info = loop.info['fl_loss']
        assert torch.isfinite(info['total']) and info['total'] > 0
        for name in loop.garment_names:
            assert 0.0 <= float(info[f'{name}_visible']) <= 1.0
            assert float(info[f'{name}_project loss']) >= 0
        vis = [float(info[f'{name}_visible']) for name in loop.garment_names]
        assert 0.05 < max(vis) < 0.95, vis          # the body hides the far side of the rings, not all of them
        moved = [float((a - b.detach()).abs().max()) for a, b in zip(before, loop.inter_free_curve.parameters())]
        assert max(moved) > 1e-5 and max(moved) < 1e-2, moved              # two AdamW steps of lr 1e-4
        sd = loop.state_dict()
        assert 'inter_free_curve.scale' in sd and 'inter_free_curve.cano_smpl_verts' in sd   # reference key names
        other = _tiny_loop(curves=True)
        other.load_state_dict(sd)
        assert torch.equal(other.inter_free_curve.scale, loop.inter_free_curve.scale)
        plain = _tiny_loop(curves=False)
        assert not any(k.startswith('inter_free_curve') for k in plain.state_dict())
        pl0, _ = plain.step(0)
        # same seeds, same frames: the curve branch changes the first iteration's loss only by the curve-aware term
        # (pc_weight.curve_aware_weight x |SDF| on the `upper_bottom` fan); project_2d_loss has its own backward
        assert 'pc_upper_bottom_circle_loss_sdf' not in plain.info
        w = loop.conf.get_float('pc_weight.curve_aware_weight')
        assert abs(float(l0) - w * circle0 - float(pl0)) < 5e-2 * max(1.0, abs(float(pl0)))
    finally:
        cpu_port.uninstall()


def test_stage_switch_is_applied_at_the_next_remesh():
    """utils.set_hierarchical_config + OptimNetwork.update_hierarchical_config (utils/utils.py:330-348,
    OptimNetwork.py:79-117): batch size and pyramid change at once, loss weights / point radius / re-mesh period at the
    next scheduled re-mesh, which also restarts the re-mesh counter.  Both later stages of the shipped config load."""
    from oracle import cpu_port
    cpu_port.install()
    try:
        loop = _tiny_loop(lr=1e-6)        # Adam's first steps at 1e-4 move the SDF by more than this tiny box holds
        loop.step(0)
        assert loop.remesh_intersect == 30 and loop.batch_size == 3 and loop.forward_time == 1
        w_coarse = loop.conf.get_float('pc_weight.weight')
        loop.set_stage('medium', resolutions=[(9, 11, 7), (17, 21, 13)])
        assert loop.batch_size == 2 and loop.next_conf is not None
        assert loop.remesh_intersect == 30 and loop.conf.get_float('pc_weight.weight') == w_coarse   # still parked
        loop.step(1)
        assert loop.forward_time == 2 and loop.next_conf is not None          # no re-mesh yet: nothing applied
        loop.forward_time = 30                                                # the next scheduled re-mesh
        loop.step(2)
        assert loop.next_conf is None and loop.remesh_intersect == 60 and loop.forward_time == 1
        assert loop.conf.get_float('pc_weight.weight') == 30. and abs(loop.pc_radius - 0.00465) < 1e-9
        loop.set_stage('fine', resolutions=[(9, 11, 7), (17, 21, 13)])
        loop.forward_time = 60
        l, _ = loop.step(3)
        assert torch.isfinite(l) and loop.batch_size == 1 and loop.remesh_intersect == 120
        assert loop.conf.get_int('sample_pix_num') == 6144
    finally:
        cpu_port.uninstall()


def test_epoch_keeps_the_short_last_batch():
    """DataLoader(drop_last=False): ceil(F / batch) iterations per epoch, the last one short; the same count feeds
    train.resumed_opt_times.  With ranks, the last position still gives every rank a frame."""
    from recmv.loop import iters_per_epoch
    from oracle import cpu_port
    cpu_port.install()
    try:
        loop = _tiny_loop()
        loop.dataset.F = 11
        assert loop.iters_per_epoch() == 4 == iters_per_epoch(11, 3, 1)
        seen = torch.cat([loop.frame_batch_at(0, pos) for pos in range(4)])
        assert sorted(seen.tolist()) == list(range(11)) and loop.frame_batch_at(0, 3).numel() == 2
        pair = [_tiny_loop(world=2, rank=r) for r in range(2)]
        for lp in pair:
            lp.dataset.F = 13                       # 13 = 2 * 6 + 1: the last position has one frame for two ranks
        assert pair[0].iters_per_epoch() == 3
        a, b = pair[0].frame_batch_at(0, 2), pair[1].frame_batch_at(0, 2)
        assert a.numel() == 1 and b.numel() == 1 and a.item() != b.item()
    finally:
        cpu_port.uninstall()


def test_frame_sharding_is_a_partition():
    from oracle import cpu_port
    cpu_port.install()
    try:
        loops = [_tiny_loop(world=2, rank=r) for r in range(2)]
        single = _tiny_loop(world=1)
        for it in range(4):
            a, b = loops[0].frame_batch(it), loops[1].frame_batch(it)
            assert a.numel() == b.numel() == loops[0].batch_size
            assert len(set(a.tolist()) & set(b.tolist())) == 0, "ranks take disjoint frames"
        assert single.frame_batch(0).numel() == single.batch_size
    finally:
        cpu_port.uninstall()


def _dp_worker(rank, world, port, out_dir, order="serial"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), RECMV_OVERLAP_ORDER="1" if order == "overlap" else "0")
    for p in (REPO / "rec-mv_amd", REPO):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    torch.set_num_threads(2)
    from oracle import cpu_port
    from recmv import dist as rdist
    cpu_port.install()
    r, _, w = rdist.init_distributed("gloo")
    loop = _tiny_loop(world=w, rank=r, seed=r, curves=True)   # different seeds: broadcast must make them agree
    rdist.broadcast_state([p for p in loop.shared_parameters()] + list(loop.sdf.parameters())
                          + list(loop.inter_free_curve.parameters()) + list(loop.inter_free_curve.buffers()))
    allreduce = rdist.GradAllReduce(w)
    # the curve-aware disc draws its samples from a generator of its own: on the host the loop's other draws (ray subset, eikonal
    # points) share ONE generator with it, and the two orders reach the disc at different positions of that stream (on the
    # device the ray subset comes from the host generator and the rest from the device's, in the same order either way)
    from recmv.loop import HotLoop, sample_fan_mesh
    gen = torch.Generator().manual_seed(77)
    loop.curve_aware_loss = lambda ratio: HotLoop.curve_aware_loss(
        loop, ratio, sampler=lambda v, f, n: sample_fan_mesh(v, f, 4000, generator=gen))
    for it in range(2):
        loop.step(it, allreduce)
    flat = torch.cat([p.detach().reshape(-1) for p in loop.shared_parameters()])
    verts = torch.cat([v.detach().reshape(-1) for v in loop.garment_vs])
    curv = torch.cat([p.detach().reshape(-1) for p in loop.inter_free_curve.parameters()])
    torch.save({"params": flat, "verts": verts, "curves": curv}, os.path.join(out_dir, f"rank{r}.pt"))
    rdist.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_world2_gloo(tmp_path):
    """Two ranks, disjoint frames, one all-reduce of the shared gradients per optimiser step (+ one of the explicit
    MC-vertex gradients): after every step the replicas must hold IDENTICAL shared parameters and MC vertices."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(a["params"], b["params"]), "shared parameters diverged across ranks"
    assert torch.equal(a["verts"], b["verts"]), "MC vertices diverged across ranks (needs deterministic MC order)"
    assert torch.equal(a["curves"], b["curves"]), "feature-curve parameters diverged across ranks"


def test_overlapped_order_equals_serial_order_world2_gloo(tmp_path):
    """The dependency-graph order of an iteration (mask loss -> ray pipeline -> curve branch -> |SDF| terms -> render loss; on the
    device: three streams) with its three exchanges — explicit vertices, curve parameters, shared gradients in two asynchronous
    buckets around the implicit differentiation — issued in that order on both ranks, against the reference's serial order: after
    two optimiser steps the shared parameters, the explicit vertices and the curve parameters are bit-identical between the
    orders and between the ranks."""
    import torch.multiprocessing as mp
    out = {}
    for k, order in enumerate(("serial", "overlap")):
        d = tmp_path / order
        d.mkdir()
        mp.spawn(_dp_worker, args=(2, 29500 + ((os.getpid() + 7 * (k + 1)) % 1000), str(d), order), nprocs=2, join=True)
        out[order] = [torch.load(d / "rank0.pt"), torch.load(d / "rank1.pt")]
        for key in ("params", "verts", "curves"):
            assert torch.equal(out[order][0][key], out[order][1][key]), (order, key)
    for key in ("params", "verts", "curves"):
        assert torch.equal(out["serial"][0][key], out["overlap"][0][key]), key


def _capture_loop(capture, large_pose, world, rank, seed):
    """A cut-down loop of one of the reference's captures: its own config file where /root/reference is mounted, the synthetic
    config under the capture's name otherwise (as test_garment_set_follows_the_capture_name)."""
    from recmv.hocon import ConfigFactory
    from recmv.loop import HotLoop
    ref_conf = REF_CONFS[capture]
    if os.path.isfile(ref_conf):
        conf = ConfigFactory.parse_file(ref_conf)
    else:
        conf = ConfigFactory.parse_file(CONF)
        conf.put('train.garment_type', capture)
        conf.put('train.is_upper_bottom', capture == 'leyang_jump')
    conf.put('train.sample_pix_num', 32)
    return HotLoop(conf, 'cpu', n_frames=12, H=64, W=64, resolutions=[(9, 11, 7), (17, 21, 13)], skin_grid=(5, 9, 7),
                   bbox=((-0.9, -1.2, -0.6), (0.9, 1.2, 0.6)), world_size=world, rank=rank, seed=seed, curves=True,
                   large_pose=large_pose)


def _capture_dp_worker(rank, world, port, out_dir, capture, large_pose):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (REPO / "rec-mv_amd", REPO):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    torch.set_num_threads(2)
    from oracle import cpu_port
    from recmv import dist as rdist
    from recmv.loop import HotLoop, sample_fan_mesh
    cpu_port.install()
    r, _, w = rdist.init_distributed("gloo")
    loop = _capture_loop(capture, large_pose, w, r, seed=r)           # different seeds: the broadcast must make the replicas agree
    rdist.broadcast_state([p for p in loop.shared_parameters()] + list(loop.sdf.parameters())
                          + [p for n in loop.garment_nets for p in n.parameters()]
                          + list(loop.inter_free_curve.parameters()) + list(loop.inter_free_curve.buffers()))
    allreduce = rdist.GradAllReduce(w)
    exchanged = []                                                     # every tensor list that enters an exchange
    real_start, real_call = allreduce.start, allreduce.__call__

    class Spy:
        def __call__(self, ts):
            exchanged.append([id(t) for t in ts])
            return allreduce(ts)

        def start(self, ts):
            exchanged.append([id(t) for t in ts])
            return allreduce.start(ts)

        def finish(self, h):
            return allreduce.finish(h)
    gen = torch.Generator().manual_seed(77)
    loop.curve_aware_loss = lambda ratio: HotLoop.curve_aware_loss(
        loop, ratio, sampler=lambda v, f, n: sample_fan_mesh(v, f, 4000, generator=gen))
    for it in range(2):
        loop.step(it, Spy())
    sdf_ids = {id(p) for n in list(loop.garment_nets) + [loop.sdf] for p in n.parameters()}
    shared = loop.shared_parameters()
    torch.save({"params": torch.cat([p.detach().reshape(-1) for p in shared]),
                "verts": torch.cat([v.detach().reshape(-1) for v in loop.garment_vs]),
                "curves": torch.cat([p.detach().reshape(-1) for p in loop.inter_free_curve.parameters()]),
                "sdf": torch.cat([p.detach().reshape(-1) for n in loop.garment_nets for p in n.parameters()]),
                "sdf_tensors_exchanged": sum(1 for ids in exchanged for i in ids if i in sdf_ids),
                "shared_grad_bytes": int(sum(p.numel() for p in shared) * 4),
                "sdf_bytes": int(sum(p.numel() for n in loop.garment_nets for p in n.parameters()) * 4),
                "garments": list(loop.garment_names), "exchanges": len(exchanged)}, os.path.join(out_dir, f"rank{r}.pt"))
    rdist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("capture,large_pose,garments", [
    ('anran', False, ['short_sleeve_upper', 'skirt']),        # BASELINE configs[3]: CUHKszCap scene, frames sharded
    ('leyang_jump', True, ['dress']),                         # BASELINE configs[4]: large-pose stage (train_large_pose.py:289-344)
])
def test_data_parallel_world2_gloo_on_the_scene_captures(tmp_path, capture, large_pose, garments):
    """The frame-sharded path on the captures BASELINE configs[3] / [4] name — `configs/gap-female/config_anran_garment_10-5-1.conf`
    (garments short_sleeve_upper + skirt, optimisation stage) and the large-pose stage of female_large_pose (one dress, SDF nets
    frozen: OptimGarmentNetwork_Large_Pose.py:122-137) — as a world-2 gloo job: after two optimiser steps both replicas hold
    bit-identical shared parameters, explicit vertices and curve parameters; in the large-pose stage the frozen SDF nets stay
    bit-identical to their broadcast state WITHOUT entering any exchange (zero bytes), and the exchanged volume shrinks by exactly
    the garment nets' size."""
    import torch.multiprocessing as mp
    port = 29500 + ((os.getpid() + (97 if large_pose else 53)) % 1000)
    mp.spawn(_capture_dp_worker, args=(2, port, str(tmp_path), capture, large_pose), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert a["garments"] == garments
    for key in ("params", "verts", "curves", "sdf"):
        assert torch.equal(a[key], b[key]), key
    assert a["exchanges"] >= 2 * 3                             # per step: explicit vertices, curves, shared gradients
    if large_pose:
        assert a["sdf_tensors_exchanged"] == 0 and b["sdf_tensors_exchanged"] == 0
        # what the optimisation stage would exchange on top: the garment net(s)
        assert a["sdf_bytes"] > 7e6 and a["shared_grad_bytes"] < 8e6
    else:
        assert a["sdf_tensors_exchanged"] > 0 and a["shared_grad_bytes"] > a["sdf_bytes"] > 15e6


def test_grad_allreduce_start_finish_two_in_flight(tmp_path):
    """Two exchanges in flight at once (start, start, finish, finish) use two staging buffers and leave the averages."""
    import torch.multiprocessing as mp
    mp.spawn(_two_in_flight_worker, args=(2, 29500 + ((os.getpid() + 31) % 1000)), nprocs=2, join=True)


def _two_in_flight_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (REPO / "rec-mv_amd", REPO):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    from recmv import dist as rdist
    r, _, w = rdist.init_distributed("gloo")
    a, b, c = (torch.nn.Parameter(torch.zeros(n)) for n in (5, 7, 3))
    a.grad, b.grad = torch.full((5,), float(r)), torch.full((7,), 10.0 + r)        # c has no gradient on either rank
    ar = rdist.GradAllReduce(w)
    h1 = ar.start([a])
    h2 = ar.start([b, c])
    assert h1[1].data_ptr() != h2[1].data_ptr()
    ar.finish(h1)
    ar.finish(h2)
    assert torch.equal(a.grad, torch.full((5,), 0.5)) and torch.equal(b.grad, torch.full((7,), 10.5)) and torch.equal(c.grad, torch.zeros(3))
    ar([a])                                          # the buffers are free again
    assert torch.equal(a.grad, torch.full((5,), 0.5)) and sum(len(p) for p in ar._flat.values()) == 2
    rdist.barrier()
    torch.distributed.destroy_process_group()


def test_grad_allreduce_handles_missing_grads_single_process():
    from recmv.dist import GradAllReduce
    p = torch.nn.Parameter(torch.ones(3))
    GradAllReduce(1)([p])                                     # world 1: no-op, no process group needed
    assert p.grad is None
