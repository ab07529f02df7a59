"""Data path: recmv.dataset reads a synthetic capture directory exactly as the reference's dataset classes
do (tests/golden/dataset.npz was produced by the reference's own `SceneDataset` / `People_Snapshot_SceneDataset` / samplers
on the identical directory, tests/golden/make_golden_dataset.py)."""
import random
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

HERE = Path(__file__).resolve().parent
sys.path[:0] = [str(HERE), str(HERE.parent / "rec-mv_amd")]
import capture_fixture as cf  # noqa: E402

CONDS = {'deformer': 16, 'render': 8}


@pytest.fixture(scope="module")
def golden():
    return {k: torch.from_numpy(v) for k, v in np.load(HERE / "golden" / "dataset.npz").items()}


@pytest.fixture(scope="module")
def capture(tmp_path_factory):
    return cf.write_capture(str(tmp_path_factory.mktemp("capture")))


def _compare(got, golden, kind):
    keys = [k for k in golden if k.startswith(kind + '_')]
    assert set(keys) == set(got), set(keys) ^ set(got)
    for k in keys:
        assert got[k].shape == golden[k].shape, k
        assert torch.allclose(got[k].float(), golden[k], rtol=1e-6, atol=1e-6), (k, float((got[k].float() - golden[k]).abs().max()))


def test_scene_dataset_matches_the_reference_class(capture, golden):
    from recmv.dataset import SceneDataset
    torch.manual_seed(11)
    ds = SceneDataset(capture, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30, curve_sampling=2)
    _compare(cf.collect(ds, 'scene'), {k: v for k, v in golden.items() if not k.endswith('apose')}, 'scene')
    assert ds.gender == 'female' and (ds.H, ds.W) == (cf.H, cf.W) and ds.frame_num == cf.FRAMES
    # learnable tensors: what the optimiser is built over (dataset.py:253-258)
    assert len(ds.learnable_weights()) == 2
    ds.poses.requires_grad_(True)
    ds.opt_camera_params(True)
    assert len(ds.learnable_weights()) == 2 + 4 + 1


@pytest.mark.parametrize("a_pose", [False, True])
def test_people_snapshot_dataset_matches_the_reference_class(capture, golden, a_pose):
    from recmv.dataset import People_Snapshot_SceneDataset
    torch.manual_seed(12)
    ds = People_Snapshot_SceneDataset(capture, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30, curve_sampling=1, a_pose=a_pose)
    kind = 'ps%d' % int(a_pose)
    got = cf.collect(ds, kind)
    got[kind + '_apose'] = torch.tensor([float(ds.a_pose_start), float(ds.a_pose_end)])
    _compare(got, golden, kind)


def test_samplers_match_the_reference_classes(capture, golden):
    from recmv.dataset import ClipSampler, RandomSampler, SceneDataset
    ds = SceneDataset(capture, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30)
    for name, cls, arg in (('random', RandomSampler, 3), ('clip', ClipSampler, 5)):
        for shuffle in (False, True):
            random.seed(5)
            torch.manual_seed(5)
            s = cls(ds, arg, shuffle)
            assert list(iter(s)) == golden['sampler_%s_%d' % (name, int(shuffle))].long().tolist()
            assert len(s) == int(golden['sampler_%s_%d_len' % (name, int(shuffle))])


def test_nearest_label_fill_and_preprocessing_step(capture, golden):
    from recmv.dataset import SceneDataset
    ds = SceneDataset(capture, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30)
    out = ds.load_parsing_mask(golden['fill_mask'], golden['fill_logits'].long())
    assert np.array_equal(out, golden['fill_out'].numpy().astype(np.uint8))
    # `parsing_mask(idx)` writes mask_parsing_<idx>.npy beside the labels: foreground fully labelled, background zero
    path = ds.parsing_mask(3)
    filled = np.load(path)
    mask = ds._read_mask(3).numpy() > 0
    assert filled.dtype == np.uint8 and (filled[mask] > 0).all() and (filled[~mask] == 0).all()


def test_loader_factory_and_collation(capture):
    from recmv.dataset import getDatasetAndLoader
    random.seed(1)
    torch.manual_seed(1)
    # (frames 1, 4, 7, 10 of the fixture have no normal map — an optional key the default collation cannot mix — so: pairs
    # of frames in order, of which the first pair (0, 1) is skipped)
    ds, loader = getDatasetAndLoader(capture, dict(CONDS), 1, False, 0, True, True, False, cf.GARMENT_TYPE, data_type='scene')
    assert ds.poses.requires_grad and ds.trans.requires_grad and not ds.camera_params['focal_length'].requires_grad
    assert len(loader) == cf.FRAMES
    ids, batch = next(iter(loader))
    assert ids.tolist() == [0] and batch['img'].shape == (1, cf.H, cf.W, 3) and batch['fl_pts'].shape == (1, 6 * 100, 2)
    assert batch['upper'].dtype == torch.bool and batch['fl_masks'].shape == (1, 6) and batch['normal'].shape == (1, cf.H, cf.W, 3)
    poses, trans, c0, c1 = ds.get_grad_parameters(ids, 'cpu')
    assert poses.shape == (1, 24, 3) and c0.shape == (1, 16) and c1.shape == (1, 8) and poses.requires_grad
    with pytest.raises(NotImplementedError):
        getDatasetAndLoader(capture, dict(CONDS), 3, True, 0, True, True, False, cf.GARMENT_TYPE, data_type='snug')


def test_synthe_dataset_matches_the_reference(capture, golden):
    """Synthe_SceneDataset (dataset/dataset.py:1004-1064): every frame keeps its feature lines whatever `curve_sampling` says
    (a SceneDataset with curve_sampling=2 blanks the odd frames), and the samples carry no 2-D joints."""
    from recmv.dataset import SceneDataset, Synthe_SceneDataset, getDatasetAndLoader
    torch.manual_seed(14)
    syn = Synthe_SceneDataset(capture, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30, curve_sampling=2)
    for idx in (1, 5):
        i, smp = syn[idx]
        assert i == idx
        assert torch.equal(smp['fl_masks'].float(), golden['synthe_s%d_fl_masks' % idx])
        torch.testing.assert_close(smp['fl_pts'].float(), golden['synthe_s%d_fl_pts' % idx], rtol=1e-6, atol=1e-5)
        assert [float(k in smp) for k in ('gt_joints2d', 'normal', 'upper', 'bottom')] == golden['synthe_s%d_keys' % idx].tolist()
    torch.manual_seed(14)
    plain = SceneDataset(capture, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30, curve_sampling=2)
    assert not plain[1][1]['fl_masks'].any() and syn[1][1]['fl_masks'].any()
    ds, _ = getDatasetAndLoader(capture, dict(CONDS), 3, True, 0, True, True, False, cf.GARMENT_TYPE, data_type='synthe')
    assert isinstance(ds, Synthe_SceneDataset) and ds.poses.requires_grad


def test_one_iteration_of_the_facade_on_a_capture_directory(capture):
    """train.py's sequence with a capture read by recmv.dataset instead of the synthetic frames: getOptNet(dataset, ...),
    a collated mini-batch of the DataLoader as `datas`, forward / backward / propagateTmpPsGrad / optimizer.step — the per-frame
    tensors that move are the DATASET's (poses, trans, DCT-initialised codes, camera), the 2-D feature lines and the garment
    regions come from the mini-batch, the per-line weights from the capture's statistic.  (CPU port of the kernels.)"""
    sys.path.insert(0, str(HERE.parent))
    from oracle import cpu_port
    from recmv import utils
    from recmv.dataset import getDatasetAndLoader
    from recmv.hocon import ConfigFactory
    from recmv.model.network import getOptNet
    conf = ConfigFactory.parse_file(str(HERE.parent / "configs" / "synthetic" / "people_snapshot_like.conf"))
    conf.put('train.sample_pix_num', 16)
    conds_lens = {'deformer': conf.get_int('mlp_deformer.condlen') * 3, 'renderer': conf.get_int('render_net.condlen')}
    random.seed(2)
    torch.manual_seed(2)
    ds, _ = getDatasetAndLoader(capture, conds_lens, 3, True, 0, True, True, conf.get_config('train.opt_camera'),
                                cf.GARMENT_TYPE, data_type='scene')
    cpu_port.install()
    try:
        res, box = [(9, 11, 7), (17, 21, 13)], ((-0.9, -1.2, -0.6), (0.9, 1.2, 0.6))
        optNet, _ = getOptNet(ds, 'result', 3, box[0], box[1], res, 'cpu', conf, curves=True, skin_grid=(5, 9, 7))
        assert optNet.dataset is ds and optNet.dataset.fl_weights == ds.fl_weights and len(ds.fl_weights) == 6
        assert max(ds.fl_weights.values()) > 1.0                      # the capture's statistic, not the synthetic all-ones
        optNet, _ = utils.set_hierarchical_config(conf, 'coarse', optNet, None, res)
        optimizer = optNet.rebuild_optimizer()
        in_opt = {id(p) for g in optimizer.param_groups for p in g['params']}
        assert id(ds.poses) in in_opt and id(ds.conds[0]) in in_opt and id(ds.camera_params['focal_length']) in in_opt
        assert id(ds.camera_params['cam2world_coord_quat']) not in in_opt
        frames = [0, 2, 3]                                             # frames with a normal map
        datas = torch.utils.data.default_collate([ds[i][1] for i in frames])
        frame_ids = torch.tensor(frames)
        before = ds.poses.detach().clone(), ds.conds[0].detach().clone()
        ratio = {'sdfRatio': 1., 'deformerRatio': 0.5, 'renderRatio': 1.}
        optimizer.zero_grad()
        loss = optNet(datas, 16, ratio, frame_ids, '/tmp/debug', global_optimizer=optimizer)
        loss.backward()
        optNet.propagateTmpPsGrad(frame_ids, ratio)
        optimizer.step()
        assert torch.isfinite(loss)
        moved = (ds.poses.detach() - before[0]).abs().amax(dim=(1, 2))
        assert (moved[frames] > 0).all()      # (their window neighbours move too: the DCT smoothness term, :1221-1250)
        assert not torch.equal(ds.conds[0].detach()[frames], before[1][frames])
        assert 'fl_loss' in optNet.info and torch.isfinite(optNet.info['fl_loss']['total'])
    finally:
        cpu_port.uninstall()


def test_capture_loader_deals_an_epoch_over_ranks(capture):
    """train.py's CaptureLoader: one shuffled pass over the capture per epoch, the same permutation on every rank, dealt
    round-robin; a new permutation per epoch; batches carry the sampler's frame ids."""
    import types
    sys.path.insert(0, str(HERE.parent / "rec-mv_amd"))
    import train
    from recmv.dataset import SceneDataset
    ds = SceneDataset(capture, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30)
    ds.img_ns  # (frames 1, 4, 7, 10 lack a normal map: drop the optional key so that any frames collate together)
    orig = ds._sample
    ds._sample = lambda idx: {k: v for k, v in orig(idx).items() if k != 'normal'}
    seen = {}
    for epoch in (0, 1):
        per_rank = []
        for rank in (0, 1):
            loop = types.SimpleNamespace(batch_size=2, world_size=2, rank=rank)
            ids = [int(i) for frame_ids, outs in train.CaptureLoader(ds, loop).set_epoch(epoch) for i in frame_ids]
            assert len(train.CaptureLoader(ds, loop)) == 3
            cl = train.CaptureLoader(ds, loop)          # what the start-up registration reads off a DataLoader
            assert (cl.batch_size, cl.sampler, cl.num_workers) == (2, None, 0)
            assert len(ds.get_init_fl_datasets(cl.batch_size, cl.sampler, cl.num_workers)) == 6
            per_rank.append(ids)
        assert sorted(per_rank[0] + per_rank[1]) == list(range(cf.FRAMES)) and not set(per_rank[0]) & set(per_rank[1])
        seen[epoch] = per_rank
    assert seen[0] != seen[1]
    # every rank yields the SAME number of batches whatever the frame count (each iteration holds three collectives): the advisor's
    # case F = 649, 8 ranks, batch 3 dealt 28 / 27 iterations before; the Python generator of the caller is left alone
    import random
    for F, world, bs in ((649, 8, 3), (13, 2, 6), (5, 8, 3), (48, 8, 3)):
        order = list(range(F))
        counts, union = set(), []
        for rank in range(world):
            cl = train.CaptureLoader(list(range(F)), types.SimpleNamespace(batch_size=bs, world_size=world, rank=rank))
            batches = cl.rank_batches(order)
            assert all(1 <= len(b) <= bs for b in batches), (F, world, bs, rank)
            counts.add(len(batches))
            assert len(batches) == len(cl)
            union += [i for b in batches for i in b]
        assert len(counts) == 1 and set(union) == set(range(F)), (F, world, bs)
        assert len(union) == F or F % (world * bs) < world            # (only a wrapped last position repeats frames)
    random.seed(5)
    want = random.Random(5).random()
    list(train.CaptureLoader(ds, types.SimpleNamespace(batch_size=2, world_size=1, rank=0)))
    assert random.random() == want


@pytest.mark.parametrize("a_pose", [False, True])
def test_large_pose_dataset_matches_the_reference_class(capture, golden, a_pose):
    """Large_Pose_SceneDataset (:681-892): frozen depth + smoothed translations, shape from the TCMR betas of the A-pose turn,
    TCMR poses after it, samples addressed from `start_idx`."""
    from recmv.dataset import Large_Pose_SceneDataset
    torch.manual_seed(13)
    ds = Large_Pose_SceneDataset(capture, dict(CONDS), cf.GARMENT_TYPE, fl_sampling=30, curve_sampling=1, a_pose=a_pose)
    kind = 'lp%d' % int(a_pose)
    got = cf.collect(ds, kind, samples=(0, 1, len(ds) - 1))
    got[kind + '_apose'] = torch.tensor([float(ds.a_pose_start), float(ds.a_pose_end)])
    got[kind + '_all_trans'], got[kind + '_all_poses'], got[kind + '_shape'] = ds.trans, ds.poses, ds.shape
    _compare(got, golden, kind)


def test_one_euro_smoother_matches_the_reference_function(golden):
    from recmv.dataset import one_euro_smooth
    x = golden['smooth_in']
    assert torch.allclose(one_euro_smooth(x, min_cutoff=0.004, beta=0.7, d_cutoff=1.), golden['smooth_out'], rtol=1e-6, atol=1e-7)
    assert torch.allclose(one_euro_smooth(x), golden['smooth_out_default'], rtol=1e-6, atol=1e-7)
    assert not torch.allclose(golden['smooth_out'][7:, 2], x[7:, 2], atol=0.5)        # the flipped joint was re-expressed
