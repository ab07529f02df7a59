"""train.py — the reference's training driver (train.py:21-356) on the MI355X hot loop.

Same command line (`--gpu-ids --conf --data --model-rm-prefix --sdf-model --save-folder --project_name --exp_name
--data_type --a_pose --curve_sampling --resume`), same HOCON schema, same entry points and the same loop body:

    optNet, sdf_initialized = getOptNet(dataset, save_folder, batch_size, bmins, bmaxs, resolutions['coarse'], device, config)
    optNet, dataloader = utils.set_hierarchical_config(config, 'coarse', optNet, dataloader, resolutions['coarse'])
    ...
    loss = optNet(outs, sample_pix_num, ratio, frame_ids, debug_root, global_optimizer=optimizer)
    loss.backward()
    optNet.propagateTmpPsGrad(frame_ids, ratio)
    optimizer.step()

coarse -> medium -> fine stages switched at `train.<stage>.start_epoch`, Adam + MultiStepLR, `coarse.pth` / `medium.pth`
at the stage switches and `latest.pth` every epoch in the reference's checkpoint layout (recmv.utils.save_model /
load_model), resume with the scheduler fast-forwarded and `opt_times` recomputed (train.py:232-260).
`train_large_pose.py` is the same driver on the large-pose variant (SDF nets frozen, resume from `a-pose.pth`).

What differs: the reference cuts its garment templates (point clouds for the SDF pre-fit, feature-line ribbons) out of SMPL
garment assets with mesh tools that are outside this tier.  The pre-fit itself is here: with
`--init-points <npz>` (body / garment point clouds) and no `initial_sdf_idr_*.pth` under the save folder yet, the nets are
fitted for `train.initial_iters` epochs (`initializeTmpSDF`, train.py:179-206) and the files written under the reference's
names; a later run loads them (getOptNet).  Without the clouds the canonical surfaces start from the geometric
initialisation.  `--fl-templates <npz>` (one ribbon mesh per feature line of the capture) registers the lines to the annotated
frames and samples the loop's curves from them (`initializeFL`, `align_fl`); without it the curves are rings on the initial
surfaces.  Scalars go to wandb when it is installed, to
`<save-folder>/logs/<exp_name>.jsonl` otherwise.  `--data <capture> --data_type scene|people_snap|large_pose`
reads a capture directory in the reference's layout through `recmv.dataset` (images, masks, garment regions, 2-D feature
lines, SMPL poses, camera); without a capture the frames are synthetic (`recmv.loop.SyntheticFrames`; `--frames` sets
their number) and `--data` is only the root under which `--save-folder` is created.  One process per GPU: under
`python -m torch.distributed.run --nproc-per-node N train.py ...` the frames of every mini-batch are sharded over the
ranks and the shared gradients all-reduced with RCCL (recmv.dist); `--gpu-ids` picks the device of a single process.
"""
from __future__ import annotations

import argparse
import os
import os.path as osp
import sys
import time

sys.path.insert(0, osp.dirname(osp.abspath(__file__)))


def build_parser(large_pose=False):
    parser = argparse.ArgumentParser(description='neu video body rec')
    parser.add_argument('--gpu-ids', nargs='+', type=int, metavar='IDs', default=[0], help='gpu ids')
    parser.add_argument('--conf', default=None, metavar='M', help='config file')
    parser.add_argument('--data', default=None, metavar='M', help='data root')
    parser.add_argument('--model-rm-prefix', nargs='+', type=str, metavar='rm prefix', help='rm model prefix')
    parser.add_argument('--sdf-model', default=None, metavar='M', help='substitute sdf model')
    parser.add_argument('--save-folder', default=None, metavar='M', help='save folder')
    parser.add_argument('--project_name', type=str, default='recmv', help='exp name show by wandb')
    parser.add_argument('--exp_name', type=str, default='run', help='exp name show by wandb')
    parser.add_argument('--data_type', type=str, default='synthetic', help='the type of dataset')
    parser.add_argument('--curve_sampling', type=int, default=1, help='the type of dataset')
    parser.add_argument('--garment_type', type=str, default=None,
                        help='capture name in utils/constant.py FL_INFOS (default: the basename of --data)')
    if not large_pose:                           # train_large_pose.py:20-37 has neither flag: it resumes from a-pose.pth
        parser.add_argument('--a_pose', action='store_true', help='the type of dataset')
        parser.add_argument('--resume', default=None, metavar='M', help='pretrained scene model')
    # extensions
    parser.add_argument('--no-curves', action='store_true',
                        help='skip the feature-curve branch (project_2d_loss) the reference runs every iteration')
    parser.add_argument('--frames', type=int, default=64, help='number of synthetic frames')
    parser.add_argument('--fl-templates', default=None, metavar='NPZ',
                        help='template feature lines as ribbon meshes: <line>_verts [V,3], <line>_faces [F,3] per line of the '
                             'capture (FL_INFOS); registered to the annotated frames before the loop (initializeFL / align_fl)')
    parser.add_argument('--init-points', default=None, metavar='NPZ',
                        help='oriented point clouds for the SDF pre-fit: body_vs, body_ns, <garment>_vs, <garment>_ns '
                             '(what the reference cuts out of its SMPL garment templates)')
    parser.add_argument('--max-iters', type=int, default=-1, help='stop after this many optimiser iterations (smoke runs)')
    return parser


class CaptureLoader:
    """`DataLoader(dataset, batch_size, sampler=RandomSampler(dataset, 1, shuffle))` of the reference (dataset/dataset.py:
    1179-1182) rebuilt per epoch with the current stage's batch size (utils/utils.py:342-346 rebuilds it at every stage
    switch), the epoch's permutation seeded so that frame-sharded ranks deal it round-robin among themselves."""

    def __init__(self, dataset, loop, epoch=0):
        self.dataset, self.loop, self.epoch = dataset, loop, epoch

    def set_epoch(self, epoch):
        self.epoch = epoch
        return self

    # what the start-up registration reads off the reference's DataLoader (engineer/core/fl_optimizer.py:121)
    sampler, num_workers = None, 0

    @property
    def batch_size(self):
        return self.loop.batch_size

    def __len__(self):
        from recmv.loop import iters_per_epoch
        return iters_per_epoch(len(self.dataset), self.loop.batch_size, self.loop.world_size)

    def rank_batches(self, order, rank=None):
        """This rank's frame indices per position of the epoch.  The permutation is cut into len(self) positions of
        batch_size * world_size frames and every rank takes its round-robin share of EACH position (HotLoop.frame_batch_at), so
        all ranks yield exactly len(self) batches — every iteration holds three collectives, a rank that ran one batch more or
        less than its peers would hang them or pair them with the next epoch's.  The last position holds the remaining frames;
        when those are fewer than the ranks the permutation wraps so that nobody is left without a frame."""
        rank = self.loop.rank if rank is None else rank
        world, bs = self.loop.world_size, self.loop.batch_size
        per_it = bs * world
        out = []
        for pos in range(len(self)):
            ids = order[pos * per_it:(pos + 1) * per_it]
            if len(ids) < world:
                ids = ids + order[:world - len(ids)]
            out.append(ids[rank::world][:bs])
        return out

    def __iter__(self):
        import random

        import torch
        from recmv.dataset import RandomSampler
        py_state, state = random.getstate(), torch.random.get_rng_state()
        random.seed(1234 + self.epoch)
        torch.manual_seed(1234 + self.epoch)
        order = list(iter(RandomSampler(self.dataset, 1, True)))
        torch.random.set_rng_state(state)
        random.setstate(py_state)                    # (the caller's Python generator is left where it was)
        batches = self.rank_batches(order)
        loader = torch.utils.data.DataLoader(self.dataset, batch_sampler=batches, num_workers=0)
        return iter(loader)


def load_fl_templates(path, names, device='cpu'):
    """{line name: FeatureLineMesh} from an npz with `<line>_verts` / `<line>_faces` for every line in `names`."""
    import numpy as np
    import torch
    from recmv.engineer.utils.matrix_transform import FeatureLineMesh
    data = np.load(path)
    missing = [n for n in names if n + '_verts' not in data or n + '_faces' not in data]
    if missing:
        raise KeyError('%s: no template for the feature line(s) %s' % (path, ', '.join(missing)))
    return {n: FeatureLineMesh(torch.from_numpy(data[n + '_verts']).float().to(device),
                               torch.from_numpy(data[n + '_faces']).long().to(device)) for n in names}


def register_feature_lines(optNet, dataloader, templates, save_root):
    """train.py:209 with what `initializeTmpSDF` does first (OptimGarmentNetwork.py:542): register the template lines to the
    annotated frames (writes fl_init/init_trans_matrix.pth; a stored file is re-applied, not re-fitted) and turn them into the
    loop's explicit curves."""
    optNet.garment_fl_templates = templates
    optNet.initializeFL(dataloader, 0, optNet.device, osp.join(save_root, 'initial_sdf.pth'))
    optNet.align_fl(osp.join(save_root, 'fl_init', 'init_trans_matrix.pth'), fl_templates=templates)


def prefit_sdf(optNet, nepochs, config, args, save_root, rank):
    """train.py:179-206 — no `initial_sdf_idr_*.pth` yet: fit the body net and the garment nets to the template point clouds
    (`optNet.initializeTmpSDF`), write the state dicts under the reference's names, extract the fitted body at the coarse
    resolution and keep it (`load_init_sdf_vertices`, `initial_sdf_idr_*.ply`).  The clouds come from `--init-points`; without
    them the nets keep their geometric initialisation (the reference builds the clouds from its SMPL garment assets)."""
    import numpy as np
    import torch
    from recmv import utils
    if args.init_points is None:
        if rank == 0:
            print('no initial_sdf_idr_*.pth under %s and no --init-points: the SDF nets start from the geometric initialisation' % save_root)
        return
    pts = np.load(args.init_points)
    cloud = lambda key: (torch.from_numpy(pts[key + '_vs']).float(),
                         torch.from_numpy(pts[key + '_ns']).float() if key + '_ns' in pts else None)
    stem = 'initial_sdf_idr_%d_%d' % (config.get_int('sdf_net.multires'),
                                      config.get_int('train.skinner_pose_type') if 'train.skinner_pose_type' in config else 0)
    optNet.initializeTmpSDF(nepochs, osp.join(save_root, stem + '.pth'), True, body_points=cloud('body'),
                            garment_points=[cloud(name) for name in optNet.garment_names], log=print if rank == 0 else None)
    verts_list, faces_list = optNet.discretizeSDF(-1, None)
    optNet.load_init_sdf_vertices(verts_list[0], faces_list[0])
    if rank == 0:
        utils.write_ply(osp.join(save_root, stem + '.ply'), verts_list[0], faces_list[0])
        for name, v, f in zip(optNet.garment_names, verts_list[1:], faces_list[1:]):
            utils.write_ply(osp.join(save_root, stem.replace('sdf', 'sdf_' + name) + '.ply'), v, f)


def stage_of_epoch(config, epoch):
    """'coarse' | 'medium' | 'fine' for an epoch (train.py:233-244, :300-314)."""
    fine, medium = config.get_int('train.fine.start_epoch'), config.get_int('train.medium.start_epoch')
    if fine >= 0 and epoch >= fine:
        return 'fine'
    if medium >= 0 and epoch >= medium:
        return 'medium'
    return 'coarse'


def resumed_opt_times(config, n_frames, start_epoch, world_size=1):
    """Optimiser iterations already done when resuming after `start_epoch` (train.py:250-260, formulas kept; the
    per-epoch count is the loop's own `iters_per_epoch`, i.e. ceil(F / (batch * ranks)) — the reference's formula at
    one rank)."""
    from recmv.loop import iters_per_epoch
    coarse_epoch = config.get_int('train.coarse.start_epoch')
    medium_epoch = config.get_int('train.medium.start_epoch')
    fine_epoch = config.get_int('train.fine.start_epoch')
    bs = {s: config.get_int(f'train.{s}.point_render.batch_size') for s in ('coarse', 'medium', 'fine')}
    coarse_time = iters_per_epoch(n_frames, bs['coarse'], world_size) * (medium_epoch - coarse_epoch)
    medium_time = iters_per_epoch(n_frames, bs['medium'], world_size) * (fine_epoch - medium_epoch)
    fine_time = iters_per_epoch(n_frames, bs['fine'], world_size) * (start_epoch - medium_epoch + 1)
    return float(coarse_time + medium_time + fine_time)


def main(argv=None, large_pose=False):
    args = build_parser(large_pose).parse_args(argv)
    import torch
    from recmv import dist as rdist, utils
    from recmv.hocon import ConfigFactory
    from recmv.loop import RESOLUTIONS, FrameLoader
    from recmv.model.network import getOptNet

    torch.set_num_threads(min(8, os.cpu_count() or 1))     # host side only launches kernels (see bench.py)
    config = ConfigFactory.parse_file(args.conf)
    rank, local_rank, world = rdist.init_distributed()
    assert torch.cuda.is_available(), "train.py needs a GPU (librecmv_hip.so has no CPU fallback)"
    device = torch.device('cuda', local_rank if world > 1 else (args.gpu_ids[0] if args.gpu_ids else 0))
    torch.cuda.set_device(device)
    if os.environ.get("RECMV_ALLOC_CONF", "roundup_power2_divisions:4"):
        # a re-mesh changes every vertex-sized shape by a percent or two: allocation sizes rounded up to a quarter of a power of two
        # find the buffers the previous mesh left in torch's cache instead of going to hipMalloc (bench.py does the same; "" disables)
        torch.cuda.memory._set_allocator_settings(os.environ.get("RECMV_ALLOC_CONF", "roundup_power2_divisions:4"))
    if args.save_folder is None:
        print('please set save-folder...')
        assert (False)
    save_root = osp.join(args.data or '.', args.save_folder)
    debug_root = osp.join(save_root, 'debug')
    if rank == 0:
        os.makedirs(debug_root, exist_ok=True)
    resolutions = RESOLUTIONS                                          # train.py:42-79
    batch_size = config.get_int('train.coarse.point_render.batch_size')
    sample_pix_num = config.get_int('train.sample_pix_num')

    # train.py:150-160: a capture directory (`--data` with imgs/ masks/ ... , `--data_type scene | people_snap | large_pose | synthe`) is read by
    # recmv.dataset with the reference's conds_lens; without one the frames are synthetic
    capture = None
    if args.data is not None and args.data_type in ('scene', 'people_snap', 'large_pose', 'synthe') and osp.isdir(osp.join(args.data, 'imgs')):
        from recmv.dataset import getDatasetAndLoader
        # the capture's name keys the garment set (train.py:107,115: `train.garment_type` of the config; --garment_type or the
        # folder name stand in when a config leaves it out); one deformer code for the body + one per garment template
        from recmv.utils.constant import TEMPLATE_GARMENT
        garment_type = args.garment_type or (config.get_string('train.garment_type') if 'train.garment_type' in config
                                             else osp.basename(osp.normpath(args.data)))
        if garment_type not in TEMPLATE_GARMENT:
            raise SystemExit("unknown capture %r: train.garment_type / --garment_type / the data folder's name must be one of %s "
                             "(recmv/utils/constant.py TEMPLATE_GARMENT)" % (garment_type, sorted(TEMPLATE_GARMENT)))
        config.put('train.garment_type', garment_type)
        conds_lens = {'deformer': config.get_int('mlp_deformer.condlen') * (1 + len(TEMPLATE_GARMENT[garment_type])),
                      'renderer': config.get_int('render_net.condlen')}
        capture, _ = getDatasetAndLoader(args.data, conds_lens, batch_size, True, 0, config.get_bool('train.opt_pose'),
                                         config.get_bool('train.opt_trans'), config.get_config('train.opt_camera'),
                                         garment_type, data_type=args.data_type, curve_sampling=args.curve_sampling,
                                         a_pose=bool(getattr(args, 'a_pose', False)))
        for t in capture.conds + [capture.poses, capture.trans, capture.shape] + list(capture.camera_params.values()):
            t.data = t.data.to(device)            # the reference keeps these on the host and moves batches per call
    # The start-up stage (first-run skinner bake, SDF pre-fit, feature-line registration) draws unsynchronised random numbers and
    # leaves files other ranks would read half-written: rank 0 runs it FIRST, the other ranks wait at a barrier and then find the
    # stored files (initial_skinner_*.pth, initial_sdf_*.pth, fl_init/init_trans_matrix.pth) like any later run does.
    staged = world > 1 and capture is not None
    if staged and rank != 0:
        rdist.startup_gate(True)               # leaves with rank 0's outcome: SystemExit on every rank when its start-up raised
    try:
        # train.py:170-171 (bmins / bmaxs None: the canonical box is sized from the initial surfaces)
        optNet, sdf_initialized = getOptNet(capture, args.save_folder, batch_size, None, None, resolutions['coarse'], device,
                                            config, opt_large=large_pose, n_frames=args.frames, H=512, W=512,
                                            world_size=world, rank=rank, curves=not args.no_curves)
        dataset = optNet.dataset
        dataloader = FrameLoader(optNet) if capture is None else CaptureLoader(capture, optNet)
        if sdf_initialized > 0:
            prefit_sdf(optNet, sdf_initialized, config, args, save_root, rank)
        if args.fl_templates is not None and capture is not None and not args.no_curves:
            from recmv.utils.constant import FL_INFOS
            register_feature_lines(optNet, dataloader, load_fl_templates(args.fl_templates, FL_INFOS[optNet.garment_type], device),
                                   save_root)
    except BaseException:
        if staged:
            import traceback
            traceback.print_exc()
            if rank == 0:
                rdist.startup_gate(False)      # releases the waiting ranks with the failure instead of the collective timeout
            else:
                rdist.startup_gate(False, "start-up stage (rank %d)" % rank)      # second gate: this rank's own failure, see below
        raise
    if staged and rank == 0:
        rdist.startup_gate(True)
    if staged:
        # second gate, every rank reporting its OWN outcome: a non-zero rank that failed in its getOptNet after the first gate (a
        # half-written file, a missing capture on its node) takes the job down here instead of at the first collective's timeout
        rdist.startup_gate(True, "start-up stage (every rank)")
    if rank == 0:                                 # train.py:86: wandb when it is there, a jsonl file under logs/ otherwise
        from recmv.engineer.visualizer import wandb_visualizer
        optNet.visualizer = wandb_visualizer(args.project_name, args.exp_name, resume=False, log_dir=osp.join(save_root, 'logs'))
    optNet, dataloader = utils.set_hierarchical_config(config, 'coarse', optNet, dataloader, resolutions['coarse'])
    rdist.broadcast_state([p for p in optNet.shared_parameters()] + list(optNet.sdf.parameters())
                          + (list(optNet.inter_free_curve.parameters()) + list(optNet.inter_free_curve.buffers())
                             if optNet.curves else []))
    allreduce = rdist.GradAllReduce(world) if world > 1 else None
    optNet._allreduce = allreduce                 # the explicit-vertex and curve gradients are shared inside forward
    optNet.train()
    optNet.opt_times = 0.
    start_epoch = 0
    milestones = config.get_list('train.scheduler.milestones')
    gamma = config.get_float('train.scheduler.factor')
    optimizer = optNet.rebuild_optimizer()                                               # train.py:213
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma=gamma)
    ratio = {'sdfRatio': None, 'deformerRatio': None, 'renderRatio': None}
    stage = 'coarse'

    resume = osp.join(args.data or '.', args.save_folder, 'a-pose.pth') if large_pose else args.resume   # train_large_pose.py:39
    if resume is not None and osp.isfile(resume):
        print('load model: ' + resume)
        optNet, dataset, start_epoch = utils.load_model(resume, optNet, dataset, device, args.sdf_model,
                                                        args.model_rm_prefix)
        if large_pose:
            start_epoch = 60                                                             # train_large_pose.py:210
        stage = stage_of_epoch(config, start_epoch)
        if stage != 'coarse':
            optNet, dataloader = utils.set_hierarchical_config(config, stage, optNet, dataloader, resolutions[stage])
            optNet.isfine = stage == 'fine'
            print('enable %s hierarchical' % stage)
        optimizer = optNet.rebuild_optimizer()
        scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma=gamma)
        for __ in range(start_epoch + 1):
            scheduler.step()
        optNet.opt_times += resumed_opt_times(config, len(dataset), start_epoch, world)
        start_epoch += 1

    nepochs = config.get_int('train.nepoch') + (1 if large_pose else 0)                  # train_large_pose.py:289
    done = 0
    optNet.reserve_memory()              # one large cached block per stream: a re-mesh's re-sized buffers never reach hipMalloc
    for epoch in range(start_epoch, nepochs):
        new_stage = stage_of_epoch(config, epoch)
        if new_stage != stage:
            if rank == 0:
                utils.save_model(osp.join(save_root, stage + ".pth"), epoch, optNet, dataset)   # coarse.pth / medium.pth
            optNet, dataloader = utils.set_hierarchical_config(config, new_stage, optNet, dataloader,
                                                               resolutions[new_stage])
            optNet.isfine = new_stage == 'fine'                  # train.py:312
            stage = new_stage
            torch.cuda.empty_cache()
            optNet.reserve_memory()
            print('enable %s hierarchical' % stage)
        for data_index, (frame_ids, outs) in enumerate(dataloader.set_epoch(epoch)):
            t0 = time.perf_counter()
            frame_ids = frame_ids.long().to(device)
            optimizer.zero_grad()
            ratio['sdfRatio'] = 1.
            ratio['deformerRatio'] = optNet.opt_times / 2500. + 0.5
            ratio['renderRatio'] = 1.
            loss = optNet(outs, sample_pix_num, ratio, frame_ids, debug_root, global_optimizer=optimizer)
            optNet.backward(loss)           # train.py:325 `loss.backward()`, started from the stream the open terms live on (HotLoop.backward)
            optNet.propagateTmpPsGrad(frame_ids, ratio)
            if allreduce is not None:
                allreduce(optNet.shared_parameters())            # frame-sharded ranks: mean of the shared gradients
            optimizer.step()
            if rank == 0:
                lr = optimizer.param_groups[0]['lr']
                info = optNet.info
                msg = '(%d/%d) loss = %.5f lr = %.2e rays = %d' % (epoch, data_index, float(loss), lr,
                                                                   int(info.get('rays_total', 0)))
                for name in optNet.garment_names:
                    msg += ' | %s: eik %.4f pc_sdf %.5f' % (name, float(info.get(name + '_grad_loss', 0.)),
                                                           float(info.get('pc_%s_loss_sdf' % name, 0.)))
                print(msg + ' (%.0f ms)' % ((time.perf_counter() - t0) * 1e3), flush=True)
                if done % 10 == 0:                                   # train.py:330 (every iteration there)
                    optNet.draw_loss(optNet.opt_times, total_loss=float(loss), learning_rate=lr, ratio=ratio)
            optNet.opt_times += 1.
            done += 1
            if 0 <= args.max_iters <= done:
                break
        if rank == 0:
            utils.save_model(osp.join(save_root, "latest.pth"), epoch, optNet, dataset)
        scheduler.step()
        if 0 <= args.max_iters <= done:
            break
    rdist.barrier()


if __name__ == '__main__':
    main()
