"""engineer/core/beta_optimizer.py of the reference: `smpl_beta_optimizer` (:132-245) — once, before the loop, fit the SMPL
shape (betas) and one translation offset shared by all frames so that the posed model's joints project onto the capture's 2-D
joints (`gt_joints2d` of the dataset samples: 17 COCO joints with a visibility flag): 150 // (frames / 8) epochs of Adam (5e-3) on
the visibility-weighted L1 distance in pixels.  getOptNet runs it when there is no `initial_skinner_<pose type>.pth` yet
(model/network.py:254-262) and stores the result in that file.

The SMPL model itself (`smpl_pytorch`, un-vendored in the reference, model files not redistributable) is an input here: `smpl=`
or `recmv.model.Deformer.getSMPL`.  Pinned against the reference function on a stand-in model (tests/golden/make_golden_startup.py).
"""
import torch

COCOPLUS2COCO = [14, 15, 16, 17, 18, 9, 8, 10, 7, 11, 6, 3, 2, 4, 1, 5, 0]          # :65-67: SMPL's 19 "cocoplus" joints -> COCO order


def batch_kp_2d_l1_loss(real_2d_kp, predict_2d_kp):
    """:69-79 — sum over joints of visibility x (|dx| + |dy|), divided by twice the number of visible joints."""
    kp_gt = real_2d_kp.view(-1, 3)
    kp_pred = predict_2d_kp.contiguous().view(-1, 2)
    vis = kp_gt[:, 2]
    k = torch.sum(vis) * 2.0 + 1e-8
    dif_abs = torch.abs(kp_gt[:, :2] - kp_pred).sum(1)
    return torch.matmul(dif_abs, vis) * 1.0 / k


def smpl_beta_optimizer(gender, initPose, dataset, device='cuda:0', smpl=None, log=print):
    """:132-245.  Returns (betas [10], extra_trans [1,3]), detached.  `initPose` is accepted and unused, as in the reference."""
    from ...model import RectifiedPerspectiveCameras
    from ...model.Deformer import getSMPL
    smpl = (smpl if smpl is not None else getSMPL(gender)).to(device)
    betas = dataset.shape.to(device).clone().requires_grad_(True)
    extra_trans = torch.zeros(1, 3, device=device, requires_grad=True)
    optimizer = torch.optim.Adam([betas, extra_trans], lr=0.005, weight_decay=0.)
    # the reference passes its RandomSampler as DataLoader's third POSITIONAL argument (:150), which is `shuffle`: the loader
    # shuffles with torch's own sampler and the custom one is never iterated — the frame order below is that one
    loader = torch.utils.data.DataLoader(dataset, 8, shuffle=True, num_workers=0)
    step = 0
    for epoch in range(150 // len(loader)):
        for frame_ids, batch in loader:
            gt_joints2d = batch['gt_joints2d'].to(device)
            params = dataset.get_grad_parameters(frame_ids, device)
            poses, trans = params[0].detach(), params[1].detach() + extra_trans
            n = poses.shape[0]
            focals, pps, Rs, Ts, H, W = dataset.get_camera_parameters(frame_ids.numel(), device)
            img_size = torch.tensor([float(W), float(H)], device=device).view(1, 2).expand(n, 2)
            cameras = RectifiedPerspectiveCameras(focals, pps, Rs, Ts, image_size=[(W, H)]).to(device)
            verts, _, _ = smpl(betas[None].expand(n, -1), poses, True)
            verts = verts + trans.view(-1, 1, 3)
            joints = torch.stack([torch.matmul(verts[:, :, i], smpl.joint_regressor) for i in range(3)], dim=2)
            screen = cameras.transform_points_screen(joints, img_size)[:, COCOPLUS2COCO, :]
            optimizer.zero_grad()
            loss = batch_kp_2d_l1_loss(gt_joints2d, screen[..., :2])
            if log is not None:
                log("iteration step {:04d}: {:.4f}".format(step, loss.item()))
            loss.backward()
            optimizer.step()
            step += 1
    return betas.detach().clone(), extra_trans.detach().clone()
