"""`OptimGarmentNetwork` — the object train.py drives (engineer/networks/OptimGarmentNetwork.py:122-2313 of the
reference), with the reference's call signatures on top of the MI355X hot loop (recmv/loop.py):

    loss = optNet(outs, sample_pix_num, ratio, frame_ids, debug_root, global_optimizer=optimizer)      train.py:324
    loss.backward()
    optNet.propagateTmpPsGrad(frame_ids, ratio)                                                         train.py:327
    optimizer.step()

`forward` takes the mini-batch dict the reference's DataLoader collates (`datas`: img, normal, mask, one segmentation
per garment, fl_pts, fl_masks — dataset/dataset.py:617-680) and reads its ground truth from it; the per-frame learnable
tensors and the camera come from `optNet.dataset`, as in the reference (:1888-1910).  Everything else
(`initializeTmpSDF`, `initializeSDF`, `initializeFL`, `propagateTmpPsGrad`, `discretizeSDF`, `marching_cube_update`, `mask_loss`, `sample_train_ray`, `surface_render_loss`,
`project_2d_loss`, `curve_aware_loss`, `dct_poses_loss`, `opt_times`, `info`, `engine`, ...) is HotLoop's.
"""
import torch

from ...loop import HotLoop


class OptimGarmentNetwork(HotLoop):
    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def forward(self, datas, sample_pix=None, ratio=None, frame_ids=None, root=None, **kwargs):
        """:1885-1969.  `sample_pix` overrides train.sample_pix_num for this call (the loss_<stage>.sample_pix_num of
        the fine stage still wins, :998); `root` (debug dump folder) is accepted and unused; `global_optimizer` is the
        caller's Adam (zeroed after the curve branch, :1934)."""
        if frame_ids is None:
            frame_ids = datas['frame_ids']
        if not torch.is_tensor(frame_ids):
            frame_ids = torch.as_tensor(frame_ids)
        frame_ids = frame_ids.long().to(self.device)
        if sample_pix is not None:
            self.sample_pix = int(sample_pix)
        self._datas = datas
        try:
            return HotLoop.forward(self, frame_ids, ratio, global_optimizer=kwargs.get('global_optimizer'))
        finally:
            self._datas = None

    # -- module-like surface train.py touches ---------------------------------------------------------------------
    def train(self, mode=True):
        return self

    def eval(self):
        return self

    def align_fl(self, path=None):
        """train.py:209.  The reference rebuilds its template feature lines from the SMPL garment assets, applies the
        registration stored in `path` (`fl_init/init_trans_matrix.pth`, written by `initializeFL` /
        `engineer.core.fl_optimizer.scale_rigid_optimizer`) and samples the explicit curves from them (:3485-3546).  The
        asset pipeline is outside this package: without templates the curves are drawn on the initial garment surfaces."""
        if not self.curves:
            self.curves = True
            self._init_curves(0)
        return self
