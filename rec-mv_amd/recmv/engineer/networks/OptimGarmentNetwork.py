"""`OptimGarmentNetwork` — the object train.py drives (engineer/networks/OptimGarmentNetwork.py:122-2313 of the
reference), with the reference's call signatures on top of the MI355X hot loop (recmv/loop.py):

    loss = optNet(outs, sample_pix_num, ratio, frame_ids, debug_root, global_optimizer=optimizer)      train.py:324
    loss.backward()
    optNet.propagateTmpPsGrad(frame_ids, ratio)                                                         train.py:327
    optimizer.step()

`forward` takes the mini-batch dict the reference's DataLoader collates (`datas`: img, normal, mask, one segmentation
per garment, fl_pts, fl_masks — dataset/dataset.py:617-680) and reads its ground truth from it; the per-frame learnable
tensors and the camera come from `optNet.dataset`, as in the reference (:1888-1910).  Everything else
(`initializeTmpSDF`, `initializeSDF`, `initializeFL`, `propagateTmpPsGrad`, `discretizeSDF`, `marching_cube_update`,
`mask_loss`, `sample_train_ray`, `surface_render_loss`,
`project_2d_loss`, `curve_aware_loss`, `dct_poses_loss`, `opt_times`, `info`, `engine`, ...) is HotLoop's.
"""
import torch

from ...loop import HotLoop


def flatten_info(meta, parents=''):
    """`make_recursive_meta_func` of the reference (utils/common_utils.py:73-84): nested dicts become 'a/b' keys, the first three
    entries of a list or tuple '000' .. '002'."""
    out = {}
    if isinstance(meta, dict):
        for k, v in meta.items():
            out.update(flatten_info(v, parents + str(k) + '/'))
    elif isinstance(meta, (list, tuple)):
        for i, v in enumerate(meta[:3]):
            out.update(flatten_info(v, parents + "{:03d}".format(i) + '/'))
    else:
        return {parents[:-1]: meta}
    return out


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

    def draw_loss(self, steps, **kwargs):
        """:3309-3316 — everything in `self.info` (per-garment losses, ray counts, curve terms) plus the caller's scalars to the
        visualizer, flattened.  Reads the 0-d device tensors of `info` back (one sync): train.py calls it every tenth
        iteration, the reference every iteration."""
        if getattr(self, 'visualizer', None) is None:
            return
        info = dict(self.info)
        info.update(kwargs)
        scalars = {}
        for k, v in flatten_info(info).items():
            if torch.is_tensor(v):
                if v.numel() != 1:
                    continue
                v = v.item()
            if isinstance(v, (bool, int, float)):
                scalars[k] = float(v)
        self.visualizer.add_scalar(scalars, int(steps))

    def align_fl(self, fl_align_path=None, epoch=0, fl_templates=None, sample_num=200):
        """train.py:209, OptimGarmentNetwork.py:3485-3546.  With template feature lines (`fl_templates` or
        `self.garment_fl_templates`: {line name: ribbon mesh}) and the registration `fl_align_path`
        (`fl_init/init_trans_matrix.pth`, written by `initializeFL`): apply the stored scale / translation / rotation to every
        line, remember how to take a registered line back onto the canonical body (`cano_fl_to_body_trans`), and sample the
        explicit curves the loop optimises from the longer boundary of every registered ribbon (`inter_free_curve`).
        The reference rebuilds the templates from its SMPL garment assets here (mesh tools outside this package); without
        templates the curves are rings drawn on the initial garment surfaces."""
        import os
        import numpy as np
        from ... import curves as fl
        from ...model import Inverse_Fl_Body
        from ...utils.constant import FL_INFOS
        from ..utils.matrix_transform import FeatureLineMesh, scale_icp_rotate_center_transform
        from ..utils.polygons import uniformsample3d
        templates = fl_templates if fl_templates is not None else getattr(self, 'garment_fl_templates', None)
        if templates is None or fl_align_path is None or not os.path.isfile(fl_align_path):
            if not self.curves:
                self.curves = True
                self._init_curves(0)
            return self
        dev = self.device
        names = FL_INFOS[self.garment_type]
        lines = [templates[n].to(dev) for n in names]
        stored = torch.load(fl_align_path)
        if 'rigid_scale' not in stored:
            raise NotImplementedError
        rigid_R, rigid_T, rigid_scale = (stored[k].to(dev) for k in ('rigid_R', 'rigid_T', 'rigid_scale'))
        self.fl_names = list(names)
        self.cano_fl_to_body_trans = Inverse_Fl_Body(lines, self.fl_names, rigid_T, rigid_scale)
        moved = scale_icp_rotate_center_transform(lines, rigid_R, rigid_T, rigid_scale)
        self.cano_fl_to_body_trans.set_rigid_center([v.mean(0, keepdim=True) for v in moved], self.fl_names)
        self.fl_meshes = [FeatureLineMesh(v, m.faces_packed()) for v, m in zip(moved, lines)]
        curves = []
        for mesh in self.fl_meshes:
            loop = fl.longest_boundary_loop(mesh.faces_packed())
            pts = uniformsample3d(mesh.verts_packed()[loop].detach().cpu().numpy(), sample_num)
            curves.append(torch.from_numpy(np.ascontiguousarray(pts)).float().to(dev))
        shortest = min(c.shape[0] for c in curves)     # (sample_num or sample_num - 1 points per line, see uniformsample3d; the
        curves = [c[:shortest] for c in curves]        #  reference stacks them as they come and needs them equal)
        # their counterparts on the canonical body: the sampled curves with translation and scale undone
        # (Intersect_Free_Curve.initialize_parameters, engineer/utils/garment_structure.py:77)
        smpl_curves = self.cano_fl_to_body_trans(curves, self.fl_names)
        self.fl_extract, _ = self._feature_line_tables(available=self.fl_names)
        self.inter_free_curve = fl.Intersect_Free_Curve(curves, smpl_curves, self.fl_names).to(dev)
        self._ensure_body_template()
        self.curves = True
        if getattr(self, 'garment_vs', None):
            self.fl_optimizer = torch.optim.AdamW(self.inter_free_curve.parameters(), lr=1e-4)
        return self
