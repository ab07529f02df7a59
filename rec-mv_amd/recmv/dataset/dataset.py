"""The reference's capture directories as a torch Dataset (dataset/dataset.py of the reference).

Layout of a capture (`read_data`, dataset.py:181-239):

    imgs/<i>.jpg|png            frames, numbered from 0 without gaps
    masks/<i>.png               foreground masks (any channel > 0)
    parsing_SCH_ATR/<i>.npy     ATR human-parsing labels [H,W]; `mask_parsing_<i>.npy` beside them = the labels spread over the
                                whole mask by nearest labelled pixel (written by `parsing_mask`, read by `__getitem__`)
    normals/<i>.png             optional normal maps
    featurelines/*.json         labelme polylines of the 2-D feature lines, named by frame number (PeopleSnapshot: mask2fl/)
    smpl_rec.npz                poses [F,72], trans [F,3], shape [10], gender, optional vid_seg_indices
    camera.npz                  fx fy cx cy quat T
    <garment_type>_tcmr_output.pkl   optional 2-D joints / TCMR poses (joblib)

What a sample is (`__getitem__`, :361-423): `(idx, {'img' [H,W,3] in [-1,1] (B,G,R order — the reference reads with OpenCV and
the colour net is trained on that order), 'mask' [H,W], 'fl_pts' [L*S,2], 'fl_masks' [L], 'normal' [H,W,3] (R,G,B), the garment
region masks 'upper' / 'bottom' / 'upper_bottom' / 'body', 'gt_joints2d'})`; the learnable per-frame tensors live on the dataset
(`poses`, `trans`, `conds`, `camera_params`) and are fetched by frame id (`get_grad_parameters`, :425-433).

Images are decoded with Pillow (no OpenCV in this image) into OpenCV's channel order.  The loaders `scene`, `people_snap` and
`large_pose` are provided; `snug` / `synthe` raise (SNUG motion files and the synthetic-outfit renders are outside this tier)."""
import os
import os.path as osp
import random
from glob import glob

import numpy as np
import torch

from ..engineer.utils.featureline_utils import check_feature_lines, obtain_feature_lines
from ..engineer.utils.polygons import uniformsample
from ..utils.constant import ATR_PARSING, FL_INFOS

__all__ = ["SceneDataset", "Synthe_SceneDataset", "Init_Fl_SceneDataset", "People_Snapshot_SceneDataset", "Large_Pose_SceneDataset", "one_euro_smooth", "ClipSampler", "RandomSampler", "getDatasetAndLoader",
           "read_image_bgr", "dct_space"]


def read_image_bgr(path):
    """uint8 [H,W,3] in B,G,R order, what `cv2.imread(path)` returns (alpha dropped, grey replicated)."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def dct_space(k, n):
    """First k rows of the orthonormal DCT-II basis of length n (utils/utils.py:293-304: DCTBasis / DCTSpace)."""
    idx = torch.arange(n, dtype=torch.float64) + 0.5
    rows = []
    for r in range(k):
        assert r < n
        scale = 1. / np.sqrt(float(n)) if r == 0 else np.sqrt(2. / float(n))
        rows.append((torch.cos((np.pi * idx * r / float(n)).float()) * scale))
    return torch.stack(rows)


def _frame_number(path):
    return int(osp.basename(path).split('.')[0])


class SceneDataset(torch.utils.data.Dataset):
    FEATURE_LINE_DIR = 'featurelines'

    def __init__(self, data_root, conds_lens={}, garment_type="", fl_sampling=100, curve_sampling=1):
        assert not garment_type == ""
        self.root = data_root
        self.garment_type = garment_type
        self.fl_names = FL_INFOS[garment_type]
        self.fl_sampling = fl_sampling
        self.curve_sampling = curve_sampling
        self.conds_lens = conds_lens
        self.require_albedo = False
        self.read_data()
        self._read_joints(osp.join(data_root, '{}_tcmr_output.pkl'.format(garment_type)))
        # per-frame codes: smooth in time — random coefficients on the lowest frame_num/5 DCT frequencies (:83-91)
        self.conds, self.cond_ns = [], []
        for name, length in conds_lens.items():
            k = self.frame_num // 5
            cond = (0.1 * torch.randn(length, k)).matmul(dct_space(k, self.frame_num)).transpose(0, 1)
            cond.requires_grad_()
            self.conds.append(cond)
            self.cond_ns.append(name)
        self.area_size_statistic()

    # ---------------------------------------------------------------------------------------------- reading
    def _read_joints(self, path):
        """Optional TCMR output (:50-78): 2-D joints per annotated frame, poses, betas."""
        self.gt_joints2d = None
        if not osp.exists(path):
            return
        import joblib
        out = joblib.load(path)[1]
        self.gt_joints2d_list = out['gt_joints2d']
        self.tcmr_poses, self.tcmr_betas = out['pose'], out['betas']
        self.gt_joints2d = {fid: j for fid, j in zip(out['frame_ids'], self.gt_joints2d_list)}
        self.joints_frame_ids = out['frame_ids'].tolist()

    def read_data(self):
        imgs = []
        for ext in ('.jpg', '.png'):
            imgs.extend(glob(osp.join(self.root, 'imgs/*' + ext)))
        imgs.sort(key=_frame_number)
        self.frame_num = len(imgs)
        self.img_ns = imgs
        self.mask_ns, self.parsing_mask_ns = [], []
        for ind, img_n in enumerate(imgs):
            assert ind == _frame_number(img_n), "frames must be numbered 0..F-1 without gaps"
            stem = osp.basename(img_n).split('.')[0]
            self.mask_ns.append(osp.join(self.root, 'masks/%s.png' % stem))
            self.parsing_mask_ns.append(osp.join(self.root, 'parsing_SCH_ATR/%s.npy' % stem))
            assert osp.isfile(self.mask_ns[-1]) and osp.isfile(self.parsing_mask_ns[-1])
        self.H, self.W, _ = read_image_bgr(self.mask_ns[0]).shape
        smpl = np.load(osp.join(self.root, 'smpl_rec.npz'))

        def f32(key, *shape):
            return torch.from_numpy(smpl[key].astype(np.float32)).view(*shape)

        self.poses, self.trans, self.shape = f32('poses', -1, 24, 3), f32('trans', -1, 3), f32('shape', -1)
        self.gender = str(smpl['gender']) if 'gender' in smpl else 'neutral'
        seg = smpl['vid_seg_indices'] if 'vid_seg_indices' in smpl else []
        self.video_segmented_index = list(np.asarray(seg).tolist()[:-1])          # cut points between concatenated videos
        cam = np.load(osp.join(self.root, 'camera.npz'))
        fl_dir = osp.join(self.root, self.FEATURE_LINE_DIR)
        assert osp.exists(fl_dir)
        self.read_feature_lines(fl_dir)
        self.camera_params = {
            'focal_length': torch.tensor(np.array([cam['fx'], cam['fy']]).astype(np.float32)),
            'princeple_points': torch.tensor(np.array([cam['cx'], cam['cy']]).astype(np.float32)),
            'cam2world_coord_quat': torch.from_numpy(cam['quat'].astype(np.float32)).view(-1),
            'world2cam_coord_trans': torch.from_numpy(cam['T'].astype(np.float32)).view(-1)}

    def _assign_feature_line_files(self, path):
        """Frame -> annotation file: its own if there is one, else the last one before it (:157-179); also which frames
        carry an annotation of their own."""
        files = sorted(glob(osp.join(path, '*.json')))
        per_frame, own, k = [], [], 0
        for frame_id in range(self.frame_num):
            name = _frame_number(files[k]) if k < len(files) else _frame_number(files[-1])
            if frame_id == name:
                per_frame.append(files[k])
                own.append(True)
                k += 1
            else:
                per_frame.append(files[k - 1])
                own.append(False)
        for f in files:
            check_feature_lines(f)
        return files, per_frame, own

    def read_feature_lines(self, path):
        _, self.fl_paths, _ = self._assign_feature_line_files(path)
        self.fl_supervised = [True for _ in self.fl_paths]          # every frame inherits an annotation (:176)

    # ---------------------------------------------------------------------------------------------- feature lines
    def obtain_fl_pts(self, fls):
        """Per feature line of the capture: `fl_sampling` points along the annotated polyline (opened at its longest gap when
        the annotation wraps around) and whether the line is annotated (:287-313)."""
        pts_out, masks = [], []
        for name in self.fl_names:
            if name not in fls:
                pts_out.append(np.zeros((self.fl_sampling, 2), np.float32))
                masks.append(False)
                continue
            pts = fls[name]
            gap = ((pts[:-1] - pts[1:]) ** 2).sum(-1)
            if ((pts[-1] - pts[0]) ** 2).sum(-1) < np.max(gap):
                cut = int(np.argmax(gap))
                pts = np.concatenate([pts[cut + 1:], pts[:cut + 1]], axis=0)
            pts_out.append(uniformsample(pts, self.fl_sampling))
            masks.append(True)
        return pts_out, masks

    def area_size_statistic(self):
        """Per-line weight of the projection loss: (largest mean extent / the line's mean extent)^2, the extent of a line
        in a frame being the larger side of its bounding box (:109-152; chamfer distances are squared lengths)."""
        total = {n: 0. for n in self.fl_names}
        seen = {n: 0 for n in self.fl_names}
        for idx in range(len(self.fl_paths)):
            if not self._counts_for_weights(idx):
                continue
            pts, masks = self.obtain_fl_pts(obtain_feature_lines(self.fl_paths[idx]))
            for p, ok, name in zip(pts, masks, self.fl_names):
                if ok:
                    ext = p.max(0) - p.min(0)
                    total[name] += max(ext[0], ext[1])
                    seen[name] += 1
        mean = {n: total[n] / seen[n] for n in self.fl_names}
        top = max([0.] + list(mean.values()))
        self.fl_weights = {n: (top / mean[n]) ** 2 for n in self.fl_names}

    # ---------------------------------------------------------------------------------------------- parsing masks
    def load_parsing_mask(self, mask, parsing_logits):
        """Every foreground pixel takes the label of the nearest labelled pixel (:317-337; the reference runs pytorch3d's
        knn_points on the GPU; here a chunked exact search wherever the tensors live)."""
        out = torch.zeros_like(mask).long()
        li, lj = torch.nonzero(parsing_logits, as_tuple=True)
        label = parsing_logits[li, lj]
        src = torch.stack([li, lj], dim=-1).float()
        mi, mj = torch.nonzero(mask, as_tuple=True)
        dst = torch.stack([mi, mj], dim=-1).float()
        if src.shape[0] and dst.shape[0]:
            dev = torch.device('cuda') if torch.cuda.is_available() else dst.device
            src_d, nearest = src.to(dev), []
            for chunk in dst.to(dev).split(4096):
                d = ((chunk[:, None, :] - src_d[None, :, :]) ** 2).sum(-1)
                nearest.append(d.argmin(dim=1).cpu())
            out[mi, mj] = label[torch.cat(nearest)].long()
        return out.numpy().astype(np.uint8)

    def _mask_parsing_path(self, idx):
        d, f = osp.split(self.parsing_mask_ns[idx])
        return osp.join(d, 'mask_parsing_' + f)

    def parsing_mask(self, idx):
        """Pre-processing step (:260-283): write `mask_parsing_<i>.npy` for frame idx, return its path."""
        parsing = torch.from_numpy(np.load(self.parsing_mask_ns[idx])).long()
        mask = self._read_mask(idx)
        path = self._mask_parsing_path(idx)
        np.save(path, self.load_parsing_mask(mask, parsing))
        return path

    def obtain_parsing_mask(self, mask_parsing):
        """Boolean regions 'upper' / 'bottom' / 'upper_bottom' (unions of ATR classes) and 'body' = labelled but in no
        garment region (:339-357)."""
        regions = {}
        any_garment = torch.zeros_like(mask_parsing, dtype=torch.bool)
        for key, classes in ATR_PARSING.items():
            region = torch.zeros_like(mask_parsing, dtype=torch.bool)
            for c in classes:
                region |= mask_parsing == c
            regions[key] = region
            any_garment |= region
        regions['body'] = (mask_parsing > 0) ^ any_garment
        return regions

    # ---------------------------------------------------------------------------------------------- samples
    def _read_mask(self, idx):
        return (torch.from_numpy(read_image_bgr(self.mask_ns[idx])) > 0).view(self.H, self.W, -1).any(-1).float()

    def _annotated(self, idx):
        return idx % self.curve_sampling == 0

    def _counts_for_weights(self, idx):
        return idx < len(self) and idx % self.curve_sampling == 0

    def _sample(self, idx):
        out = {}
        img = read_image_bgr(self.img_ns[idx]).astype(np.float32)
        out['img'] = torch.from_numpy((img / 255. - 0.5) * 2).view(self.H, self.W, 3)        # [-1,1], B,G,R
        mask_parsing = torch.from_numpy(np.load(self._mask_parsing_path(idx))).long()
        out['mask'] = self._read_mask(idx)
        fl_pts, fl_masks = self.obtain_fl_pts(obtain_feature_lines(self.fl_paths[idx]))
        fl_masks = torch.Tensor(fl_masks).bool()
        if not self._annotated(idx):
            fl_masks[...] = False
        out['fl_pts'] = torch.cat([torch.from_numpy(np.asarray(p)).float() for p in fl_pts], dim=0)
        out['fl_masks'] = fl_masks
        norm_f = self.img_ns[idx].replace('/imgs/', '/normals/')[:-3] + 'png'
        if osp.isfile(norm_f):
            normals = read_image_bgr(norm_f)[:, :, ::-1]                                    # R,G,B
            out['normal'] = 2. * normals.astype(np.float32) / 255. - 1.
        out.update(self.obtain_parsing_mask(mask_parsing))
        return out

    def __len__(self):
        return self.frame_num

    def all_size(self):
        return len(self)

    def __getitem__(self, idx):
        out = self._sample(idx)
        out['gt_joints2d'] = self.gt_joints2d[idx]
        if self.require_albedo:
            alb = read_image_bgr(osp.join(self.root, 'albedos/%d.png' % idx)).astype(np.float32)
            out['albedo'] = torch.from_numpy((alb / 255. - 0.5) * 2.).view(self.H, self.W, 3)
        return idx, out

    def get_init_fl_datasets(self, batch_size, sampler, num_workers):
        """The frames that supervise the feature lines, as a loader of their own for the start-up registration of the
        line templates (:97-106; same in every capture class).  `sampler` is accepted and replaced, as in the reference."""
        sampler_idx = np.where(np.asarray(self.fl_supervised))[0].tolist()
        init_fl = Init_Fl_SceneDataset(self.root, self.conds_lens, self.garment_type, self.fl_sampling, self.curve_sampling,
                                       sampler_idx)
        return torch.utils.data.DataLoader(init_fl, batch_size, sampler=RandomSampler(init_fl, 1, True),
                                           num_workers=num_workers)

    # ---------------------------------------------------------------------------------------------- learnable state
    def opt_camera_params(self, conf):
        keys = {'focal_length': 'focal_length', 'princeple_points': 'princeple_points', 'cam2world_coord_quat': 'quat',
                'world2cam_coord_trans': 'T'}
        for name, conf_key in keys.items():
            self.camera_params[name].requires_grad_(conf if type(conf) == bool else conf.get_bool(conf_key))

    def learnable_weights(self):
        ws = [c for c in self.conds if c.requires_grad]
        ws.extend(v for v in self.camera_params.values() if v.requires_grad)
        ws.extend(v for v in (self.shape, self.poses, self.trans) if v.requires_grad)
        return ws

    def get_grad_parameters(self, idxs, device):
        """(poses, trans, *conds) of the frames `idxs` on `device`; fetched here because a DataLoader cannot collate tensors
        that require grad (:425-433)."""
        conds = [c[idxs].to(device) for c in self.conds]
        if len(conds) > 1:
            return (self.poses[idxs].to(device), self.trans[idxs].to(device), *conds)
        return (self.poses[idxs].to(device), self.trans[idxs].to(device), *conds, None)

    def get_camera_parameters(self, N, device):
        from ..utils import quat2mat
        cp = self.camera_params
        return (cp['focal_length'].to(device).view(1, 2).expand(N, 2), cp['princeple_points'].to(device).view(1, 2).expand(N, 2),
                quat2mat(cp['cam2world_coord_quat'].to(device).view(1, 4)).expand(N, 3, 3),
                cp['world2cam_coord_trans'].to(device).view(1, 3).expand(N, 3), self.H, self.W)

    def get_batchframe_data(self, name, fids, batchsize):
        """Windows of `batchsize` consecutive frames around each frame id, shifted to stay inside the video (or inside the
        frame's segment when the capture is two videos), and each frame's position in its window (:438-501)."""
        assert hasattr(self, name)
        data = getattr(self, name)
        assert data.shape[0] >= self.frame_num
        data = data[:self.frame_num].to(fids.device)
        if len(self.video_segmented_index) == 0:
            segments = [(0, self.frame_num)]
        elif len(self.video_segmented_index) == 1:
            cut = self.video_segmented_index[0]
            segments = [(0, cut), (cut, self.frame_num)]
        else:
            raise NotImplementedError
        starts = torch.zeros_like(fids) - 1
        for lo, hi in segments:
            assert batchsize < hi - lo
            sel = (fids >= lo) & (fids < hi)
            s = fids[sel] - batchsize // 2
            s = torch.where(s < lo, torch.full_like(s, lo), s)
            s = torch.where(s + batchsize > hi, torch.full_like(s, hi - batchsize), s)
            starts[sel] = s
        assert bool((starts >= 0).all())
        window = starts.view(-1, 1) + torch.arange(0, batchsize, device=fids.device).view(1, batchsize)
        return data[window], fids - starts


class People_Snapshot_SceneDataset(SceneDataset):
    """PeopleSnapshot captures (:503-679): the feature lines are annotated on the A-pose turn only (`mask2fl/`); frames
    without an annotation of their own are not supervised by the curves."""

    def __init__(self, data_root, conds_lens={}, garment_type="", fl_sampling=100, curve_sampling=1, a_pose=False):
        super().__init__(data_root, conds_lens, garment_type, fl_sampling, curve_sampling=curve_sampling)
        self.a_pose = a_pose
        fl_dir = osp.join(data_root, 'mask2fl')
        self.start_idx = 0
        if osp.exists(fl_dir):
            self.read_feature_lines(fl_dir)
            self.area_size_statistic()
            n_all = len(self)
            if self.a_pose:
                self.frame_num = self.a_pose_end - self.a_pose_start + 1
                self.start_idx = self.a_pose_start
            else:
                self.frame_num = n_all - self.a_pose_end - 1
                self.start_idx = self.a_pose_end + 1

    def read_feature_lines(self, path):
        # (also called by the base constructor for `featurelines/`, as in the reference, :588-612)
        files, self.fl_paths, self.fl_supervised = self._assign_feature_line_files(path)
        self.a_pose_start, self.a_pose_end = _frame_number(files[0]), _frame_number(files[-1])

    def _annotated(self, idx):
        return bool(self.fl_supervised[idx])

    def _counts_for_weights(self, idx):          # the frames with an annotation of their own (:542-587)
        return bool(self.fl_supervised[idx])

    def __getitem__(self, idx):
        out = self._sample(idx)
        if self.require_albedo:
            alb = read_image_bgr(osp.join(self.root, 'albedos/%d.png' % idx)).astype(np.float32)
            out['albedo'] = torch.from_numpy((alb / 255. - 0.5) * 2.).view(self.H, self.W, 3)
        return idx, out


class Synthe_SceneDataset(SceneDataset):
    """Synthetic-outfit renders (:1004-1064): a capture whose every frame carries its feature lines (no `curve_sampling` thinning)
    and that has no 2-D joints (its SMPL shape is known, no shape fit)."""

    def _annotated(self, idx):
        return True

    def __getitem__(self, idx):
        out = self._sample(idx)
        if self.require_albedo:
            alb = read_image_bgr(osp.join(self.root, 'albedos/%d.png' % idx)).astype(np.float32)
            out['albedo'] = torch.from_numpy((alb / 255. - 0.5) * 2.).view(self.H, self.W, 3)
        return idx, out


class Init_Fl_SceneDataset(SceneDataset):
    """A capture restricted to the frames `sample_idx` (:894-1000): what `get_init_fl_datasets` hands to the start-up
    registration.  Feature lines come from `mask2fl/` when the capture has it, else `featurelines/`; a frame without an
    annotation file of its own has every line flagged absent."""

    def __init__(self, data_root, conds_lens={}, garment_type="", fl_sampling=100, curve_sampling=1, sample_idx=[]):
        super().__init__(data_root, conds_lens, garment_type, fl_sampling, curve_sampling=curve_sampling)
        fl_dir = osp.join(data_root, 'mask2fl')
        if not osp.exists(fl_dir):
            fl_dir = osp.join(data_root, 'featurelines')
        self.read_feature_lines(fl_dir)
        self.frame_num = len(sample_idx)
        self.idx = list(sample_idx)

    def read_feature_lines(self, path):
        files, self.fl_paths, self.fl_supervised = self._assign_feature_line_files(path)
        self.a_pose_start, self.a_pose_end = _frame_number(files[0]), _frame_number(files[-1])

    def _annotated(self, idx):
        return bool(self.fl_supervised[idx])

    def __getitem__(self, idx):
        idx = self.idx[idx]
        out = self._sample(idx)
        if self.gt_joints2d is not None:
            out['gt_joints2d'] = self.gt_joints2d[idx]
        if self.require_albedo:
            alb = read_image_bgr(osp.join(self.root, 'albedos/%d.png' % idx)).astype(np.float32)
            out['albedo'] = torch.from_numpy((alb / 255. - 0.5) * 2.).view(self.H, self.W, 3)
        return idx, out


def one_euro_smooth(seq, min_cutoff=0.004, beta=1.5, d_cutoff=1.):
    """One-euro filter (Casiez et al. 2012: an exponential smoother whose cutoff grows with the smoothed speed) along the first
    axis of `seq`, unit time steps, as `smooth_poses` of the reference applies it (engineer/utils/smooth_poses.py:34-71 over
    engineer/utils/filter.py:14-58): x_hat_0 = x_0, dx_hat_0 = 0;  a(c) = 2 pi c / (2 pi c + 1);
    dx_hat = a(d_cutoff) (x - x_hat_prev) + (1 - a(d_cutoff)) dx_hat_prev;  x_hat = a(c) x + (1 - a(c)) x_hat_prev with
    c = min_cutoff + beta |dx_hat|.  For axis-angle sequences [F,J,3] a sample that jumped by more than 0.5 (L1) from the
    previous estimate is first replaced by its equivalent rotation about the opposite axis (angle 2 pi - theta)."""
    x = seq.detach().clone()
    out = torch.zeros_like(x)
    out[0] = x[0]
    x_prev, dx_prev = x[0], torch.zeros_like(x[0])

    def alpha(cutoff):
        r = 2 * np.pi * cutoff
        return r / (r + 1)

    for i in range(1, x.shape[0]):
        cur = x[i]
        jump = torch.abs(x[i] - out[i - 1]).sum(-1)
        if cur.dim() >= 2 and bool(jump.max() > 0.5):
            angle = torch.norm(cur + 1e-8, p=2, dim=1).unsqueeze(-1)
            flip = 1 + (2 * np.pi - 2 * angle) / angle
            mask = jump > 0.5
            cur = cur.clone()
            cur[mask] = -cur[mask]
            cur[mask] *= flip[mask]
        dx = cur - x_prev
        dx_hat = alpha(d_cutoff) * dx + (1 - alpha(d_cutoff)) * dx_prev
        a = alpha(min_cutoff + beta * torch.abs(dx_hat))
        x_hat = a * cur + (1 - a) * x_prev
        x_prev, dx_prev = x_hat, dx_hat
        out[i] = x_hat
    return out


def _lower_bound(arr, target):
    """First position whose value is >= target (utils/common_utils.py:30-39)."""
    lo, hi = 0, len(arr)
    while lo < hi:
        mid = (lo + hi) >> 1
        if arr[mid] < target:
            lo = mid + 1
        else:
            hi = mid
    return lo


class Large_Pose_SceneDataset(People_Snapshot_SceneDataset):
    """Large-pose captures (:681-892): as PeopleSnapshot, with the depth of the translation frozen after the A-pose turn and
    the translations smoothed, the SMPL shape taken from the TCMR betas of the A-pose turn, the poses after it replaced by the
    TCMR poses of the nearest annotated frame; samples are addressed relative to `start_idx`."""

    def __init__(self, data_root, conds_lens={}, garment_type="", fl_sampling=100, curve_sampling=1, a_pose=False):
        SceneDataset.__init__(self, data_root, conds_lens, garment_type, fl_sampling, curve_sampling=curve_sampling)
        self.a_pose = a_pose
        n_all = len(self)
        joints_idx = [_lower_bound(self.joints_frame_ids, idx) for idx in range(n_all)]
        first, last = self.a_pose_start, self.a_pose_end
        self.trans[last:, 2] = self.trans[last, 2]                                   # depth frozen after the A-pose turn
        self.trans = one_euro_smooth(self.trans.detach().cpu(), min_cutoff=0.004, beta=0.7, d_cutoff=1.)
        self.shape = torch.from_numpy(self.tcmr_betas[first:last + 1].mean(0)).float()
        tcmr = torch.from_numpy(self.tcmr_poses[joints_idx]).float().view(-1, 24, 3)
        self.poses[last + 1:len(tcmr)] = tcmr[last + 1:]
        fl_dir = osp.join(data_root, 'mask2fl')
        self.start_idx = 0
        if osp.exists(fl_dir):
            self.read_feature_lines(fl_dir)
            self.area_size_statistic()
            if self.a_pose:
                self.frame_num = self.a_pose_end - self.a_pose_start + 1
                self.start_idx = 0
            else:
                self.frame_num = n_all - self.a_pose_end - 1
                self.start_idx = self.a_pose_end + 1

    def all_size(self):
        return self.poses.shape[0]

    def __getitem__(self, idx):
        idx = self.start_idx + idx
        out = self._sample(idx)
        out['gt_joints2d'] = self.gt_joints2d[self.joints_frame_ids[_lower_bound(self.joints_frame_ids, idx)]]
        if self.require_albedo:
            alb = read_image_bgr(osp.join(self.root, 'albedos/%d.png' % idx)).astype(np.float32)
            out['albedo'] = torch.from_numpy((alb / 255. - 0.5) * 2.).view(self.H, self.W, 3)
        return idx, out


class ClipSampler(torch.utils.data.Sampler):
    """Consecutive clips of `clip_size` frames in random order, from a random offset (:1113-1133)."""

    def __init__(self, data_source, clip_size, shuffle):
        self.data_source, self.clip_size, self.shuffle = data_source, clip_size, shuffle
        n_frames = len(data_source)
        self.n = n_frames // clip_size
        if n_frames == self.n * clip_size:
            self.n -= 1
        self.start = n_frames - self.n * clip_size

    def __iter__(self):
        first = random.sample(range(self.start + 1), 1)[0] if self.shuffle else 0
        assert first + self.n * self.clip_size <= len(self.data_source)
        clips = torch.arange(first, first + self.n * self.clip_size).view(self.n, self.clip_size)
        if self.shuffle:
            clips = clips[torch.randperm(self.n)]
        return iter(clips.view(-1).tolist())

    def __len__(self):
        return self.n * self.clip_size


class RandomSampler(torch.utils.data.Sampler):
    """Every `intersect`-th frame from a random offset, shuffled (:1135-1157); `intersect` = 1 in the training driver."""

    def __init__(self, data_source, intersect, shuffle):
        self.length, self.intersect, self.shuffle = len(data_source), intersect, shuffle
        self.n = (self.length - 1) // intersect + 1
        self.start = self.length - intersect * (self.n - 1)

    def __iter__(self):
        first = random.sample(range(self.start), 1)[0] if self.shuffle else 0      # (one draw of Python's generator)
        picks = torch.arange(first, self.length, self.intersect)
        assert picks.numel() == self.n
        if self.shuffle:
            picks = picks[torch.randperm(self.n)]                                  # (one permutation of torch's)
        return iter(picks.tolist())

    def __len__(self):
        return self.n


def getDatasetAndLoader(root, conds_lens, batch_size, shuffle, num_workers, opt_pose, opt_trans, opt_camera, garment_type,
                        data_type=None, curve_sampling=1, a_pose=False):
    """dataset/dataset.py:1159-1183: the capture as a dataset with its learnable tensors switched on, and a DataLoader
    over a shuffled RandomSampler."""
    with_a_pose = {'people_snap': People_Snapshot_SceneDataset, 'large_pose': Large_Pose_SceneDataset}
    if data_type == 'scene':
        dataset = SceneDataset(root, conds_lens, garment_type, curve_sampling=curve_sampling)
    elif data_type in with_a_pose:
        dataset = with_a_pose[data_type](root, conds_lens, garment_type, curve_sampling=curve_sampling, a_pose=a_pose)
    elif data_type == 'synthe':
        dataset = Synthe_SceneDataset(root, conds_lens, garment_type, curve_sampling=curve_sampling)
    elif data_type == 'snug':
        raise NotImplementedError("data type snug: it animates a capture with CMU motion files of the SNUG repository "
                                  "(../snug/assets, `load_motion`), which are outside this package (recmv/dataset/dataset.py)")
    else:
        raise NotImplementedError('data type {} is not implemented'.format(data_type))
    for tensor, learn in ((dataset.poses, opt_pose), (dataset.trans, opt_trans)):
        if learn:
            tensor.requires_grad_(True)
    dataset.opt_camera_params(opt_camera)
    loader = torch.utils.data.DataLoader(dataset, batch_size, sampler=RandomSampler(dataset, 1, shuffle),
                                         num_workers=num_workers)
    return dataset, loader
