"""Input producer with the reference's dataset interface over frames that are RESIDENT in HBM.

Mirrors code/lib/datasets/Hi4D.py: `Hi4DDataset` (:90-306), `Hi4DValDataset` (:329-363), `Hi4DTestDataset` (:365-484, the
training-camera branch) and the on-disk scene format they read (written by preprocessing/preprocessing_multiple_trace.py):
    <root>/image/*.png, mask/<person>/*.png, [edge/*.png], poses.npy (F,P,72), mean_shape.npy (P,10),
    normalize_trans.npy (F,P,3), cameras_normalize.npz (scale_mat_i, world_mat_i), gender.npy
The reference decodes the frame's PNGs (cv2.imread) and gathers 512 sub-pixel samples on the CPU for EVERY item, in 8
DataLoader workers.  Here the sequence is decoded once, its frames live in device memory as bytes (a 300-frame 1 MP
sequence is < 1.2 GB of the 288 GB), and an item is one kernel launch (csrc/data.hip, mp_sample_pixels) over positions
drawn on the host from the same numpy random stream, in the same order, as `weighted_sampling` (:59-88) -- use it with
DataLoader(num_workers=0).  Item dicts carry the reference's keys; per-sample tensors are device tensors.
"""
import glob
import os

import numpy as np
import torch

from . import hip


def read_png_rgb(path):
    """8-bit RGB bytes of a PNG (the reference: cv2.imread -> BGR, flipped to RGB at Hi4D.py:232)"""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def gray_nonzero(rgb):
    """cv2.cvtColor(BGR2GRAY) > 0 (Hi4D.py:238-240) with OpenCV's 8-bit fixed-point luma"""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14) > 0


def load_K_Rt_from_P(P):
    """rend_util.load_K_Rt_from_P (code/lib/utils/rend_util.py:21-42): P[:3,:3] = K R with K upper triangular and a
    positive diagonal, camera centre C = -M^-1 p4.  (cv2.decomposeProjectionMatrix in the reference; its algorithm,
    RQ decomposition, written with numpy.)  -> intrinsics (4,4) float64 with K/K[2,2], pose (4,4) float32 [R^T | C]."""
    P = np.asarray(P, dtype=np.float64)
    M = P[:3, :3]
    q, u = np.linalg.qr(np.flipud(M).T)
    K = np.flipud(np.fliplr(u.T))
    R = np.flipud(q.T)
    for _ in range(2):
        s = np.where(np.diag(K) < 0, -1.0, 1.0)
        K, R = K * s[None, :], s[:, None] * R
        if np.linalg.det(R) > 0:
            break
        K, R = -K, -R                     # P is homogeneous: flip its sign so that R is a proper rotation
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K / K[2, 2]
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.T
    pose[:3, 3] = -np.linalg.solve(M, P[:3, 3])
    return intrinsics, pose


class SceneStore:
    """All frames [start, end) of a preprocessed sequence, decoded once and kept in device memory:
    images (F,H,W,3) uint8 RGB, object_masks (F,H,W) uint8 = number of person masks covering the pixel
    (Hi4D.py:236-245), optional edge masks, and per frame the mask's bounding box for the sampler."""

    def __init__(self, root, frame_ids, device, with_edges=False):
        img_paths = sorted(glob.glob(os.path.join(root, "image", "*.png")))
        self.img_paths = [img_paths[i] for i in frame_ids]
        folders = sorted(glob.glob(os.path.join(root, "mask", "*")))
        if folders and all(f.endswith(".png") for f in folders):      # threedpw.py:83-84: one person, mask/*.png
            folders = [os.path.join(root, "mask")]
        self.mask_paths = [[sorted(glob.glob(os.path.join(f, "*.png")))[i] for i in frame_ids] for f in folders]
        first = read_png_rgb(self.img_paths[0])
        self.img_size = first.shape[:2]
        F, (H, W) = len(self.img_paths), self.img_size
        imgs = np.empty((F, H, W, 3), dtype=np.uint8)
        masks = np.empty((F, H, W), dtype=np.uint8)
        self.bbox = np.empty((F, 2, 2), dtype=np.int64)           # [frame][min|max][row|col]
        for i in range(F):
            imgs[i] = first if i == 0 else read_png_rgb(self.img_paths[i])
            m = np.zeros((H, W), dtype=np.uint8)
            for person in self.mask_paths:
                m += gray_nonzero(read_png_rgb(person[i]))
            masks[i] = m
            where = np.asarray(np.where(m))
            self.bbox[i, 0], self.bbox[i, 1] = where.min(axis=1), where.max(axis=1)
        self.images = torch.from_numpy(imgs).to(device)
        self.object_masks = torch.from_numpy(masks).to(device)
        self.edge_masks = None
        if with_edges:
            edge_paths = sorted(glob.glob(os.path.join(root, "edge", "*.png")))
            edges = np.stack([gray_nonzero(read_png_rgb(edge_paths[i])) for i in frame_ids])
            self.edge_masks = torch.from_numpy(edges & (masks > 0)).to(device)      # Hi4D.py:247-251
        self.device = device

    def sample(self, frame, pos, extra=None):
        """pos (n,2) float64 (row, col) -> rgb (n,3), uv (n,2), object_mask (n,) [, extra (n,C)] fp32 device tensors"""
        L = hip.lib()
        H, W = self.img_size
        n = pos.shape[0]
        dpos = torch.from_numpy(np.ascontiguousarray(pos, dtype=np.float64)).to(self.device)
        f32 = dict(dtype=torch.float32, device=self.device)
        rgb, uv, om = torch.empty(n, 3, **f32), torch.empty(n, 2, **f32), torch.empty(n, **f32)
        n_extra, ex_out = 0, None
        if extra is not None:
            extra = extra.to(self.device).float().contiguous()
            n_extra = extra.shape[-1]
            ex_out = torch.empty(n, n_extra, **f32)
        hip.check(L.mp_sample_pixels(hip.ptr(self.images[frame]), hip.ptr(self.object_masks[frame]), hip.ptr(extra), n_extra,
                                     hip.ptr(dpos), n, H, W, hip.ptr(rgb), hip.ptr(uv), hip.ptr(om), hip.ptr(ex_out),
                                     hip.stream()), "mp_sample_pixels")
        self._keep = (dpos, extra)
        return rgb, uv, om, ex_out


def draw_positions(bbox_min, bbox_max, img_size, num_sample, rng=np.random):
    """the random draws of weighted_sampling (Hi4D.py:61-75), in its order: rand(n_bbox, 2), then rand(n_uniform, 2)
    -> positions (num_sample, 2) float64 (row, col), index_outside (uniform samples that fell outside the box)"""
    n_bbox = int(num_sample * 0.9)
    pos_bbox = rng.rand(n_bbox, 2) * (bbox_max - bbox_min) + bbox_min
    pos_uni = rng.rand(num_sample - n_bbox, 2)
    pos_uni *= (img_size[0] - 1, img_size[1] - 1)
    r, c = pos_uni[:, 0], pos_uni[:, 1]
    outside = np.where((r < bbox_min[0]) | (r > bbox_max[0]) | (c < bbox_min[1]) | (c > bbox_max[1]))[0] + n_bbox
    return np.concatenate([pos_bbox, pos_uni], axis=0), outside


class Hi4DDataset(torch.utils.data.Dataset):
    """Training items (Hi4D.py:90-306).  opt: data_dir, start_frame, end_frame, num_sample, using_SAM, and optionally
    ratio_uncertain, ratio_decrease, edge_sampling, data_root (default '../data' like the reference)."""

    def __init__(self, opt, device=None, rng=None):
        hip.require_device()
        self.device = torch.device("cuda") if device is None else device
        self.rng = np.random if rng is None else rng
        root = os.path.abspath(os.path.join(opt.get("data_root", "../data"), opt.data_dir))
        self.root = root
        self.start_frame, self.end_frame, self.skip_step = opt.start_frame, opt.end_frame, 1
        self.training_indices = list(range(opt.start_frame, opt.end_frame, self.skip_step))
        self.init_params(opt)
        self.store = SceneStore(root, self.training_indices, self.device, with_edges=self.edge_sampling)
        self.img_paths, self.img_size = self.store.img_paths, self.store.img_size
        self.n_images = len(self.img_paths)
        self.shape = np.load(os.path.join(root, "mean_shape.npy"))
        self.num_person = self.shape.shape[0]
        self.poses = np.load(os.path.join(root, "poses.npy"))[self.training_indices]
        self.trans = np.load(os.path.join(root, "normalize_trans.npy"))[self.training_indices]
        cams = np.load(os.path.join(root, "cameras_normalize.npz"))
        self.scale_mat_all = [cams["scale_mat_%d" % i].astype(np.float32) for i in self.training_indices]
        self.world_mat_all = [cams["world_mat_%d" % i].astype(np.float32) for i in self.training_indices]
        self.scale = 1 / self.scale_mat_all[0][0, 0]
        self.P, self.C, self.intrinsics_all, self.pose_all = [], [], [], []
        for scale_mat, world_mat in zip(self.scale_mat_all, self.world_mat_all):
            P = world_mat @ scale_mat
            self.P.append(P)
            self.C.append(-np.linalg.solve(P[:3, :3], P[:3, 3]))
            K, pose = load_K_Rt_from_P(P[:3, :4])
            self.intrinsics_all.append(torch.from_numpy(K).float())
            self.pose_all.append(torch.from_numpy(pose).float())
        self.num_sample = opt.num_sample
        self.sampling_strategy = "weighted"
        self.using_SAM = opt.get("using_SAM", False)       # threedpw.py has no such option
        self.pre_mask_path, self.pre_mask = "", None
        self.smpl_sam_iou = np.ones(self.n_images)
        self.uncertain_thereshold = 0.0
        self.uncertain_frame_list = []

    def init_params(self, opt):
        self.ratio_uncertain = opt.get("ratio_uncertain", 0.5)      # the higher, the more uncertain frames
        self.ratio_decrease = opt.get("ratio_decrease", 0.0)        # per mask refresh
        self.edge_sampling = opt.get("edge_sampling", False)

    def __len__(self):
        return self.n_images

    def load_body_model_params(self):
        return {}

    def _sam_mask(self, idx):
        """the trainer's refreshed SAM masks (Hi4D.py:184-226): newest stage_sam_mask/*/sam_opt_mask.npy, (F,P,H,W) logits;
        frames whose SAM / SMPL mask IoU is below the `ratio_uncertain` quantile are flagged uncertain"""
        mask_list = sorted(glob.glob("stage_sam_mask/*"))
        if len(mask_list) == 0:
            return None
        mask_path = os.path.join(mask_list[-1], "sam_opt_mask.npy")
        if mask_path != self.pre_mask_path:
            smpl_mask = np.load(os.path.join(sorted(glob.glob("stage_instance_mask/*"))[-1], "all_person_smpl_mask.npy")) > 0.8
            try:
                logits = np.load(mask_path)
            except Exception:
                print("ERROR: cannot load current sam mask, use previous sam mask")
                mask_path = self.pre_mask_path
                logits = np.load(mask_path)
            binary = logits > 0.0
            iou = np.logical_and(binary, smpl_mask).sum(axis=(2, 3)) / np.logical_or(binary, smpl_mask).sum(axis=(2, 3))
            self.smpl_sam_iou = iou.mean(axis=-1)
            self.uncertain_thereshold = np.sort(self.smpl_sam_iou)[int(len(self.smpl_sam_iou) * self.ratio_uncertain)]
            self.ratio_uncertain -= self.ratio_decrease
            self.uncertain_frame_list = [i for i, v in enumerate(self.smpl_sam_iou) if v < self.uncertain_thereshold]
            self.pre_mask_path = mask_path
            self.pre_mask = torch.from_numpy(np.ascontiguousarray(logits.transpose(0, 2, 3, 1))).float().to(self.device)
        return self.pre_mask[idx]

    def smpl_params(self, idx):
        p = torch.zeros([self.num_person, 86]).float()
        p[:, 0] = torch.from_numpy(np.asarray(self.scale)).float()
        p[:, 1:4] = torch.from_numpy(self.trans[idx]).float()
        p[:, 4:76] = torch.from_numpy(self.poses[idx]).float()
        p[:, 76:] = torch.from_numpy(self.shape).float()
        return p

    def full_uv(self):
        H, W = self.img_size
        rows, cols = torch.meshgrid(torch.arange(H, device=self.device), torch.arange(W, device=self.device), indexing="ij")
        return torch.stack([cols, rows], dim=-1).float()             # uv[r][c] = (c, r)  (Hi4D.py:254-255)

    def __getitem__(self, idx):
        is_certain, sam_mask = True, None
        if self.using_SAM:
            sam_mask = self._sam_mask(idx)
            is_certain = self.smpl_sam_iou[idx] >= self.uncertain_thereshold
        st, (H, W) = self.store, self.img_size
        smpl_params = self.smpl_params(idx)
        org_img = st.images[idx].float() / 255
        if self.num_sample > 0:
            pos, index_outside = draw_positions(st.bbox[idx, 0], st.bbox[idx, 1], (H, W), self.num_sample, self.rng)
            rgb, uv, _, sam = st.sample(idx, pos, sam_mask)
            inputs = {"uv": uv, "P": self.P[idx], "C": self.C[idx], "intrinsics": self.intrinsics_all[idx],
                      "pose": self.pose_all[idx], "smpl_params": smpl_params, "index_outside": index_outside, "idx": idx,
                      "smpl_sam_iou": self.smpl_sam_iou, "is_certain": is_certain, "org_img": org_img,
                      "img_size": self.img_size}
            images = {"rgb": rgb}
            if sam_mask is not None:
                inputs.update({"sam_mask": sam, "org_sam_mask": sam_mask})
            if self.edge_sampling:             # Hi4D.py:28-56: integer pixel picks, 50 % mask / 40 % edge / 10 % anywhere
                n_mask, n_edge = int(self.num_sample * 0.5), int(self.num_sample * 0.4)
                mask_loc = torch.nonzero(st.object_masks[idx].reshape(-1) > 0).flatten()
                edge_loc = torch.nonzero(st.edge_masks[idx].reshape(-1)).flatten()
                pick = torch.cat([mask_loc[torch.from_numpy(self.rng.randint(0, len(mask_loc), n_mask)).to(self.device)],
                                  edge_loc[torch.from_numpy(self.rng.randint(0, len(edge_loc), n_edge)).to(self.device)],
                                  torch.from_numpy(self.rng.randint(0, H * W, self.num_sample - n_mask - n_edge)).to(self.device)])
                inputs["edge_uv"] = self.full_uv().reshape(-1, 2)[pick]
                images["edge_rgb"] = org_img.reshape(-1, 3)[pick]
                if sam_mask is not None:
                    inputs["edge_sam_mask"] = sam_mask.reshape(H * W, -1)[pick]
            return inputs, images
        uv = self.full_uv()
        inputs = {"uv": uv.reshape(-1, 2), "P": self.P[idx], "C": self.C[idx], "intrinsics": self.intrinsics_all[idx],
                  "pose": self.pose_all[idx], "smpl_params": smpl_params, "idx": idx, "org_uv": uv, "org_img": org_img,
                  "org_object_mask": st.object_masks[idx], "img_size": self.img_size}
        if sam_mask is not None:
            inputs["org_sam_mask"] = sam_mask
        return inputs, {"rgb": org_img.reshape(-1, 3), "img_size": self.img_size}


class Hi4DValDataset(torch.utils.data.Dataset):
    """one random frame per epoch (Hi4D.py:329-363)"""

    def __init__(self, opt, device=None, rng=None):
        self.dataset = Hi4DDataset(opt, device, rng)
        self.img_size = self.dataset.img_size
        self.total_pixels = np.prod(self.img_size)
        self.pixel_per_batch = opt.pixel_per_batch

    def __len__(self):
        return 1

    def __getitem__(self, idx):
        image_id = int(self.dataset.rng.choice(len(self.dataset), 1)[0])
        inputs, images = self.dataset[image_id]
        inputs = {k: inputs[k] for k in ("uv", "P", "C", "intrinsics", "pose", "smpl_params", "idx")}
        inputs["image_id"] = image_id
        images = {"rgb": images["rgb"], "img_size": images["img_size"], "pixel_per_batch": self.pixel_per_batch,
                  "total_pixels": self.total_pixels}
        return inputs, images


def novel_view_camera(scale_mat, world_mat, gt_cur, gt_tgt):
    """One frame of the novel-view cameras (Hi4D.py:398-425).  The sequence was trained under the studio camera
    `current_view`, whose ground-truth calibration is gt_cur = (intrinsics (3,3), extrinsics (3,4)); gt_tgt is the studio
    camera to render from.  The rigid motion that carries the studio frame onto the training frame is read off the two
    descriptions of the current camera and applied to the target camera; the target intrinsics are brought to the training
    image scale (ratio of the two focal lengths of the current camera).  -> P (4,4), C (3,), intrinsics (4,4), pose (4,4)."""
    _, pose_tr = load_K_Rt_from_P(np.asarray(world_mat)[:3, :4])
    K_tr, _ = load_K_Rt_from_P(np.asarray(world_mat)[:3, :4])
    (K_cur, E_cur), (K_tgt, E_tgt) = gt_cur, gt_tgt
    zoom = K_cur[0, 0] / K_tr[0, 0]
    R_tr = pose_tr[:3, :3].T                                  # world -> camera of the training description
    t_tr = -R_tr @ pose_tr[:3, 3]
    R_rel = R_tr.T @ E_cur[:3, :3]                            # studio frame -> training frame
    t_rel = R_tr.T @ (E_cur[:3, 3] - t_tr)
    R_new = E_tgt[:3, :3] @ R_rel.T
    t_new = E_tgt[:3, 3] - R_new @ t_rel
    K_new = np.array(K_tgt[:3, :3], dtype=np.float64, copy=True)
    K_new[[0, 1, 0, 1], [0, 1, 2, 2]] /= zoom                 # fx, fy, cx, cy (the skew entry is left as it is)
    world_new = np.eye(4)
    world_new[:3, :4] = K_new @ np.concatenate([R_new, t_new[:, None]], axis=1)
    P = world_new @ scale_mat
    K, pose = load_K_Rt_from_P(P[:3, :4])
    return P, -np.linalg.solve(P[:3, :3], P[:3, 3]), K, pose


class Hi4DTestDataset(torch.utils.data.Dataset):
    """every frame under its training camera, or -- with opt.novel_view / current_view / pair / action / GT_DIR -- under
    another studio camera of the Hi4D ground truth <GT_DIR>/<pair>/<action>/cameras/rgb_cameras.npz (ids, intrinsics,
    extrinsics)  (Hi4D.py:365-484)"""

    def __init__(self, opt, device=None, rng=None):
        self.dataset = Hi4DDataset(opt, device, rng)
        self.img_size = self.dataset.img_size
        self.total_pixels = np.prod(self.img_size)
        self.pixel_per_batch = opt.pixel_per_batch
        self.novel_view, self.current_view = opt.get("novel_view", None), opt.get("current_view", None)
        if self.novel_view is not None and self.current_view is not None:
            cams = dict(np.load(os.path.join(opt.GT_DIR, opt.pair, opt.action, "cameras", "rgb_cameras.npz")))
            pick = lambda view: int(np.where(cams["ids"] == view)[0][0])
            cur, tgt = pick(self.current_view), pick(self.novel_view)
            gt_cur = (cams["intrinsics"][cur], cams["extrinsics"][cur])
            gt_tgt = (cams["intrinsics"][tgt], cams["extrinsics"][tgt])
            self.new_P, self.new_C, self.new_intrinsics_all, self.new_pose_all = [], [], [], []
            for scale_mat, world_mat in zip(self.dataset.scale_mat_all, self.dataset.world_mat_all):
                P, C, K, pose = novel_view_camera(scale_mat, world_mat, gt_cur, gt_tgt)
                self.new_P.append(P)
                self.new_C.append(C)
                self.new_intrinsics_all.append(torch.from_numpy(K).float())
                self.new_pose_all.append(torch.from_numpy(pose).float())
        else:
            self.novel_view = self.current_view = None

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        inputs, images = self.dataset[idx]
        if self.novel_view is not None:                       # Hi4D.py:433-450
            new = {"uv": inputs["uv"], "P": self.new_P[idx], "C": self.new_C[idx], "intrinsics": self.new_intrinsics_all[idx],
                   "pose": self.new_pose_all[idx], "smpl_params": inputs["smpl_params"], "idx": inputs["idx"],
                   "novel_view": self.novel_view}
            return new, {"rgb": images["rgb"], "img_size": images["img_size"]}, self.pixel_per_batch, self.total_pixels, idx
        new = {k: inputs[k] for k in ("uv", "P", "C", "intrinsics", "pose", "smpl_params", "idx")}
        new["img_size"] = torch.from_numpy(np.array(inputs["img_size"]))
        images = {"rgb": images["rgb"], "img_size": images["img_size"], "org_uv": inputs["org_uv"],
                  "org_img": inputs["org_img"], "org_object_mask": inputs["org_object_mask"]}
        if "org_sam_mask" in inputs:
            new["org_sam_mask"] = inputs["org_sam_mask"]
            images["org_sam_mask"] = inputs["org_sam_mask"]
        return new, images, self.pixel_per_batch, self.total_pixels, idx


class ThreeDPWDataset(Hi4DDataset):
    """The single-person sequence layout (code/lib/datasets/threedpw.py:60-178): mask/*.png without person folders,
    poses.npy (F,72), normalize_trans.npy (F,3), mean_shape.npy (10,); items carry a flat smpl_params (86,) and none of
    the SAM / uncertainty keys."""

    def __init__(self, opt, device=None, rng=None):
        super().__init__(opt, device, rng)
        self.num_person, self.using_SAM = 1, False

    def smpl_params(self, idx):
        p = torch.zeros([86]).float()
        p[0] = torch.from_numpy(np.asarray(self.scale)).float()
        p[1:4] = torch.from_numpy(self.trans[idx]).float()
        p[4:76] = torch.from_numpy(self.poses[idx]).float()
        p[76:] = torch.from_numpy(self.shape).float()
        return p

    def __getitem__(self, idx):
        st, (H, W) = self.store, self.img_size
        base = {"P": self.P[idx], "C": self.C[idx], "intrinsics": self.intrinsics_all[idx], "pose": self.pose_all[idx],
                "smpl_params": self.smpl_params(idx)}
        if self.num_sample > 0:
            pos, index_outside = draw_positions(st.bbox[idx, 0], st.bbox[idx, 1], (H, W), self.num_sample, self.rng)
            rgb, uv, _, _ = st.sample(idx, pos)
            return {"uv": uv, **base, "index_outside": index_outside, "idx": idx}, {"rgb": rgb}
        return ({"uv": self.full_uv().reshape(-1, 2), **base, "idx": idx},
                {"rgb": (st.images[idx].float() / 255).reshape(-1, 3), "img_size": self.img_size})


class ThreeDPWValDataset(Hi4DValDataset):
    """threedpw.py:180-211"""

    def __init__(self, opt, device=None, rng=None):
        self.dataset = ThreeDPWDataset(opt, device, rng)
        self.img_size = self.dataset.img_size
        self.total_pixels = np.prod(self.img_size)
        self.pixel_per_batch = opt.pixel_per_batch


class ThreeDPWTestDataset(torch.utils.data.Dataset):
    """threedpw.py:213-243"""

    def __init__(self, opt, device=None, rng=None):
        self.dataset = ThreeDPWDataset(opt, device, rng)
        self.img_size = self.dataset.img_size
        self.total_pixels = np.prod(self.img_size)
        self.pixel_per_batch = opt.pixel_per_batch

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        inputs, images = self.dataset[idx]
        inputs = {k: inputs[k] for k in ("uv", "P", "C", "intrinsics", "pose", "smpl_params", "idx")}
        return inputs, {"rgb": images["rgb"], "img_size": images["img_size"]}, self.pixel_per_batch, self.total_pixels, idx


def find_dataset_using_name(name):
    """code/lib/datasets/__init__.py:5-17"""
    mapping = {"ThreeDPW": ThreeDPWDataset, "ThreeDPWVal": ThreeDPWValDataset, "ThreeDPWTest": ThreeDPWTestDataset,
               "Hi4D": Hi4DDataset, "Hi4DVal": Hi4DValDataset, "Hi4DTest": Hi4DTestDataset}
    if name not in mapping:
        raise ValueError(f"Fail to find dataset {name}")
    return mapping[name]


def create_dataset(opt, device=None):
    """code/lib/datasets/__init__.py:20-44: a DataLoader over the named dataset; always num_workers = 0 -- the frames are
    resident in device memory and an item is one kernel launch, there is nothing for worker processes to do."""
    dataset = find_dataset_using_name(opt.dataset)(opt, device)
    return torch.utils.data.DataLoader(dataset, batch_size=opt.batch_size, drop_last=opt.drop_last, shuffle=opt.shuffle,
                                       num_workers=0)
