"""CPU restatement of the reference's input producer -- TEST INFRASTRUCTURE ONLY (tests/, incl. the CPU-baseline timing tests/bench_producer.py); the
product path is multiply_amd/datasets.py + csrc/data.hip.

Follows code/lib/datasets/Hi4D.py: bilinear_interpolation (:8-20), get_index_outside_of_bbox (:22-26),
weighted_sampling (:59-88), Hi4DDataset.__init__ / __getitem__ (:90-306), and rend_util.load_K_Rt_from_P
(code/lib/utils/rend_util.py:21-42).  Pinned by tests/golden/dataset_golden.npz, generated from the reference's own
functions (tests/golden/make_dataset_golden.py).  Third-party pieces the reference calls and this image lacks (OpenCV):
  * cv2.imread -> PIL (same decoded bytes for 8-bit PNGs; channel order handled explicitly),
  * cv2.cvtColor(BGR2GRAY) -> OpenCV's documented fixed-point formula (R*4899 + G*9617 + B*1868 + 8192) >> 14,
  * cv2.decomposeProjectionMatrix -> RQ decomposition with a positive-diagonal K and the camera centre as the null
    vector of P (its published algorithm); parity for these three is UNPINNED (no OpenCV here to compare with).
"""
import glob
import os

import numpy as np


def bilinear_interpolation(xs, ys, dist_map):
    """Hi4D.py:8-20"""
    x1 = np.floor(xs).astype(np.int32)
    y1 = np.floor(ys).astype(np.int32)
    x2, y2 = x1 + 1, y1 + 1
    dx = np.stack([x2 - xs, xs - x1], axis=1)[:, None, :]
    dy = np.stack([y2 - ys, ys - y1], axis=1)[:, :, None]
    Q = np.stack([dist_map[x1, y1], dist_map[x1, y2], dist_map[x2, y1], dist_map[x2, y2]], axis=1).reshape(-1, 2, 2)
    return np.squeeze(dx @ Q @ dy)


def index_outside_of_bbox(samples_uniform, bbox_min, bbox_max):
    """Hi4D.py:22-26"""
    r, c = samples_uniform[:, 0], samples_uniform[:, 1]
    return np.where((r < bbox_min[0]) | (r > bbox_max[0]) | (c < bbox_min[1]) | (c > bbox_max[1]))[0]


def draw_positions(mask, img_size, num_sample, rng=np.random):
    """the random part of weighted_sampling (Hi4D.py:61-75): 90 % of the samples in the mask's bounding box, the rest
    uniform over the image; consumes rng.rand(n_bbox, 2) then rng.rand(n_uniform, 2)"""
    where = np.asarray(np.where(mask))
    bbox_min, bbox_max = where.min(axis=1), where.max(axis=1)
    n_bbox = int(num_sample * 0.9)
    samples_bbox = rng.rand(n_bbox, 2) * (bbox_max - bbox_min) + bbox_min
    samples_uniform = rng.rand(num_sample - n_bbox, 2)
    samples_uniform *= (img_size[0] - 1, img_size[1] - 1)
    index_outside = index_outside_of_bbox(samples_uniform, bbox_min, bbox_max) + n_bbox
    return np.concatenate([samples_bbox, samples_uniform], axis=0), index_outside


def weighted_sampling(data, img_size, num_sample, rng=np.random):
    """Hi4D.py:59-88"""
    indices, index_outside = draw_positions(data["object_mask"], img_size, num_sample, rng)
    out = {}
    for key, val in data.items():
        if val.ndim == 3:
            new = np.stack([bilinear_interpolation(indices[:, 0], indices[:, 1], val[:, :, i]) for i in range(val.shape[2])],
                           axis=-1)
        else:
            new = bilinear_interpolation(indices[:, 0], indices[:, 1], val)
        out[key] = new.reshape(-1, *val.shape[2:])
    return out, index_outside


def read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def bgr2gray(rgb):
    """OpenCV's 8-bit BGR2GRAY (color.cpp: fixed point, 14 fractional bits)"""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def load_K_Rt_from_P(P):
    """rend_util.py:21-42 with cv2.decomposeProjectionMatrix restated: P[:3,:3] = K R (K upper triangular, positive
    diagonal; R a rotation), camera centre C with P [C;1] = 0.  Returns intrinsics (4,4) float64, pose (4,4) float32."""
    P = np.asarray(P, dtype=np.float64)
    if np.linalg.det(P[:3, :3]) < 0:          # P is homogeneous: choose the sign that makes R a proper rotation
        P = -P
    M = P[:3, :3]
    # RQ from the QR of the row-reversed transpose; then force a positive diagonal on K
    Q, U = np.linalg.qr(np.flipud(M).T)
    K = np.flipud(np.fliplr(U.T))
    R = np.flipud(Q.T)
    sgn = np.where(np.diag(K) < 0, -1.0, 1.0)
    K, R = K * sgn[None, :], sgn[:, None] * R
    C = -np.linalg.solve(M, P[:3, 3])
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K / K[2, 2]
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.T
    pose[:3, 3] = C
    return intrinsics, pose


class Hi4DDatasetOracle:
    """Hi4DDataset (Hi4D.py:90-306) for sampling_strategy 'weighted', without SAM masks and edge sampling."""

    def __init__(self, root, start_frame, end_frame, num_sample):
        idx = list(range(start_frame, end_frame))
        self.img_paths = [sorted(glob.glob(os.path.join(root, "image", "*.png")))[i] for i in idx]
        self.mask_paths = [[sorted(glob.glob(os.path.join(folder, "*.png")))[i] for i in idx]
                           for folder in sorted(glob.glob(os.path.join(root, "mask", "*")))]
        self.img_size = read_png(self.img_paths[0]).shape[:2]
        self.shape = np.load(os.path.join(root, "mean_shape.npy"))
        self.num_person = self.shape.shape[0]
        self.poses = np.load(os.path.join(root, "poses.npy"))[idx]
        self.trans = np.load(os.path.join(root, "normalize_trans.npy"))[idx]
        cams = np.load(os.path.join(root, "cameras_normalize.npz"))
        scale_mats = [cams["scale_mat_%d" % i].astype(np.float32) for i in idx]
        world_mats = [cams["world_mat_%d" % i].astype(np.float32) for i in idx]
        self.scale = 1 / scale_mats[0][0, 0]
        self.P, self.C, self.intrinsics_all, self.pose_all = [], [], [], []
        for s, w in zip(scale_mats, world_mats):
            P = w @ s
            self.P.append(P)
            self.C.append(-np.linalg.solve(P[:3, :3], P[:3, 3]))
            K, pose = load_K_Rt_from_P(P[:3, :4])
            self.intrinsics_all.append(K.astype(np.float32))
            self.pose_all.append(pose.astype(np.float32))
        self.num_sample = num_sample

    def __len__(self):
        return len(self.img_paths)

    def frame(self, i):
        img = read_png(self.img_paths[i]) / 255             # Hi4D.py:229-232 (BGR -> RGB -> [0,1], float64)
        mask = np.sum(np.stack([bgr2gray(read_png(p[i])) > 0 for p in self.mask_paths], axis=-1), axis=-1)
        return img, mask

    def __getitem__(self, i, rng=np.random):
        img, mask = self.frame(i)
        H, W = self.img_size
        uv = np.mgrid[:H, :W].astype(np.int32)
        uv = np.flip(uv, axis=0).copy().transpose(1, 2, 0).astype(np.float32)
        smpl_params = np.zeros((self.num_person, 86), dtype=np.float32)
        smpl_params[:, 0] = self.scale
        smpl_params[:, 1:4] = self.trans[i]
        smpl_params[:, 4:76] = self.poses[i]
        smpl_params[:, 76:] = self.shape
        if self.num_sample > 0:
            samples, index_outside = weighted_sampling({"rgb": img, "uv": uv, "object_mask": mask}, (H, W), self.num_sample, rng)
            inputs = {"uv": samples["uv"].astype(np.float32), "P": self.P[i], "C": self.C[i],
                      "intrinsics": self.intrinsics_all[i], "pose": self.pose_all[i], "smpl_params": smpl_params,
                      "index_outside": index_outside, "idx": i, "img_size": (H, W)}
            return inputs, {"rgb": samples["rgb"].astype(np.float32)}
        inputs = {"uv": uv.reshape(-1, 2).astype(np.float32), "P": self.P[i], "C": self.C[i],
                  "intrinsics": self.intrinsics_all[i], "pose": self.pose_all[i], "smpl_params": smpl_params, "idx": i,
                  "org_object_mask": mask, "img_size": (H, W)}
        return inputs, {"rgb": img.reshape(-1, 3).astype(np.float32), "img_size": (H, W)}


def novel_view_camera(scale_mat, world_mat, gt_intr_cur, gt_extr_cur, gt_intr_tgt, gt_extr_tgt):
    """Hi4DTestDataset.__init__'s per-frame body (Hi4D.py:398-425), step for step: the training description of the current
    studio camera (R3, t3) and its ground-truth description (R1, t1) give the studio -> training rigid motion (Rab, tab);
    the target studio camera (R2, t2) seen from the training frame is (R4, t4); its intrinsics are divided by the focal
    ratio of the two descriptions of the current camera.  -> P (4,4), C (3,), intrinsics (4,4), pose (4,4)."""
    intr_tr, pose_tr = load_K_Rt_from_P(np.asarray(world_mat)[:3, :4])
    scale_factor = gt_intr_cur[0, 0] / intr_tr[0, 0]
    R3 = pose_tr[:3, :3].transpose()
    t3 = -R3 @ pose_tr[:3, 3]
    R1, t1 = gt_extr_cur[:3, :3], gt_extr_cur[:3, 3]
    Rab = R3.transpose() @ R1
    tab = R3.transpose() @ (t1 - t3)
    R2, t2 = gt_extr_tgt[:3, :3], gt_extr_tgt[:3, 3]
    R4 = R2 @ Rab.transpose()
    t4 = t2 - R4 @ tab
    Kt = gt_intr_tgt[:3, :3].copy()
    for (i, j) in ((0, 0), (1, 1), (0, 2), (1, 2)):
        Kt[i, j] = Kt[i, j] / scale_factor
    novel = np.eye(4)
    novel[:3, :4] = Kt @ np.concatenate((R4, t4.reshape(3, 1)), axis=1)
    P = novel @ scale_mat
    C = -np.linalg.solve(P[:3, :3], P[:3, 3])
    intr, pose = load_K_Rt_from_P(P[:3, :4])
    return P, C, intr, pose
