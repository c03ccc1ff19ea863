"""Prompt construction and the refinement loop around Segment-Anything that the trainer runs every 50 epochs
(code/lib/model/sam_model.py:35-239): from the rasterised instance masks (mesh_losses.frame_instance_masks) and the projected
key points of every person to SAM's box / point / mask prompts, three predictor rounds feeding the mask logits back, and the
`stage_sam_mask/<epoch>/sam_opt_mask.npy` file the datasets read (datasets.Hi4DDataset._sam_mask).

SAM itself (segment_anything's ViT-H, a third-party model and checkpoint) is NOT part of this repository: `SAMServer` takes any
object with SamPredictor's `set_image(rgb_uint8)` / `predict(point_coords, point_labels, mask_input, box, multimask_output,
return_logits)` interface and only builds one from `segment_anything` when that package and the checkpoint are present.
Everything here is host-side numpy on a handful of points per person; the random draws follow the reference's stream
(np.random.seed(42) once per refresh; per person: fallback positives, one permutation, the negative points), so the prompts
are reproducible against it.  cv2.resize (absent here) is restated for the one use made of it: a bilinear 8-bit downscale of
a binary mask, which is a threshold of the interpolated coverage at one half -- parity with OpenCV's fixed-point rounding
on exact ties is unpinned."""
import glob
import os

import numpy as np

N_KEYPOINTS = 27          # 24 joints + nose + eyes (multiply_model.py:858)
N_NEGATIVE = 10


def box_from_mask(mask):
    """xyxy box of a binary mask, widened by 3 % per side in the reference's update order (sam_model.py:79-93: the right /
    bottom margins are computed from the already widened left / top)"""
    rows, cols = np.nonzero(mask)
    x0, x1, y0, y1 = cols.min(), cols.max(), rows.min(), rows.max()
    x0 = max(0, x0 - int(0.03 * (x1 - x0)))
    y0 = max(0, y0 - int(0.03 * (y1 - y0)))
    x1 = min(mask.shape[1], x1 + int(0.03 * (x1 - x0)))
    y1 = min(mask.shape[0], y1 + int(0.03 * (y1 - y0)))
    return np.array([x0, y0, x1, y1])


def resize_binary_256(canvas):
    """cv2.resize(uint8 canvas, (256, 256)) with the default bilinear filter, for 0 / 1 images: sample positions
    (i + 0.5) * scale - 0.5 clamped to the image, the interpolated value rounded to the nearest integer"""
    n = canvas.shape[0]
    pos = (np.arange(256) + 0.5) * (n / 256.0) - 0.5
    lo = np.floor(pos).astype(np.int64)
    frac = pos - lo
    a, b = np.clip(lo, 0, n - 1), np.clip(lo + 1, 0, n - 1)
    c = canvas.astype(np.float64)
    rows = c[a] * (1 - frac)[:, None] + c[b] * frac[:, None]
    out = rows[:, a] * (1 - frac)[None, :] + rows[:, b] * frac[None, :]
    return np.floor(out + 0.5).astype(np.uint8)


def mask_prompt(mask, eps=1e-6):
    """SAM's low-resolution mask input from an instance mask (sam_model.py:95-113, :203-204): pad to a square (rows kept at
    the top; columns right-aligned when the image is wider than tall), 256 x 256, logit with clamping"""
    h, w = mask.shape
    n = max(h, w)
    canvas = np.zeros((n, n), dtype=np.uint8)
    if h > w:
        canvas[:h, :w] = mask
    else:
        canvas[:h, n - w:] = mask
    p = np.clip(resize_binary_256(canvas).astype(np.float32), eps, 1 - eps)
    return np.log(p / (1 - p))[None]


def _inside(mask, pt):
    """mask[y, x] with numpy's own index rules (negative indices wrap, out of range -> None), as the reference's try / except
    around the lookup behaves"""
    try:
        return mask[pt[1], pt[0]]
    except IndexError:
        return None


def point_prompts(masks, joints, person, rng):
    """Positive / negative point prompts of `person` (sam_model.py:115-196).  masks (P, H, W) instance masks, joints
    (P, >= 27, 2) integer pixel positions (x, y).  -> coords (n, 2), labels (n,)"""
    own = masks[person]
    others = [q for q in range(masks.shape[0]) if q != person]
    other_any = np.max(masks[others], axis=0)
    pos = [p for p in joints[person, :N_KEYPOINTS] if (v := _inside(own, p)) is not None and v > 0.7]
    pos = np.array(pos)
    if len(pos) == 0:                                     # no key point on the mask: one random pixel of it
        picked = []
        for _ in range(10000000):
            x = rng.randint(0, own.shape[1])
            y = rng.randint(0, own.shape[0])
            if own[y, x] > 0.7:
                picked.append([x, y])
                break
        if not picked:
            picked.append(joints[person, N_KEYPOINTS - 1])
        pos = np.array(picked)
    order = rng.choice(len(pos), len(pos), replace=False)
    pos = pos[order]
    neg = []
    while len(neg) < N_NEGATIVE:                          # background or other persons: anywhere off the own mask
        x = rng.randint(0, own.shape[1])
        y = rng.randint(0, own.shape[0])
        if own[y, x] == 0:
            neg.append([x, y])
    for q in others:                                      # the other persons' key points that lie on THEIR masks, off this one
        for p in joints[q, :N_KEYPOINTS]:
            v = _inside(own, p)
            if v is not None and v < 0.7 and other_any[p[1], p[0]] > 0.7:
                neg.append([p[0], p[1]])
    neg = np.array(neg)
    return np.concatenate((pos, neg), axis=0), np.concatenate((np.ones(len(pos)), np.zeros(len(neg))))


def refine(predictor, coords, labels, box, mask_logit, rounds=3):
    """sam_model.py:205-231: `rounds` predictions, each fed the previous round's low-resolution logits -> mask logits (1, H, W)"""
    masks = None
    for _ in range(rounds):
        masks, _, mask_logit = predictor.predict(point_coords=coords, point_labels=labels, mask_input=mask_logit,
                                                 box=box[None, :], multimask_output=False, return_logits=True)
    return masks


def frame_masks(predictor, image_rgb, masks, joints, rng):
    """all persons of one frame -> (P, H, W) refined mask logits"""
    predictor.set_image(image_rgb)
    out = []
    for person in range(masks.shape[0]):
        coords, labels = point_prompts(masks, joints, person, rng)
        out.append(refine(predictor, coords, labels, box_from_mask(masks[person]), mask_prompt(masks[person])))
    return np.concatenate(out, axis=0)


class SAMServer:
    """sam_model.py:35-56, 58-239.  opt: data_dir, start_frame, end_frame (and data_root, default '../data')."""

    def __init__(self, opt, predictor=None):
        root = os.path.abspath(os.path.join(opt.get("data_root", "../data"), opt.data_dir))
        paths = sorted(glob.glob(os.path.join(root, "image", "*.png")))
        self.training_indices = list(range(opt.start_frame, opt.end_frame, 1))
        self.img_paths = [paths[i] for i in self.training_indices]
        self.opt = opt
        if predictor is None:
            try:
                from segment_anything import SamPredictor, sam_model_registry
            except ImportError as e:
                raise RuntimeError("segment_anything is not installed: pass predictor= (any object with SamPredictor's "
                                   "set_image / predict)") from e
            sam = sam_model_registry["vit_h"](checkpoint=os.path.abspath("./outputs/sam_vit_h_4b8939.pth"))
            sam.to(device="cuda")
            predictor = SamPredictor(sam)
        self.predictor = predictor

    def get_sam_mask(self, current_epoch, stage_dir="."):
        from .datasets import read_png_rgb
        rng = np.random.RandomState(42)
        src = os.path.join(stage_dir, f"stage_instance_mask/{current_epoch:05d}")
        smpl_mask = np.load(os.path.join(src, "all_person_smpl_mask.npy"))
        smpl_joint = np.load(os.path.join(src, "2d_keypoint.npy"))
        frames = [frame_masks(self.predictor, read_png_rgb(path), smpl_mask[i], smpl_joint[i], rng)
                  for i, path in enumerate(self.img_paths)]
        out = np.stack(frames, axis=0)
        dst = os.path.join(stage_dir, f"stage_sam_mask/{current_epoch:05d}")
        os.makedirs(dst, exist_ok=True)
        np.save(os.path.join(dst, "sam_opt_mask.npy"), out)
        return out
