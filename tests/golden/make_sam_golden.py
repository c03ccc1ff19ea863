#!/usr/bin/env python
"""Generates tests/golden/sam_golden.npz by RUNNING the reference's SAMServer.get_sam_mask (code/lib/model/sam_model.py,
imported from /root/reference) on written sequences, with its three absent third-party imports stubbed:

  * segment_anything -> tests/sam_standin.StandInPredictor, which records every prompt the reference hands to predict()
  * hydra            -> utils.to_absolute_path = os.path.abspath
  * cv2              -> imread / cvtColor through PIL; resize(src, (256, 256)) through torch's bilinear interpolation
                        (half-pixel sample positions like INTER_LINEAR), rounded to nearest.  OpenCV's own fixed-point
                        rounding is therefore NOT pinned by this fixture; everything else the reference computes is.

The fixture holds the inputs (images, instance masks, key points) and, per predict() call in the reference's order, the point
prompts, labels, box and first-round mask input, plus the sam_opt_mask.npy the reference wrote.
Run in the build container only:  python tests/golden/make_sam_golden.py"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from sam_standin import StandInPredictor      # noqa: E402

predictors = []


def make_predictor(_sam):
    predictors.append(StandInPredictor())
    return predictors[-1]


class _Sam:
    def to(self, device=None):
        return self


def _resize(src, size):
    assert size == (256, 256) and src.dtype == np.uint8 and src.ndim == 2
    t = torch.from_numpy(src.astype(np.float64))[None, None]
    out = torch.nn.functional.interpolate(t, size=size, mode="bilinear", align_corners=False)[0, 0].numpy()
    return np.floor(out + 0.5).astype(np.uint8)


sys.modules["hydra"] = types.SimpleNamespace(utils=types.SimpleNamespace(to_absolute_path=os.path.abspath))
sys.modules["segment_anything"] = types.SimpleNamespace(sam_model_registry={"vit_h": lambda checkpoint=None: _Sam()},
                                                        SamPredictor=make_predictor)
sys.modules["cv2"] = types.SimpleNamespace(
    COLOR_BGR2RGB=4, imread=lambda p: np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1].copy(),
    cvtColor=lambda img, code: img[:, :, ::-1].copy(), resize=_resize)
sys.path.insert(0, "/root/reference/code")
from lib.model.sam_model import SAMServer      # noqa: E402  (the reference)


def scene(rs, F, P, H, W, n_joints=27):
    yy, xx = np.mgrid[:H, :W]
    masks = np.zeros((F, P, H, W), dtype=bool)
    joints = np.zeros((F, P, n_joints, 2), dtype=np.int32)
    images = (rs.rand(F, H, W, 3) * 255).astype(np.uint8)
    for f in range(F):
        for p in range(P):
            cx, cy = W * (0.5 + 0.52 * (p - (P - 1) / 2) / P) + 2 * f, H * (0.5 + 0.05 * p)
            blob = (((xx - cx) / (0.15 * W)) ** 2 + ((yy - cy) / (0.36 * H)) ** 2) < 1
            masks[f, p] = blob & ~masks[f, :p].any(axis=0)            # instance masks are disjoint (front-most person wins)
            # key points: most on the body, some on the neighbours, some off the image on either side (negative indices wrap
            # in the reference's lookup, indices past the end raise and are skipped)
            joints[f, p, :, 0] = np.clip(rs.normal(cx, 0.16 * W, n_joints), -6, W + 5).astype(np.int32)
            joints[f, p, :, 1] = np.clip(rs.normal(cy, 0.3 * H, n_joints), -6, H + 5).astype(np.int32)
    return images, masks, joints


def run_reference(images, masks, joints, start, end, epoch):
    F = images.shape[0]
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "data", "seq", "image"))
        os.makedirs(os.path.join(tmp, "code", "stage_instance_mask", f"{epoch:05d}"))
        for f in range(F):
            Image.fromarray(images[f]).save(os.path.join(tmp, "data", "seq", "image", "%04d.png" % f))
        cwd = os.getcwd()
        os.chdir(os.path.join(tmp, "code"))
        try:
            np.save(f"stage_instance_mask/{epoch:05d}/all_person_smpl_mask.npy", masks[start:end])
            np.save(f"stage_instance_mask/{epoch:05d}/2d_keypoint.npy", joints[start:end])
            server = SAMServer(types.SimpleNamespace(data_dir="seq", start_frame=start, end_frame=end))
            server.get_sam_mask(epoch)
            written = np.load(f"stage_sam_mask/{epoch:05d}/sam_opt_mask.npy")
        finally:
            os.chdir(cwd)
    return predictors[-1].calls, written


out = {}
rs = np.random.RandomState(7)
cases = dict(wide=dict(F=3, P=3, H=48, W=64, start=1, end=3), tall=dict(F=1, P=2, H=72, W=40, start=0, end=1),
             square=dict(F=2, P=4, H=56, W=56, start=0, end=2))
for name, c in cases.items():
    images, masks, joints = scene(rs, c["F"], c["P"], c["H"], c["W"])
    if name == "wide":
        joints[2, 1, :, :] = [0, 0]                     # no key point of person 1 on its mask: the random fallback pixel
    if name == "square":
        joints[1, 2, :, 0] = masks.shape[3] + 3         # every lookup of person 2 raises: fallback as well
    calls, written = run_reference(images, masks, joints, c["start"], c["end"], 50)
    n_person_calls = (c["end"] - c["start"]) * c["P"]
    assert len(calls) == 3 * n_person_calls
    out[f"{name}_images"], out[f"{name}_masks"], out[f"{name}_joints"] = images, masks, joints
    out[f"{name}_range"] = np.array([c["start"], c["end"]])
    out[f"{name}_written"] = written.astype(np.float32)
    assert written.dtype == np.float32
    for k in range(n_person_calls):
        first = calls[3 * k]
        for r in (1, 2):                                 # the three rounds of a person share the prompts
            assert all(np.array_equal(first[q], calls[3 * k + r][q]) for q in ("coords", "labels", "box"))
        out[f"{name}_coords_{k}"], out[f"{name}_labels_{k}"], out[f"{name}_box_{k}"] = first["coords"], first["labels"], first["box"]
        vals = np.unique(first["mask_input"])
        assert vals.size == 2
        out[f"{name}_mask_on_{k}"] = np.packbits(first["mask_input"][0] > 0)
        out[f"{name}_mask_vals_{k}"] = vals
        out[f"{name}_mask_mean_rounds_{k}"] = np.array([calls[3 * k + r]["mask_input"].astype(np.float64).mean() for r in range(3)])
    print(name, "persons x frames", n_person_calls, "points per call", [int(calls[3 * k]["coords"].shape[0]) for k in range(n_person_calls)])
np.savez_compressed(os.path.join(HERE, "sam_golden.npz"), **out)
print("wrote", os.path.join(HERE, "sam_golden.npz"), os.path.getsize(os.path.join(HERE, "sam_golden.npz")), "bytes")
