#!/usr/bin/env python
"""Generates tests/golden/dataset_golden.npz from the REFERENCE's own input-producer functions
(code/lib/datasets/Hi4D.py: bilinear_interpolation, get_index_outside_of_bbox, weighted_sampling), imported from
/root/reference with the third-party modules this image lacks (cv2, hydra, imageio, skimage, trimesh) stubbed out -- the three functions are pure
numpy.  Run in the build container only:  python tests/golden/make_dataset_golden.py"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/code"
for name in ("cv2", "hydra", "hydra.utils", "imageio", "skimage", "trimesh"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, REF)
from lib.datasets import Hi4D as R      # noqa: E402

rs = np.random.RandomState(7)
H, W, P = 40, 48, 2
img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
person = np.zeros((P, H, W), dtype=bool)
person[0, 8:30, 10:26] = True
person[1, 12:34, 22:40] = True
mask = person.sum(0)                                   # Hi4D.py:244-245: sum of the per-person boolean masks
uv = np.mgrid[:H, :W].astype(np.int32)
uv = np.flip(uv, axis=0).copy().transpose(1, 2, 0).astype(np.float32)
sam = rs.rand(H, W, P).astype(np.float32)
data = {"rgb": img[:, :, ::-1][:, :, ::-1] / 255, "uv": uv, "object_mask": mask, "sam_mask": sam}
n = 64
np.random.seed(123)
state = np.random.get_state()
out, index_outside = R.weighted_sampling(data, (H, W), n)
# the same positions, re-derived from the same RNG state with the reference's own helper functions
np.random.set_state(state)
where = np.asarray(np.where(mask))
bmin, bmax = where.min(axis=1), where.max(axis=1)
nb = int(n * 0.9)
pos_b = np.random.rand(nb, 2) * (bmax - bmin) + bmin
pos_u = np.random.rand(n - nb, 2) * (H - 1, W - 1)
pos = np.concatenate([pos_b, pos_u], 0)
assert np.array_equal(R.get_index_outside_of_bbox(pos_u, bmin, bmax) + nb, index_outside)
assert np.allclose(R.bilinear_interpolation(pos[:, 0], pos[:, 1], mask), out["object_mask"])
# edge_sampling (Hi4D.py:28-56): integer pixel picks
edge = np.zeros((H, W), dtype=bool)
edge[8:30, 10] = edge[8:30, 25] = edge[12:34, 22] = edge[12:34, 39] = True
np.random.seed(321)
eout = R.edge_sampling({"rgb": data["rgb"], "uv": uv, "person_mask": mask, "edge_mask": np.logical_and(mask, edge)}, 40)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dataset_golden.npz"),
                    img=img, person=person, sam=sam, n=n, seed=123, pos=pos, index_outside=index_outside,
                    rgb=out["rgb"], uv=out["uv"], object_mask=out["object_mask"], sam_mask=out["sam_mask"],
                    edge=edge, edge_seed=321, edge_n=40, edge_uv=eout["uv"], edge_rgb=eout["rgb"])
print("wrote dataset_golden.npz:", {k: v.shape for k, v in out.items()}, "outside", index_outside)
