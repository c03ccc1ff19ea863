#!/usr/bin/env python
"""GPU box, library built with -DMP_GEOM_PROF (tools/ab_build.sh geomprof:-DMP_GEOM_PROF) and MP_LIB_PATH pointing at it: one
training iteration's k_warp_inverse / k_warp_jacobian waves -- slabs, cycles per slab, clusters opened."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import bench   # noqa: E402
from tools.geom_prof import read   # noqa: E402
from multiply_amd.config import load_config   # noqa: E402
from multiply_amd.loss import Loss   # noqa: E402

model, inp, tables, sc = bench.build_model(128, seed=0)
gin = bench.to_dev(inp)
model.train()
loss_fn = Loss(load_config().loss)
g = torch.Generator().manual_seed(0)
R = gin["uv"].shape[1]
for it in range(3):
    sel = torch.randperm(R, generator=g)[:512].cuda()
    tin = dict(gin); tin["uv"] = gin["uv"][:, sel].contiguous()
    tin.update(current_epoch=301, index_outside=torch.zeros(512, dtype=torch.bool, device="cuda"), smpl_pose_last=gin["smpl_pose"] + 0.01)
    gt = {"rgb": torch.rand(1, 512, 3, generator=g).cuda()}
    read()
    out = model(tin)
    v = read()
    per = lambda i: v[i] / max(v[0], 1)
    print(f"forward: slabs {v[0]:7d}  cycles/slab {per(1):8.0f}  load {per(6):6.0f}  cull {per(2):6.0f}  scan {per(3):7.0f}  "
          f"epilogue {per(7):6.0f}  clusters boxed {per(4):5.2f} scanned {per(5):5.2f}")
    loss_fn(out, gt)["loss"].backward()
