#!/usr/bin/env python
"""Host-side profile (cProfile) of training iterations: where the Python time of one iteration goes."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import bench   # noqa: E402
from multiply_amd.config import load_config   # noqa: E402
from multiply_amd.loss import Loss   # noqa: E402

model, inp, tables, sc = bench.build_model(128, seed=0)
gin = bench.to_dev(inp)
model.train()
model.async_setup = True
loss_fn = Loss(load_config().loss)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
g = torch.Generator().manual_seed(0)
R = gin["uv"].shape[1]
batches = []
for _ in range(8):
    sel = torch.randperm(R, generator=g)[:512].cuda()
    tin = dict(gin); tin["uv"] = gin["uv"][:, sel].contiguous()
    tin.update(current_epoch=301, index_outside=torch.zeros(512, dtype=torch.bool, device="cuda"), smpl_pose_last=gin["smpl_pose"] + 0.01)
    batches.append((tin, {"rgb": torch.rand(1, 512, 3, generator=g).cuda()}))


def step(i):
    tin, gt = batches[i]
    out = model(tin)
    lo = loss_fn(out, gt)
    opt.zero_grad(set_to_none=True)
    lo["loss"].backward()
    opt.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(3, 8):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
