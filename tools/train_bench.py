#!/usr/bin/env python
"""Training iterations only (bench.py's train_iter leg): ms/iter, GPU phase times; run it under
`rocprofv3 --kernel-trace` + tools/rocpd_summary.py for the per-kernel table of ONE iteration's work.
    python tools/train_bench.py [steps] [warmup]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import bench   # noqa: E402

steps = int(args[0]) if args else 20
warm = int(args[1]) if len(args) > 1 else 3
model, inp, tables, sc = bench.build_model(128, seed=0)
model.convergence_group = 512
gin = bench.to_dev(inp)


def barrier():
    torch.cuda.synchronize()


dt, ph, loss, stats, host_ms = bench.train_iterations(model, gin, steps, warm, False, barrier, seed=0, rays=512)
print(json.dumps({"ms_per_iter": 1e3 * dt / steps, "host_ms_per_iter": host_ms, "gpu_ms": {"forward+loss": ph[0], "backward": ph[1], "allreduce": ph[2],
                                                               "adam": ph[3]}, "hit_rays": stats["n_hit"], "loss": loss}))

# ---- host-side view: how long does the HOST need to enqueue each part (it runs ahead of the GPU unless something syncs)?
import time
from multiply_amd.config import load_config
from multiply_amd.loss import Loss
model.train()
loss_fn = Loss(load_config().loss)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
g = torch.Generator().manual_seed(0)
R = gin["uv"].shape[1]
acc = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0, "adam": 0.0, "sync": 0.0}
model.async_setup = os.environ.get('MP_ASYNC_SETUP', '1') == '1'
n = 10
for it in range(n + 2):
    sel = torch.randperm(R, generator=g)[:512].cuda()
    tin = dict(gin); tin["uv"] = gin["uv"][:, sel].contiguous()
    tin.update(current_epoch=301, index_outside=torch.zeros(512, dtype=torch.bool, device="cuda"), smpl_pose_last=gin["smpl_pose"] + 0.01)
    gt = {"rgb": torch.rand(1, 512, 3, generator=g).cuda()}
    torch.cuda.synchronize()
    t0 = time.perf_counter(); out = model(tin)
    t1 = time.perf_counter(); lo = loss_fn(out, gt)
    t2 = time.perf_counter(); opt.zero_grad(set_to_none=True); lo["loss"].backward()
    t3 = time.perf_counter(); opt.step()
    t4 = time.perf_counter(); torch.cuda.synchronize()
    t5 = time.perf_counter()
    if it >= 2:
        for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += 1e3 * d / n
print("host ms per iteration until each call RETURNS (then the final synchronize):", {k: round(v, 2) for k, v in acc.items()})
