#!/usr/bin/env python
"""Which Python call sites of a training iteration create zero-fill / copy launches?  Wraps the torch entry points that fill or
copy a device tensor and histograms their callers (first frame outside torch) over one iteration of bench.py's training leg.
    python tools/count_fills.py"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import bench   # noqa: E402

hist = collections.Counter()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/torch/" not in fr.filename and "count_fills" not in fr.filename:
            return f"{os.path.relpath(fr.filename)}:{fr.lineno}"
    return "?"


def wrap_fn(mod, name, kind, cond=lambda *a, **k: True):
    orig = getattr(mod, name)

    def f(*a, **k):
        if ON[0] and cond(*a, **k):
            hist[(kind, name, site())] += 1
        return orig(*a, **k)
    setattr(mod, name, f)


is_dev = lambda *a, **k: (str(k.get("device", "")).startswith("cuda")) or any(torch.is_tensor(x) and x.is_cuda for x in a)
for n in ("zeros", "full", "ones"):
    wrap_fn(torch, n, "fill", is_dev)
for n in ("zeros_like", "full_like", "ones_like"):
    wrap_fn(torch, n, "fill", is_dev)
for n in ("zero_", "fill_", "new_zeros", "new_full"):
    wrap_fn(torch.Tensor, n, "fill", lambda s, *a, **k: s.is_cuda)
wrap_fn(torch.Tensor, "clone", "copy", lambda s, *a, **k: s.is_cuda)
wrap_fn(torch.Tensor, "copy_", "copy", lambda s, *a, **k: s.is_cuda)
wrap_fn(torch.Tensor, "contiguous", "copy", lambda s, *a, **k: s.is_cuda and not s.is_contiguous())
wrap_fn(torch.Tensor, "float", "copy", lambda s, *a, **k: s.is_cuda and s.dtype != torch.float32)
wrap_fn(torch.Tensor, "to", "copy", lambda s, *a, **k: s.is_cuda)
wrap_fn(torch.Tensor, "index_select", "copy", lambda s, *a, **k: s.is_cuda)
wrap_fn(torch.Tensor, "__setitem__", "copy", lambda s, *a, **k: s.is_cuda)
wrap_fn(torch, "cat", "copy")

from multiply_amd.config import load_config   # noqa: E402
from multiply_amd.loss import Loss            # noqa: E402
model, inp, tables, sc = bench.build_model(128, seed=0)
model.convergence_group = 512
gin = bench.to_dev(inp)
model.train()
model.async_setup = True
loss_fn = Loss(load_config().loss)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
g = torch.Generator().manual_seed(0)
R = gin["uv"].shape[1]
for it in range(4):
    sel = torch.randperm(R, generator=g)[:512].cuda()
    tin = dict(gin); tin["uv"] = gin["uv"][:, sel].contiguous()
    tin.update(current_epoch=301, index_outside=torch.zeros(512, dtype=torch.bool, device="cuda"), smpl_pose_last=gin["smpl_pose"] + 0.01)
    gt = {"rgb": torch.rand(1, 512, 3, generator=g).cuda()}
    torch.cuda.synchronize()
    ON[0] = it == 3
    out = model(tin)
    lo = loss_fn(out, gt)
    opt.zero_grad(set_to_none=True)
    lo["loss"].backward()
    opt.step()
    ON[0] = False
torch.cuda.synchronize()
for kind in ("fill", "copy"):
    rows = [(k, v) for k, v in hist.items() if k[0] == kind]
    print(f"== {kind}: {sum(v for _, v in rows)} calls in one iteration")
    for (k, name, s), v in sorted(rows, key=lambda kv: -kv[1]):
        print(f"  {v:3d}  {name:12s} {s}")
