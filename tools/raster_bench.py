#!/usr/bin/env python
"""Per-person mesh z-buffer at the Hi4D frame size: SMPL-sized and canonical-mesh-sized inputs (run under
`rocprofv3 --kernel-trace` + tools/rocpd_summary.py for the per-kernel table; prints host-side times and the algorithmic
bytes: faces 12 B + 3 gathered vertices 36 B per face, 8 B key cleared + 8 B read + 24 B written per pixel)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd.render import Renderer                      # noqa: E402
from tests.test_raster_cpu import uv_sphere                    # noqa: E402

H, W = 940, 1280
K = np.array([[1400.0, 0, 640.3], [0, 1400.0, 469.8], [0, 0, 1.0]])
r = Renderer(img_size=[H, W], cam_intrinsic=K)
r.set_camera(torch.tensor(np.diag([1.0, -1.0, -1.0]))[None].float(), torch.tensor([[0.0, 0.0, 3.0]]))
for name, (nlat, nlon) in (("13.8 k faces (SMPL size)", (60, 116)), ("65 k faces", (128, 256)), ("261 k faces", (256, 512))):
    v, f = uv_sphere([0.013, -0.007, 0.0], 0.5, nlat, nlon)
    vt, ft = torch.tensor(v).float().cuda(), torch.tensor(f).cuda()
    for _ in range(3):
        r.rasterize(vt, ft)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        frag = r.rasterize(vt, ft)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    cover = float((frag.zbuf > 0).float().mean())
    alg = f.shape[0] * 48 + H * W * 40
    print(f"{name:26s} {f.shape[0]:7d} faces  {ms:6.3f} ms / mesh (host-timed)   covered {100 * cover:4.1f} % of the frame   "
          f"algorithmic {alg / 1e6:5.1f} MB -> {alg / ms / 1e6:6.1f} GB/s")

# the soft silhouette render of a three-body scene (render.softrender_multiple_meshes: sigma 5e-5, 100 faces per pixel)
print()
for name, (nlat, nlon) in (("3 x 13.8 k faces", (60, 116)), ("3 x 65 k faces", (128, 256))):
    vl, fl, cl = [], [], []
    for p, cx in enumerate((-0.55, 0.0, 0.5)):
        v, f = uv_sphere([cx, -0.007, 0.15 * p], 0.38, nlat, nlon)
        vl.append(torch.tensor(v).float().cuda()[None]); fl.append(torch.tensor(f).cuda()[None])
        cl.append(torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]][p]).repeat(v.shape[0], 1).cuda()[None])
    with torch.no_grad():
        for _ in range(2):
            img = r.softrender_multiple_meshes(vl, fl, cl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            img = r.softrender_multiple_meshes(vl, fl, cl)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    a = img[0, :, :, 3]
    print(f"soft render {name:18s} {ms:8.2f} ms / scene (host-timed, incl. the one host read)   silhouette > 0.5 on "
          f"{100 * float((a > 0.5).float().mean()):4.1f} % of the frame, partial (0.01 .. 0.99) on {100 * float(((a > 0.01) & (a < 0.99)).float().mean()):4.2f} %")
vg = [v.clone().requires_grad_(True) for v in vl]
torch.cuda.synchronize()
t0 = time.perf_counter()
img = r.softrender_multiple_meshes(vg, fl, cl)
(img[0, :, :, 3].mean()).backward()
torch.cuda.synchronize()
print(f"soft render with gradient (3 x 65 k faces): {1e3 * (time.perf_counter() - t0):8.1f} ms forward + backward, "
      f"|d mean(alpha) / d verts| max {max(float(v.grad.abs().max()) for v in vg):.3e}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
