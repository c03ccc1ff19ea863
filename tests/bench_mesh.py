#!/usr/bin/env python
"""(Lives under tests/ because it runs the reference extractor built into oracle/_ref as the baseline.)
Canonical-mesh extraction: multiply_amd.mesh (dense device lattice, whole-pass network queries, marching-cubes kernels)
vs the reference's structure -- its own CPU octree extractor (oracle/_ref/mise*.so, built from code/lib/libmise/mise.pyx
by `make -C oracle`) fed through 10 000-point network batches with a host round trip per batch
(code/lib/utils/mesh.py:88-109; the marching cubes that follows there is skimage's and is not timed).
    python tests/bench_mesh.py [res_up ...]      (default 2 4: the trainer's refresh and the validation setting)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
from multiply_amd import hip                                     # noqa: E402
from multiply_amd.mesh import generate_mesh, lattice_to_world    # noqa: E402
from tests.test_render_gpu import build                           # noqa: E402

try:
    import mise as ref_mise
except ImportError:
    ref_mise = None

model, _, _ = build(H=9, W=9)
imp = model.foreground_implicit_network_list[0]
cond = torch.zeros(69, device="cuda")
vc = model.smpl_server_list[0].verts_c[0]
func = lambda x: hip.implicit_sdf(imp, x.contiguous(), cond)

for res_up in [int(a) for a in sys.argv[1:]] or [2, 4]:
    generate_mesh(func, vc, 0.0, 32, min(res_up, 2))             # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = generate_mesh(func, vc, 0.0, 32, res_up)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    line = (f"res_up {res_up} (lattice {m['resolution'] + 1}^3): device path {t_dev * 1e3:.0f} ms, {sum(m['n_queried'])} network "
            f"queries in {len(m['n_queried'])} passes, {m['vertices'].shape[0]} vertices / {m['faces'].shape[0]} faces")
    if ref_mise is not None:
        v = vc.float()
        lo, hi = v.min(0).values, v.max(0).values
        center, scale = (lo + hi) * 0.5, (hi - lo).max()
        t0 = time.perf_counter()
        ex = ref_mise.MISE(32, res_up, 0.0)
        pts = ex.query()
        while pts.shape[0]:
            world = lattice_to_world(torch.from_numpy(pts).cuda(), ex.resolution, scale, center)
            vals = [func(c).cpu().numpy() for c in torch.split(world, 10000, dim=0)]
            ex.update(pts, np.concatenate(vals).astype(np.float64))
            pts = ex.query()
        dense = ex.to_dense()
        t_ref = time.perf_counter() - t0
        same = np.array_equal(dense, m["value_grid"].cpu().numpy().astype(np.float64))
        line += f"; reference structure (CPU octree + 10k batches, no surface extraction) {t_ref * 1e3:.0f} ms, x{t_ref / t_dev:.1f}; same value grid: {same}"
    print(line)
