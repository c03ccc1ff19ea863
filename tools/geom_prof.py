#!/usr/bin/env python
"""GPU box, with a library built with -DMP_GEOM_PROF (tools/ab_build.sh geomprof:-DMP_GEOM_PROF) and MP_LIB_PATH pointing at
it: renders one frame and prints where the waves of k_warp_inverse / k_warp_jacobian spend their cycles, per phase of the
frame (sampler iterations, shading warp, shading Jacobian)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import hip                     # noqa: E402
from tests.test_render_gpu import build          # noqa: E402

NAMES = ["slabs", "cycles", "cull_cycles", "scan_cycles", "clusters_boxed", "clusters_scanned", "load_cycles", "epilogue_cycles"]


def read(reset=True):
    buf = (C.c_ulonglong * 16)()
    fn = hip.lib().mp_geom_prof_read
    fn.argtypes, fn.restype = [C.c_void_p, C.c_int], C.c_int
    torch.cuda.synchronize()
    assert fn(C.cast(buf, C.c_void_p), int(reset)) == 0
    return list(buf)


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    model, _, inp = build(H=res, W=res)
    model.ray_sampler.N_samples = 128
    gin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    with torch.no_grad():
        model(gin)
        read()
        marks = {}
        orig = model._ph

        class Ph:
            def __init__(self, name): self.name = name
            def __enter__(self): read(); return self
            def __exit__(self, *a):
                v = read()
                acc = marks.setdefault(self.name, [0] * 16)
                for i in range(16): acc[i] += v[i]
        model._ph = lambda name: Ph(name)
        model(gin)
        model._ph = orig
    for ph, v in marks.items():
        if v[0] == 0:
            continue
        per = lambda i: v[i] / v[0]
        print(f"{ph:18s} slabs {v[0]:8d}  cycles/slab {per(1):8.0f}  load {per(6):6.0f}  cull {per(2):6.0f}  scan {per(3):7.0f}  "
              f"epilogue {per(7):6.0f}  clusters boxed {per(4):5.2f} scanned {per(5):5.2f}")


if __name__ == "__main__":
    main()
