#!/usr/bin/env python
"""Micro-benchmark of the fused MLP kernels on random canonical points (no scene):  python tools/mlp_microbench.py [n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import hip            # noqa: E402
from tests.util import seeded_networks  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "all"
m, _ = seeded_networks(2, 0)
m = m.cuda()
imp, ren = m.foreground_implicit_network_list[0], m.foreground_rendering_network_list[0]
x = (torch.rand(n, 3, device="cuda") - 0.5) * 1.6
jinv = torch.eye(3, device="cuda").reshape(1, 9).repeat(n, 1).contiguous()
cond = torch.randn(69, device="cuda") * 0.1


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


M_IMP, M_REN = 542208, 266496
if which in ("all", "shade"):
    t = timeit(lambda: hip.shade_points(imp, ren, x, jinv, cond))
    print(f"shade+color: {n / t / 1e6:.1f} Mpts/s, algorithmic {(4 * M_IMP + 2 * M_REN) * n / t / 1e12:.0f} TFLOP/s  ({t * 1e3:.1f} ms)")
if which in ("all", "sdf"):
    t = timeit(lambda: hip.implicit_sdf(imp, x, cond))
    print(f"sdf only   : {n / t / 1e6:.1f} Mpts/s, algorithmic {2 * M_IMP * n / t / 1e12:.0f} TFLOP/s  ({t * 1e3:.1f} ms)")
if which in ("all", "sdf", "sdf_x2"):
    t = timeit(lambda: hip.implicit_sdf(imp, x, cond, mode="f16x2"))
    print(f"sdf, split activations (mp_mlp_sdf_x2): {n / t / 1e6:.1f} Mpts/s, algorithmic {2 * M_IMP * n / t / 1e12:.0f} TFLOP/s, "
          f"executed {4 * M_IMP * n / t / 1e12:.0f}  ({t * 1e3:.1f} ms)")
if which in ("all", "sdf", "sdf_b3"):
    from multiply_amd import train as T
    fs = T.fused_sdf_state(imp).refresh(cond)
    outb = torch.empty(n, device="cuda")
    t = timeit(lambda: hip.check(hip.lib().mp_tf_sdf_val(hip.ptr(fs.wpack), hip.ptr(fs.bias_all), hip.ptr(x), None, None, n, hip.ptr(outb),
                                                          hip.stream()), "val"))
    print(f"sdf, split bf16 x3 (mp_tf_sdf_val): {n / t / 1e6:.1f} Mpts/s, algorithmic {2 * M_IMP * n / t / 1e12:.0f} TFLOP/s, "
          f"executed {6 * M_IMP * n / t / 1e12:.0f}  ({t * 1e3:.1f} ms)")
if which in ("all", "shade_forward"):
    t = timeit(lambda: hip.shade_points(imp, ren, x, jinv, cond, mode="forward"))
    print(f"shade+color (forward-mode kernel): {n / t / 1e6:.1f} Mpts/s  ({t * 1e3:.1f} ms)")
