#!/usr/bin/env python
"""Layer-fused SDF training kernels (csrc/tfuse.hip) against the layer-wise path on one person's worth of points:
    python tools/tfuse_bench.py [points] [reps]
prints ms per forward / backward of ImplicitTrainFused and ImplicitTrainRev (split-bf16 GEMMs) and the kernels' own times."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import hip, train as T   # noqa: E402
from tests.util import seeded_networks    # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 63000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m, _ = seeded_networks(1, 0)
m = m.cuda()
net = m.foreground_implicit_network_list[0]
torch.manual_seed(0)
x = (torch.rand(P, 3, device="cuda") - 0.5) * 1.6
cond = torch.randn(69, device="cuda") * 0.1
dZ = torch.randn(P, 257, device="cuda") * 1e-3
dg = torch.randn(P, 3, device="cuda") * 1e-3
dfeat, dsdf = dZ[:, 1:].contiguous(), dZ[:, 0].contiguous()


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, cls in (("fused", T.ImplicitTrainFused), ("layer-wise", T.ImplicitTrainRev)):
    holder = {}

    def fwd():
        holder["it"] = cls(net, x, cond)

    def bwd():
        it = holder["it"]
        if isinstance(it, T.ImplicitTrainFused):
            it.backward(dfeat, dsdf, dg)
        else:
            it.backward(dZ, dg)
        it.param_grads()

    t_f = timed(fwd, reps)
    t_fb = timed(lambda: (fwd(), bwd()), reps)
    print(f"{name:10s} P={P}: forward {t_f:.3f} ms, forward+backward {t_fb:.3f} ms", flush=True)

# the two fused kernels alone
it = T.ImplicitTrainFused(net, x, cond)
fs, L, st = it.fs, hip.lib(), hip.stream()
dw8 = torch.zeros(257, device="cuda")
t1 = timed(lambda: L.mp_tf_sdf_fwd(T._p(fs.wpack), T._p(fs.bias_all), T._p(it.w8), T._p(it.arena), P, T._p(it.feat), T._p(it.sdf), st),
           reps)
it.backward(dfeat, dsdf, dg)
t2 = timed(lambda: L.mp_tf_sdf_bwd(T._p(fs.wpack), T._p(it.w8), T._p(it.arena), P, T._p(dfeat), T._p(dsdf), T._p(dw8), T._p(dw8), st),
           reps)
flop = P * 2 * 542208 * 2            # value sweep + gradient sweep, algorithmic
print(f"k_tf_sdf_fwd {t1:.3f} ms ({flop / t1 / 1e9:.1f} TFLOP/s algorithmic), k_tf_sdf_bwd {t2:.3f} ms "
      f"({flop / t2 / 1e9:.1f} TFLOP/s)")
