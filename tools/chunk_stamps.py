#!/usr/bin/env python
"""Ablation tool: per-chunk shader-clock stamps of workgroup 0 of the fused MLP kernels.
Needs a library built with the stamp instrumentation (it costs ~100 cycles per stamp, so never the product build):
    MP_EXTRA_FLAGS=-DMP_EXP_STAMP python -m multiply_amd.build --force && cp multiply_amd/libmultiply_hip.so /tmp/stamp.so
    python -m multiply_amd.build --force          # restore the product build
    MP_LIB_PATH=/tmp/stamp.so python tools/chunk_stamps.py
Prints, per kernel, the mean cycles a wave spends per weight chunk in compute / DMA wait / barrier, per wave of the
workgroup (waves 0-3 are the first wave of their SIMD, 4-7 the second)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import hip
from tests.util import seeded_networks
m, _ = seeded_networks(2, 0); m = m.cuda()
imp, ren = m.foreground_implicit_network_list[0], m.foreground_rendering_network_list[0]
n = 2_000_000
x = (torch.rand(n, 3, device="cuda") - 0.5) * 1.6
jinv = torch.eye(3, device="cuda").reshape(1, 9).repeat(n, 1).contiguous()
cond = torch.randn(69, device="cuda") * 0.1
L = hip.lib()
L.mp_debug_stamps.argtypes = [C.c_void_p]; L.mp_debug_stamps.restype = C.c_int
buf = np.zeros(8 * 128 * 4, dtype=np.uint64)
def grab(tag):
    torch.cuda.synchronize()
    assert L.mp_debug_stamps(buf.ctypes.data) == 0
    s = buf.reshape(8, 128, 4).astype(np.int64)
    n = int((s[0, :, 0] > 0).sum())
    comp, vm, bar = s[:, :n, 1] - s[:, :n, 0], s[:, :n, 2] - s[:, :n, 1], s[:, :n, 3] - s[:, :n, 2]
    print(f"{tag}: {n} chunks stamped; mean cycles per chunk: compute {comp.mean():.0f}, dma wait {vm.mean():.0f}, "
          f"barrier {bar.mean():.0f}")
    print("   compute per wave:", np.round(comp.mean(1)).astype(int).tolist())
    print("   barrier per wave:", np.round(bar.mean(1)).astype(int).tolist())
for _ in range(2):
    hip.shade_points(imp, ren, x, jinv, cond)      # fwdsave, grad, colour: the last kernel's stamps survive
grab("color")
pki = hip.packed(imp, "full", 2); pki.refresh(cond)
sdf = torch.empty(n, device="cuda"); nrm = torch.empty(n, 3, device="cuda")
feat = torch.empty((n + 255) // 256 * 4 * 8 * 4 * 1024, dtype=torch.uint8, device="cuda")
hip.shade_rev_launch(pki, hip.grad_net(imp), x, jinv, None, None, n, sdf, nrm, feat)
grab("grad")
hip.implicit_sdf(imp, x, cond)
grab("sdf")
