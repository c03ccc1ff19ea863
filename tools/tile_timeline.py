#!/usr/bin/env python
"""Ablation tool: shader-clock timeline of the LAST tile of workgroup 0 of k_mlp_sdf (library built with -DMP_EXP_STAMP):
tile-level phases (input staging, prologue, network, output) and the start time of every weight chunk per wave.
    MP_LIB_PATH=.../libmultiply_hip_stamp.so python tools/tile_timeline.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import hip
from tests.util import seeded_networks
m, _ = seeded_networks(2, 0); m = m.cuda()
imp = m.foreground_implicit_network_list[0]
n = 256 * 256 * 30
x = (torch.rand(n, 3, device="cuda") - 0.5) * 1.6
cond = torch.randn(69, device="cuda") * 0.1
L = hip.lib()
L.mp_debug_stamps.argtypes = [C.c_void_p]; L.mp_debug_stamps.restype = C.c_int
buf = np.zeros(8 * 128 * 4, dtype=np.uint64)
which = sys.argv[1] if len(sys.argv) > 1 else "sdf"       # sdf | fwdsave | grad | color (the last kernel launched is stamped)
ren = m.foreground_rendering_network_list[0]
jinv = torch.eye(3, device="cuda").reshape(1, 9).repeat(n, 1).contiguous()
for _ in range(2):
    if which == "sdf":
        hip.implicit_sdf(imp, x, cond)
    elif which == "color":
        hip.shade_points(imp, ren, x, jinv, cond)
    else:
        pki = hip.packed(imp, "full", 2); pki.refresh(cond)
        sdf = torch.empty(n, device="cuda"); nrm = torch.empty(n, 3, device="cuda")
        feat = torch.empty((n + 255) // 256 * 4 * 8 * 4 * 1024, dtype=torch.uint8, device="cuda")
        if which == "grad":
            hip.shade_rev_launch(pki, hip.grad_net(imp), x, jinv, None, None, n, sdf, nrm, feat)
        else:   # fwdsave: launch the pair, then the forward sweep alone is not separable -> stamp build marks only fwdsave
            os.environ["MP_TIMELINE_FWD_ONLY"] = "1"
            hip.shade_rev_launch(pki, hip.grad_net(imp), x, jinv, None, None, n, sdf, nrm, feat)
torch.cuda.synchronize()
assert L.mp_debug_stamps(buf.ctypes.data) == 0
s = buf.reshape(8, 128, 4).astype(np.int64)
t0 = s[:, 120, 0].min() if s[:, 120, 0].min() > 0 else s[:, 120, 2].min()   # kernels without the staging stamps
print("tile phases (cycles from the earliest wave's tile start), per wave 0..7")
for name, (slot, ev) in {"tile start": (120, 0), "inputs staged": (120, 1), "prologue done": (120, 2), "network done": (120, 3),
                         "outputs written": (121, 0)}.items():
    print(f"  {name:16s}", (s[:, slot, ev] - t0).tolist())
nch = int((s[0, :120, 0] > 0).sum())
print(f"{nch} chunks; per chunk and wave (0 = first wave of SIMD 0, 4 = second): start, then the spans between the stamps")
print("  phase-separated layers: waves 0..3: M | V | barrier wait;  waves 4..7: M | barrier wait + DMA issue | V;   old stream: compute | dma wait | barrier")
for c in range(nch):
    r = []
    for w in (0, 4):
        r.append((int(s[w, c, 0] - t0), int(s[w, c, 1] - s[w, c, 0]), int(s[w, c, 2] - s[w, c, 1]), int(s[w, c, 3] - s[w, c, 2])))
    print(f"  chunk {c:2d}: wave0 @{r[0][0]:7d} {r[0][1:]}   wave4 @{r[1][0]:7d} {r[1][1:]}")
print("  next tile start ", "(stamps of the LAST tile: its start relative to the previous tile is not recorded)")
tot = max(s[:, 121, 0].max(), s[:, 120, 3].max()) - t0
print("tile total", tot, "cycles")
