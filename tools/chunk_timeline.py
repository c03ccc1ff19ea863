#!/usr/bin/env python
"""Ablation tool (round 6): absolute shader-clock timeline of consecutive weight chunks of workgroup 0, per wave -- when each wave
starts M(c), ends M(c), passes / reaches the chunk barrier and ends V(c) (mlp_core.hpp MP_STAMP events 0..3).  Needs a stamp build
restricted to one kernel kind:   tools/ab_build.sh stamp1:"-DMP_EXP_STAMP -DMP_STAMP_HID=1"   (1 = ReLU: the colour kernel)
    MP_LIB_PATH=multiply_amd/ab_libs/libmultiply_hip_stamp1.so python tools/chunk_timeline.py [first_chunk] [n_chunks]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import hip
from tests.util import seeded_networks
c0 = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m, _ = seeded_networks(2, 0); m = m.cuda()
imp, ren = m.foreground_implicit_network_list[0], m.foreground_rendering_network_list[0]
n = 2_000_000
x = (torch.rand(n, 3, device="cuda") - 0.5) * 1.6
jinv = torch.eye(3, device="cuda").reshape(1, 9).repeat(n, 1).contiguous()
cond = torch.randn(69, device="cuda") * 0.1
L = hip.lib()
L.mp_debug_stamps.argtypes = [C.c_void_p]; L.mp_debug_stamps.restype = C.c_int
buf = np.zeros(8 * 128 * 4, dtype=np.uint64)
for _ in range(2):
    hip.shade_points(imp, ren, x, jinv, cond)
    hip.implicit_sdf(imp, x, cond)
torch.cuda.synchronize()
assert L.mp_debug_stamps(buf.ctypes.data) == 0
s = buf.reshape(8, 128, 4).astype(np.int64)
t0 = s[:, c0, 0].min()
print("cycles relative to the earliest wave's start of chunk", c0, "; waves 0-3 early, 4-7 late; SIMD pairs (w, w+4)")
print("chunk wave |  M start   M end   barrier   V end | M len  V len(+barrier wait)")
for c in range(c0, c0 + nc):
    for w in (0, 4, 1, 5, 2, 6, 3, 7):
        e = s[w, c] - t0
        print(f"{c:5d} {w:4d} | {e[0]:8d} {e[1]:7d} {e[2]:9d} {e[3]:7d} | {e[1]-e[0]:5d} {e[3]-e[1]:6d}")
per = (s[:, c0 + nc, 0] - s[:, c0, 0]) / nc
print("cycles per chunk per wave:", np.round(per).astype(int).tolist())
