"""Normal error of the reverse-mode shading pair for the library loaded through MP_LIB_PATH (side libraries built with
-DMP_EXP_SIGBITS=4 | 6 emulate a 4- / 6-bit stored-sigmoid code inside the byte pipeline; -DMP_EXP_SIG16 stores halves):
    MP_LIB_PATH=multiply_amd/ab_libs/libmultiply_hip_sig4.so python tools/sig_bits.py [n_points]
20 000 random points in the body's box against the fp32 oracle under torch autograd (the set-up of
tests/test_mlp_gpu.py::test_shade_points); asserted bound there: max 1.5e-2 (tests/tolerances.py MLP['shade_normal'])."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import hip                     # noqa: E402
from oracle import multiply_oracle as O          # noqa: E402
from tests.util import seeded_networks           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m, _ = seeded_networks(2, 0)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
m = m.cuda()
rng = np.random.RandomState(4)
x = torch.tensor(rng.uniform(-0.9, 0.9, (n, 3)), dtype=torch.float32)
cond = torch.tensor(rng.normal(0, 0.1, 69), dtype=torch.float32)
A = torch.tensor(rng.normal(0, 0.3, (n, 3, 3)), dtype=torch.float32) + torch.eye(3)
jinv = torch.linalg.inv(A)
xg = x.clone().requires_grad_(True)
out = O.implicit_forward(sd, "foreground_implicit_network_list.0.", xg, cond, multires=6)
g = torch.autograd.grad(out[:, :1], xg, torch.ones(n, 1))[0]
nrm = torch.nn.functional.normalize(torch.einsum("bi,bij->bj", g, jinv), dim=1)
sdf_g, nrm_g, rgb_g = hip.shade_points(m.foreground_implicit_network_list[0], m.foreground_rendering_network_list[0], x.cuda(),
                                       jinv.cuda(), cond.cuda(), mode="reverse")
torch.cuda.synchronize()
e = (nrm_g.cpu().double() - nrm.double()).abs()
ray = e.max(1).values
print(f"lib {os.path.basename(os.environ.get('MP_LIB_PATH', 'libmultiply_hip.so'))}: sig bytes/point {hip.lib().mp_sig_bytes_per_point()}; "
      f"normals vs fp32 oracle on {n} random points: max {float(e.max()):.3e} mean {float(e.mean()):.3e} p99 {float(torch.quantile(ray, 0.99)):.2e} "
      f"p99.9 {float(torch.quantile(ray, 0.999)):.2e}  points > 1.5e-2: {int((ray > 1.5e-2).sum())}, > 5e-3: {int((ray > 5e-3).sum())}")
