"""The oracle's two end-to-end entry points run on the CPU on a handful of rays: every GPU parity test, bench.py's cpu_baseline leg
and smoke() call them on the GPU box, where a broken oracle would only show at round end (round 5: an edit of forward_train had
also landed in forward_eval and no CPU test noticed)."""
import numpy as np
import torch

from oracle import multiply_oracle as O
from tests.util import seeded_networks, t32


def _scene(P=2, H=3, W=3):
    from multiply_amd.synthetic import make_scene, make_smpl_tables
    tables = make_smpl_tables(0)
    sc = make_scene(P, seed=0, H=H, W=W)
    m, _ = seeded_networks(P, 0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sp = t32(sc["smpl_params"])
    inp = dict(uv=t32(sc["uv"]), intrinsics=t32(sc["intrinsics"]), pose=t32(sc["pose"]), smpl_params=sp, smpl_pose=sp[:, :, 4:76],
               smpl_shape=sp[:, :, 76:], smpl_trans=sp[:, :, 1:4], idx=torch.tensor([3]))
    return O.MultiplyOracle(sd, tables, sc["smpl_params"][0, :, 76:]), inp


def test_oracle_eval_and_training_forward_run():
    oracle, inp = _scene()
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(0, R, 2)]
    out = oracle.forward_eval(inp, hit)
    assert out["rgb_values"].shape == (R, 3) and out["acc_person_list"].shape == (R, 2)
    assert bool(torch.isfinite(out["acc_map"]).all()) and float(out["acc_map"].max()) <= 1.0 + 1e-5
    z = [torch.cat([zv, zm[:, None]], 1) for zv, zm in zip(out["z_vals"], out["z_max"])]
    g = torch.Generator().manual_seed(0)
    nv = oracle.persons[0].server.verts_c.shape[0]
    draws = {"person": {p: dict(eik_idx=torch.randint(0, nv, (16,), generator=g), eik_noise=torch.randn(16, 3, generator=g),
                                surf_idx=torch.randint(0, nv, (R,), generator=g)) for p in range(2)},
             "bg_rand": torch.rand(R, 32, generator=g),
             "zp_idx": {(q, p): torch.randint(0, nv, (8,), generator=g) for q in range(2) for p in range(2)}}
    for v in oracle.sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    tr = oracle.forward_train(inp, hit, z, draws)
    assert tr["rgb_values"].shape == (R, 3) and tr["grad_theta"].shape == (1, 32, 3)
    assert float(tr["smpl_surface_loss"]) >= 0.0 and tr["zero_pose_loss"].shape == (1,)
    loss = tr["rgb_values"].nan_to_num().sum() + tr["smpl_surface_loss"].sum() + tr["zero_pose_loss"].sum()
    gw = torch.autograd.grad(loss, [v for v in oracle.sd.values() if v.requires_grad], allow_unused=True)
    assert sum(g is not None and bool(torch.isfinite(g).all()) for g in gw) > 50
