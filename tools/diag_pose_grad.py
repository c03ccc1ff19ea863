#!/usr/bin/env python
"""Diagnostic (round 6): d loss / d smpl_pose of a training step, device vs oracle, with and without non-zero pose-conditioning
columns in layer 0 of the SDF nets (the geometric initialisation zeroes them, so tests on fresh weights never exercise d cond)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_train_step_gpu import _train_setup, _cpu

def run(perturb_imp, perturb_ren, epoch=301):
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    torch.manual_seed(21)
    with torch.no_grad():
        if perturb_imp:
            for net in model.foreground_implicit_network_list:
                net.lin0.weight_v[:, 39:] += 0.05 * torch.randn_like(net.lin0.weight_v[:, 39:])
        if perturb_ren:
            for net in model.foreground_rendering_network_list:
                net.lin_pose.weight += 0.05 * torch.randn_like(net.lin_pose.weight)
    oracle.sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for pp in oracle.persons:
        pp.sd = oracle.sd
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(R)]
    gin["smpl_pose"] = gin["smpl_pose"].clone().requires_grad_(True)
    gin["smpl_pose_last"] = gin["smpl_pose"].detach() + 0.01
    out = model({**gin, "hit_index": hit, "current_epoch": epoch})
    lo = loss_fn(out, gt)
    lo["loss"].backward()
    torch.cuda.synchronize()
    graph = model._last_train
    oin = dict(inp)
    oin["smpl_pose"] = inp["smpl_pose"].clone().requires_grad_(True)
    z_given = [graph.fg[p]["zfinal"].cpu() for p in range(2)]
    want = oracle.forward_train(oin, hit, z_given, _cpu(graph.draws))
    tl = torch.mean(torch.square(inp["smpl_pose"] + 0.01 - oin["smpl_pose"]))
    v = out["index_in_surface"]
    want.update(fg_rgb_values_each_person_list=[], index_in_surface=None if v is None else v.cpu(), epoch=epoch, temporal_loss=tl,
                smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1), sam_mask=gin["sam_mask"].squeeze().cpu())
    lw = loss_fn(want, gt)
    (g,) = torch.autograd.grad(lw["loss"], [oin["smpl_pose"]])
    a = gin["smpl_pose"].grad.cpu()
    print(f"perturb imp={perturb_imp} ren={perturb_ren} epoch={epoch}: loss gpu {float(lo['loss']):.6f} oracle {float(lw['loss']):.6f}")
    for p in range(2):
        for name, sl in (("global orient [0:3]", slice(0, 3)), ("body pose [3:72]", slice(3, 72))):
            d = (a[0, p, sl] - g[0, p, sl]).abs().max().item()
            print(f"   person {p} {name:20s} |want|max {g[0, p, sl].abs().max().item():.3e}  max err {d:.3e}  rel {d / (g[0, p, sl].abs().max().item() + 1e-12):.2e}")

run(False, False)
run(True, False)
run(False, True)
run(True, False, epoch=30)
