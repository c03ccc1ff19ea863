"""diagnostic: one training iteration with P persons, device vs oracle: per-person grad_theta, loss terms, worst gradient tensors"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_render_gpu import build
from tests.test_train_step_gpu import _cpu
from multiply_amd.loss import Loss
from multiply_amd.config import load_config
import multiply_amd.loss as LM

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model, oracle, inp = build(P=P, H=11, W=11, seed=1)
model.train()
R = inp["uv"].shape[1]
gin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
gin.update(current_epoch=301, index_outside=torch.zeros(R, dtype=torch.bool), smpl_pose_last=gin["smpl_pose"] + 0.01)
g = torch.Generator().manual_seed(5)
gt = {"rgb": torch.rand(1, R, 3, generator=g)}
loss_fn = Loss(load_config().loss)
hit = [torch.arange(R) for _ in range(P)]
for fused in (True, False):
    LM.FUSED = fused
    torch.manual_seed(3)
    out = model({**gin, "hit_index": hit})
    lo = loss_fn(out, {"rgb": gt["rgb"].cuda()})
    model.zero_grad()
    lo["loss"].backward()
    torch.cuda.synchronize()
    graph = model._last_train
    print("fused loss" if fused else "torch loss", {k: round(float(v), 6) for k, v in lo.items() if torch.is_tensor(v)})
    if fused:
        gg = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
        keep = (out, graph)
out, graph = keep
for v in oracle.sd.values():
    if v.is_floating_point():
        v.requires_grad_(True)
z_given = [graph.fg[p]["zfinal"].cpu() for p in range(P)]
want = oracle.forward_train(inp, hit, z_given, _cpu(graph.draws))
tl = torch.mean(torch.square((inp["smpl_pose"] + 0.01) - inp["smpl_pose"]))
want.update(fg_rgb_values_each_person_list=[], index_in_surface=None, epoch=301, temporal_loss=tl.reshape(()))
lw = loss_fn(want, gt)
print("oracle    ", {k: round(float(v), 6) for k, v in lw.items() if torch.is_tensor(v)})
gth_g, gth_o = out["grad_theta"].detach().cpu().reshape(P, -1, 3), want["grad_theta"].detach().reshape(P, -1, 3)
for p in range(P):
    print(f"person {p}: grad_theta max err {float((gth_g[p] - gth_o[p]).abs().max()):.3e}  |g| mean gpu {float(gth_g[p].norm(dim=-1).mean()):.4f} oracle {float(gth_o[p].norm(dim=-1).mean()):.4f}")
for k in ("rgb_values", "acc_map", "acc_person_list"):
    print(k, float((out[k].detach().cpu() - want[k].detach()).abs().nan_to_num().max()))
names = [k for k, v in oracle.sd.items() if v.requires_grad]
gw = torch.autograd.grad(lw["loss"], [oracle.sd[k] for k in names], allow_unused=True)
rows = []
for k, w in zip(names, gw):
    if w is None or k not in gg:
        continue
    a, b = gg[k].double().reshape(-1), w.double().reshape(-1)
    rows.append((float((a - b).norm() / (b.norm() + 1e-12)), k, float(b.norm())))
for r in sorted(rows, reverse=True)[:12]:
    print("%.3e  %-60s |g| %.3e" % r)
