"""diagnostic: bench.py's training-parity leg (512 random rays of the frame, own cull) for any persons / samples, with the loss
terms, per-person eikonal gradients and the worst gradient tensors printed:  python tools/diag_bench_train.py [persons] [samples]"""
import contextlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import numpy as np, torch
import bench
from oracle import multiply_oracle as O
from multiply_amd.config import load_config
from multiply_amd.loss import Loss

P = int(args[0]) if args else 4
NS = int(args[1]) if len(args) > 1 else 256
rays = 512
model, inp, tables, sc = bench.build_model(NS, seed=0, P=P)
model.convergence_group = 512
gin = bench.to_dev(inp)
dev = gin["uv"].device
g = torch.Generator().manual_seed(0)
R = gin["uv"].shape[1]
sel = torch.randperm(R, generator=g)[:rays]
tin = dict(gin)
tin["uv"] = gin["uv"][:, sel.to(dev)].contiguous()
tin.update(current_epoch=301, index_outside=torch.zeros(rays, dtype=torch.bool, device=dev), smpl_pose_last=gin["smpl_pose"] + 0.01)
loss_fn = Loss(load_config().loss)
gt = {"rgb": torch.rand(1, rays, 3, generator=g)}
model.train()
model.zero_grad(set_to_none=True)
out = model(tin)
lo_gpu = loss_fn(out, {"rgb": gt["rgb"].to(dev)})
lo_gpu["loss"].backward()
torch.cuda.synchronize()
graph = model._last_train
model.eval()
print("gpu   ", {k: round(float(v), 6) for k, v in lo_gpu.items() if torch.is_tensor(v)})
cx = graph.cx
print("hit rays", cx["n_hit"], "Pt", [graph.fg[p]["Pt"] for p in cx["persons"]])
hit = [cx["per"][p]["hit_index"][:graph.fg[p]["Rp"]].long().cpu() for p in cx["persons"]]
z_given = [graph.fg[p]["zfinal"].cpu() for p in cx["persons"]]
draws = {"person": {p: {k: v.cpu() for k, v in d.items()} for p, d in graph.draws["person"].items()}, "bg_rand": graph.draws["bg_rand"].cpu()}
sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
oracle = O.MultiplyOracle(sd, tables, sc["smpl_params"][0, :, 76:], O.SamplerCfg(N_samples=NS, N_samples_eval=max(128, NS)))
oracle.sd = sd
for pp in oracle.persons:
    pp.sd = sd
oin = dict(inp)
oin["uv"] = inp["uv"][:, sel]
tl = torch.mean(torch.square((inp["smpl_pose"] + 0.01) - inp["smpl_pose"])).reshape(())
want = oracle.forward_train(oin, hit, z_given, draws)
want.update(fg_rgb_values_each_person_list=[], index_in_surface=None, epoch=301, temporal_loss=tl)
lo = loss_fn(want, gt)
print("oracle", {k: round(float(v), 6) for k, v in lo.items() if torch.is_tensor(v)})
gg, go = out["grad_theta"].detach().cpu().reshape(P, -1, 3), want["grad_theta"].detach().reshape(P, -1, 3)
for p in range(P):
    e = (gg[p] - go[p]).abs().max(1).values
    print(f"person {p}: grad_theta max err {float(e.max()):.3e}, points off by > 1e-3: {int((e > 1e-3).sum())} of {len(e)}; first bad rows {torch.nonzero(e > 1e-3).flatten()[:8].tolist()}")
for k in ("rgb_values", "acc_map", "acc_person_list", "normal_values"):
    print(k, float((out[k].detach().cpu() - want[k].detach()).abs().nan_to_num().max()))
