"""2-rank check of the ray-sharded data-parallel path (BASELINE.json configs[2]) against the CPU ORACLE
(run by tests/test_parallel_gpu.py through torch.distributed.run; the ranks may share one GPU: backend gloo).

Render: the frame's convergence groups are dealt on a diagonal lattice (parallel.shard_input_interleaved); every rank renders its share
through the HIP path, compares it with the oracle's render of the SAME rays, and the all_gather'ed image equals the
single-process render bit for bit.  Training: every rank runs forward + loss + backward on its own pixels, ONE flat gradient
all-reduce (parallel.GradientAllReduce); the averaged gradient of every parameter is compared with the average of the
oracle's torch-autograd gradients of the two shards."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import parallel                 # noqa: E402
from tests.test_render_gpu import build, report   # noqa: E402
from tests import tolerances as TOL               # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(rank % torch.cuda.device_count())
    GROUP = 64
    model, oracle, inp = build(H=16, W=32)                     # 512 rays = 8 convergence groups, 4 per rank
    R = inp["uv"].shape[1]
    dev = lambda d: {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
    model.convergence_group = GROUP
    share, ids = parallel.shard_input_interleaved(inp, rank, world, GROUP)
    mine = [g for g in range(R // GROUP) if (g % world + g // world) % world == rank]        # diagonal-lattice deal of the groups
    assert len(ids) == R // world and [int(ids[i * GROUP]) for i in range(len(mine))] == [g * GROUP for g in mine]
    ok = True

    def check(cond, what):
        nonlocal ok
        if not cond:
            print(f"[rank {rank}] FAILED: {what}", flush=True)
        ok = ok and bool(cond)
    # ---- render: my share vs the oracle on the same rays (the oracle's convergence vote is per call = per group here)
    got = model(dev(share))
    torch.cuda.synchronize()
    n_hit = model.last_stats["n_hit"]
    hit = [model._last["per"][p]["hit_index"][:n].long().cpu() for p, n in zip(model._last["persons"], n_hit)]
    keys = ("rgb_values", "acc_map", "normal_values", "acc_person_list")
    parts = {k: [] for k in keys}
    for g0 in range(0, len(ids), GROUP):                       # one oracle call per convergence group
        sub = dict(share)
        sub["uv"] = share["uv"][:, g0:g0 + GROUP]
        hg = [h[(h >= g0) & (h < g0 + GROUP)] - g0 for h in hit]
        hg = [h if len(h) else torch.zeros(1, dtype=torch.long) for h in hg]           # multiply.py:262-263 per chunk
        want = oracle.forward_eval(sub, hg)
        for k in keys:
            parts[k].append(want[k])
    for k in keys:
        st = report(f"[rank {rank}] my {len(ids)} rays, {k}", got[k], torch.cat(parts[k], 0))
        check(TOL.within(st, TOL.EVAL[k]), f"{k} max {st[0]:.3e} mean {st[1]:.3e} vs {TOL.EVAL[k]}")
    # ---- the gathered image equals the single-process render with the same convergence groups
    whole = model(dev(inp))
    full = parallel.gather_rays_interleaved(got["rgb_values"], R, world, GROUP)
    same = torch.equal(torch.nan_to_num(full), torch.nan_to_num(whole["rgb_values"]))
    print(f"[rank {rank}] gathered image identical to the single-process render: {same}", flush=True)
    check(same, "gathered image differs from the single-process render")

    # ---- data-parallel training step: my pixels, flat all-reduce, vs the oracle's averaged autograd gradients
    from multiply_amd.config import load_config
    from multiply_amd.loss import Loss
    loss_fn = Loss(load_config().loss)
    g = torch.Generator().manual_seed(11 + rank)
    rays = 48
    sel = torch.randperm(R, generator=g)[:rays]
    tin = dict(inp)
    tin["uv"] = inp["uv"][:, sel]
    gt = {"rgb": torch.rand(1, rays, 3, generator=g)}
    tg = dev(tin)
    tg.update(current_epoch=301, index_outside=torch.zeros(rays, dtype=torch.bool), smpl_pose_last=tg["smpl_pose"] + 0.01)
    model.convergence_group = None
    model.train()
    out = model(tg)
    lo = loss_fn(out, {"rgb": gt["rgb"].cuda()})
    model.zero_grad()
    lo["loss"].backward()
    ar = parallel.GradientAllReduce(model.parameters())
    ar()
    torch.cuda.synchronize()
    graph = model._last_train
    cx = graph.cx
    hit_t = [cx["per"][p]["hit_index"][:graph.fg[p]["Rp"]].long().cpu() for p in cx["persons"]]
    z_given = [graph.fg[p]["zfinal"].cpu() for p in cx["persons"]]
    cpu = lambda d: {k: (cpu(v) if isinstance(v, dict) else v.detach().cpu()) for k, v in d.items()}
    for v in oracle.sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    want = oracle.forward_train(tin, hit_t, z_given, cpu(graph.draws))
    tl = torch.mean(torch.square((tin["smpl_pose"] + 0.01) - tin["smpl_pose"]))      # multiply.py:242-243, from the inputs
    want.update(fg_rgb_values_each_person_list=[], index_in_surface=None, epoch=301,
                temporal_loss=tl.reshape(()), smpl_surface_loss=torch.zeros(1),
                zero_pose_loss=torch.zeros(1))
    lw = loss_fn(want, gt)
    names = [k for k, v in oracle.sd.items() if v.requires_grad]
    gw = torch.autograd.grad(lw["loss"], [oracle.sd[k] for k in names], allow_unused=True)
    got_p = dict(model.named_parameters())
    worst_rel = 0.0
    for k, gg in zip(names, gw):
        if k not in got_p:
            continue
        ref = torch.zeros_like(oracle.sd[k]) if gg is None else gg.detach().clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)             # the oracle's gradients averaged over the two shards
        ref /= world
        a = got_p[k].grad
        if float(ref.abs().max()) == 0.0:
            check(a is None or float(a.abs().max()) == 0.0, f"{k}: gradient where the oracle has none")
            continue
        a = a.detach().cpu().double().reshape(-1)
        b = ref.double().reshape(-1)
        rel = float((a - b).norm() / (b.norm() + 1e-12))
        tol = TOL.TRAIN_GRAD_REL_RENDERING if "rendering" in k else TOL.TRAIN_GRAD_REL
        check(rel < tol or float((a - b).abs().max()) < 1e-7, f"gradient of {k}: rel {rel:.3e}")
        worst_rel = max(worst_rel, rel)
    print(f"[rank {rank}] loss gpu {float(lo['loss']):.6f} oracle {float(lw['loss']):.6f}; worst relative error of the "
          f"all-reduced gradients vs the averaged oracle gradients {worst_rel:.3e}", flush=True)
    check(abs(float(lo["loss"]) - float(lw["loss"])) < 1e-4 * max(1.0, abs(float(lw["loss"]))), "loss")
    # ---- the bucketed, overlapped variant (buckets all-reduced while the backward sweep goes on): the same averaged gradients
    flat_ref = {k: p_.grad.detach().clone() for k, p_ in model.named_parameters() if p_.grad is not None}
    model.grad_bucket_sync = parallel.BucketedGradientSync()
    from multiply_amd import train as T
    out2 = T.forward_train(model, tg, draws=graph.draws)            # the same draws: the same forward
    lo2 = loss_fn(out2, {"rgb": gt["rgb"].cuda()})
    model.zero_grad()
    lo2["loss"].backward()                                          # p.grad is already the average when this returns
    torch.cuda.synchronize()
    model.grad_bucket_sync = None
    worst_b = 0.0
    for k, p_ in model.named_parameters():
        if k in flat_ref:
            g2 = p_.grad if p_.grad is not None else torch.zeros_like(flat_ref[k])      # parameters no loss term reaches
            d = float((g2 - flat_ref[k]).abs().max()) / (float(flat_ref[k].abs().max()) + 1e-30)
            worst_b = max(worst_b, d)
    print(f"[rank {rank}] bucketed overlapped all-reduce vs the flat one: worst relative difference {worst_b:.2e}", flush=True)
    check(worst_b < 2e-2, "bucketed gradient sync differs from the flat all-reduce")     # two runs: fp32 atomics, ReLU flips
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() > 0.5 else 1)


if __name__ == "__main__":
    main()
