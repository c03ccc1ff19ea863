"""2-rank check of person-sharded TRAINING (parallel.train_person_sharded + PersonShardedGradSync) against the
single-process training step with the same random draws (run by tests/test_parallel_gpu.py through torch.distributed.run;
both ranks may share one GPU: backend gloo)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import parallel, train          # noqa: E402
from multiply_amd.config import load_config      # noqa: E402
from multiply_amd.loss import Loss                # noqa: E402
from tests.test_render_gpu import build           # noqa: E402


def step(model, gin, gt, loss_fn, draws, sharded):
    model.zero_grad(set_to_none=True)
    out = parallel.train_person_sharded(model, gin, draws=draws) if sharded else train.forward_train(model, gin, draws=draws)
    loss = loss_fn(out, gt)["loss"]
    loss.backward()
    if sharded:
        parallel.PersonShardedGradSync(model)()
    torch.cuda.synchronize()
    return out, loss.detach(), {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in model.named_parameters()}


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(rank % torch.cuda.device_count())
    ok = True
    for epoch in (301, 100):                      # 100: in / off-surface flags travel with the rows (multiply.py:311-315)
        model, oracle, inp = build(H=11, W=11)
        model.train()
        R = inp["uv"].shape[1]
        gin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
        gin.update(current_epoch=epoch, index_outside=torch.zeros(R, dtype=torch.bool), smpl_pose_last=gin["smpl_pose"] + 0.01)
        g = torch.Generator().manual_seed(5)
        gt = {"rgb": torch.rand(1, R, 3, generator=g)}
        loss_fn = Loss(load_config().loss)
        cx_all = model._setup(gin, -1, False)
        gen = torch.Generator(device="cuda").manual_seed(11)
        draws = train.make_draws(model, cx_all, gen)
        out1, loss1, g1 = step(model, gin, gt, loss_fn, draws, sharded=False)
        out2, loss2, g2 = step(model, gin, gt, loss_fn, draws, sharded=True)
        d_loss = abs(float(loss1) - float(loss2))
        print(f"[rank {rank}] epoch {epoch}: loss single {float(loss1):.6f} sharded {float(loss2):.6f}", flush=True)
        ok = ok and d_loss <= 1e-6 * max(1.0, abs(float(loss1)))
        for k in ("rgb_values", "acc_map", "acc_person_list", "grad_theta", "points"):
            d = (torch.nan_to_num(out1[k].detach()) - torch.nan_to_num(out2[k].detach())).abs().max().item()
            if not d < 1e-5:
                print(f"[rank {rank}] epoch {epoch}: output {k} differs by {d:.3e}", flush=True)
            ok = ok and d < 1e-5
        if epoch < 250:
            same = torch.equal(out1["index_off_surface"], out2["index_off_surface"]) \
                and torch.equal(out1["index_in_surface"], out2["index_in_surface"])
            if not same:
                print(f"[rank {rank}] epoch {epoch}: in / off-surface flags differ", flush=True)
            ok = ok and same
        worst, n_cmp, n_none = 0.0, 0, 0
        for name, a in g1.items():
            b = g2[name]
            person = None
            if name.startswith("foreground_"):
                person = int(name.split(".")[1])
            if person is not None and person % world != rank:
                n_none += int(b is None)
                ok = ok and b is None             # another rank's person: no gradient here
                continue
            if a is None:
                continue
            rel = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-12)
            if not rel < 2e-4:
                print(f"[rank {rank}] epoch {epoch}: gradient of {name} differs by {rel:.3e} (relative to its maximum)", flush=True)
            worst, n_cmp = max(worst, rel), n_cmp + 1
        print(f"[rank {rank}] epoch {epoch}: {n_cmp} gradients compared, worst relative difference {worst:.2e}; "
              f"{n_none} remote-person tensors without gradient; host-hull fall-backs so far {getattr(model, 'hull_host_fallbacks', 0)}", flush=True)
        ok = ok and worst < 2e-4 and n_cmp > 20 and n_none > 20
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() > 0.5 else 1)


if __name__ == "__main__":
    main()
