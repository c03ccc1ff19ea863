"""2-rank check of frame-sharded data-parallel training and of the epoch stages under it (run by
tests/test_parallel_gpu.py through torch.distributed.run, the ranks may share one GPU: backend gloo).

Step: rank r trains on frame r -- its rows of the BodyModelParams tables feed the model's smpl_pose / smpl_trans / smpl_shape --
and ONE flat all-reduce averages the gradients of the networks and of the tables.  Checked against the same process computing
both frames one after the other: the all-reduced gradient of EVERY tensor equals the mean of the two single-frame gradients, each
table's gradient is non-zero exactly in the two frames' rows, and after an Adam step the replicas are bit-identical.
Stages: rank 0 extracts the canonical meshes and rasterises the instance masks on the device; rank 1 receives them bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import parallel                         # noqa: E402
from multiply_amd.body_model_params import BodyModelParams  # noqa: E402
from tests.test_render_gpu import build                   # noqa: E402


def main():
    backend = os.environ.get("MP_DIST_BACKEND", "gloo")     # "nccl" (= RCCL): one GPU per rank, real device collectives
    dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"              # where the small control tensors of the collectives live
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(rank % torch.cuda.device_count())
    from multiply_amd.config import load_config
    from multiply_amd.loss import Loss
    model, oracle, inp = build(H=11, W=11)
    model.train()
    R = inp["uv"].shape[1]
    F, P = 4, 2
    dev = lambda d: {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
    gin = dev(inp)
    gin.update(current_epoch=301, index_outside=torch.zeros(R, dtype=torch.bool, device="cuda"))
    ok = True

    def check(cond, what):
        nonlocal ok
        if not cond:
            print(f"[rank {rank}] FAILED: {what}", flush=True)
        ok = ok and bool(cond)

    # per-frame body parameters: the scene's pose, nudged per frame (identical on every rank)
    g = torch.Generator().manual_seed(7)
    body = []
    for p in range(P):
        bm = BodyModelParams(F).cuda()
        sp = inp["smpl_params"][0, p]
        bm.init_parameters("betas", sp[76:86][None].cuda(), requires_grad=True)
        bm.init_parameters("global_orient", (sp[4:7][None] + 0.02 * torch.randn(F, 3, generator=g)).cuda(), requires_grad=True)
        bm.init_parameters("body_pose", (sp[7:76][None] + 0.02 * torch.randn(F, 69, generator=g)).cuda(), requires_grad=True)
        bm.init_parameters("transl", (sp[1:4][None] + 0.01 * torch.randn(F, 3, generator=g)).cuda(), requires_grad=True)
        body.append(bm)
    loss_fn = Loss(load_config().loss)
    targets = [{"rgb": torch.rand(1, R, 3, generator=torch.Generator().manual_seed(50 + f)).cuda()} for f in range(F)]
    hit = [torch.arange(R), torch.arange(R)]
    tin = {**gin, "hit_index": hit}

    trainer = parallel.FrameShardedTrainer(model, body, loss_fn, optimizers=[])
    names = [n for n, p_ in model.named_parameters() if p_.requires_grad] + \
            [f"body{p}.{n}" for p in range(P) for n, p_ in body[p].named_parameters() if p_.requires_grad]
    check(len(names) == len(trainer.sync.params), "parameter list")

    # ---- reference: both frames in this process, one after the other (same draws: the generator is seeded per frame)
    single = []
    for f in range(world):
        torch.manual_seed(100 + f)
        trainer.step(tin, targets[f], f, sync=False)     # no collective: this is the single-process gradient of frame f
        single.append([p_.grad.detach().clone() if p_.grad is not None else torch.zeros_like(p_) for p_ in trainer.sync.params])
    want = [sum(gs) / world for gs in zip(*single)]

    # ---- the sharded step: my frame only, one flat all-reduce
    torch.manual_seed(100 + rank)
    lo = trainer.step(tin, targets[rank], rank)
    torch.cuda.synchronize()
    worst = 0.0
    for n, p_, w in zip(names, trainer.sync.params, want):
        a, b = p_.grad.double().reshape(-1), w.double().reshape(-1)
        rel = float((a - b).norm() / (b.norm() + 1e-12))
        if float((a - b).abs().max()) > 1e-7:
            # two runs of the same frame are not bitwise equal (fp32 atomics of the split-K weight-gradient GEMMs; a ReLU mask
            # of a colour net can flip for a sample whose pre-activation is within round-off of 0): same bounds as vs the oracle
            worst = max(worst, rel)
            check(rel < (2e-2 if "rendering" in n else 5e-3), f"{n}: all-reduced gradient vs mean of the single-frame gradients, rel {rel:.2e}")
    for p in range(P):
        for n in ("global_orient", "body_pose", "transl"):
            gr = getattr(body[p], n).weight.grad
            rows = torch.nonzero(gr.abs().sum(1) > 0).flatten().tolist()
            check(rows == list(range(world)), f"body{p}.{n}: gradient rows {rows}, expected the {world} frames of this step")
    print(f"[rank {rank}] frame {rank}: loss {float(lo['loss']):.6f}; {len(names)} tensors (networks + body-model tables), worst "
          f"relative difference to the mean of the single-frame gradients {worst:.2e}", flush=True)

    # ---- a real optimiser step on both ranks: the replicas stay identical
    opt = torch.optim.Adam([{"params": [p_ for p_ in model.parameters() if p_.requires_grad]},
                            {"params": [p_ for bm in body for p_ in bm.parameters() if p_.requires_grad], "lr": 1e-4}], lr=5e-4)
    trainer.opts = [opt]
    torch.manual_seed(200 + rank)
    trainer.step(tin, targets[rank], rank)
    torch.cuda.synchronize()
    flat = torch.cat([p_.detach().reshape(-1).double().cpu() for p_ in trainer.sync.params]).to(cdev)
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    check(all(torch.equal(both[0], b) for b in both[1:]), "replicas differ after the optimiser step")

    # ---- epoch stages: produced on rank 0 by the device extractors, received bit for bit
    model.eval()
    vs, fs = parallel.refresh_canonical_meshes_broadcast(model, res_up=1)
    sig = torch.tensor([float(sum(v.double().sum() for v in vs)), float(sum(f.double().sum() for f in fs)),
                        float(sum(v.numel() for v in vs)), float(sum(f.numel() for f in fs))], dtype=torch.float64, device=cdev)
    sigs = [torch.zeros_like(sig) for _ in range(world)]
    dist.all_gather(sigs, sig)
    check(all(torch.equal(sigs[0], s_) for s_ in sigs[1:]), "canonical meshes differ between the ranks")
    check(all(v.shape[1] > 100 for v in vs) and model.mesh_face_vertices_list[0].shape[2:] == (3, 3), "canonical meshes look empty")
    print(f"[rank {rank}] canonical meshes from rank 0: {[int(v.shape[1]) for v in vs]} vertices, {[int(f.shape[0]) for f in fs]} faces",
          flush=True)
    flag = torch.tensor([0 if ok else 1], device=cdev)
    dist.all_reduce(flag)
    dist.destroy_process_group()
    sys.exit(1 if int(flag) else 0)


if __name__ == "__main__":
    main()
