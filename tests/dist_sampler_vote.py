"""2-rank check of the data-parallel sampler vote (run by tests/test_parallel_gpu.py through torch.distributed.run,
the ranks may share one GPU: backend gloo).

The reference's convergence vote `not_converge = beta.max() > beta0` (ray_sampler.py:137) spans every ray of the call.  With the
call's rays split over the ranks and `model.sampler_vote_group` set, ONE MAX all-reduce of the per-iteration flag makes every
rank's sampler take the single-process decisions: the depths of rank r's rays equal, BIT FOR BIT, the rows of those rays in the
single-process sampler run on all rays with the same per-ray draws -- and differ without the vote whenever a rank's own rays
converge earlier than the rest (checked: the scene is chosen so that they do)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import train                      # noqa: E402
from tests.test_render_gpu import build             # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(rank % torch.cuda.device_count())
    model, oracle, inp = build(H=16, W=32)           # 512 rays
    model.train()
    R = inp["uv"].shape[1]
    dev = lambda d: {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
    gin = dev(inp)
    gin.update(current_epoch=301)
    ok = True

    def check(cond, what):
        nonlocal ok
        if not cond:
            print(f"[rank {rank}] FAILED: {what}", flush=True)
        ok = ok and bool(cond)

    # ---- single process: all R rays, every ray of every person sampled (explicit hit sets: the cull is not under test)
    hit_all = [torch.arange(R), torch.arange(R)]
    cx = model._setup({**gin, "hit_index": hit_all}, -1, False)
    torch.manual_seed(1234)                          # the same draws on every rank
    draws = train.make_draws(model, cx, None)
    z_full, it_full = [], []
    for n, p in enumerate(cx["persons"]):
        z, iters, _ = model._sample_person(cx, n, p, draws["person"][p])
        z_full.append(z.clone())
        it_full.append(int(iters.max()))
    torch.cuda.synchronize()

    # ---- my share: rank 0 takes a corner block of the image (rays that miss both bodies: their error bound is met at once),
    #      the other ranks deal the rest -- so that rank 0's OWN vote stops earlier than the call's
    H, W = 16, 32
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    corner = ((yy < 4) & (xx < 8)).reshape(-1)
    rest = torch.nonzero(~corner).flatten()
    ids = torch.nonzero(corner).flatten() if rank == 0 else rest[(rank - 1)::max(world - 1, 1)]
    sub = dict(gin)
    sub["uv"] = gin["uv"][:, ids.cuda()].contiguous()
    hit_my = [torch.arange(len(ids)), torch.arange(len(ids))]
    cxs = model._setup({**sub, "hit_index": hit_my}, -1, False)
    my = {p: {k: (v[ids.cuda()].contiguous() if k in ("t_rand", "u_final") else v) for k, v in draws["person"][p].items()}
          for p in cx["persons"]}
    import time
    res, wall = {}, {}
    T = model.ray_sampler.max_total_iters
    for vote in (False, True):
        model.sampler_vote_group = True if vote else None
        for rep in range(3):                     # the last repetition is the timed one (same draws: same result)
            model.vote_collectives = 0
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            out = model._sample_persons(cxs, my)      # all persons advance together: ONE vote collective per iteration
            torch.cuda.synchronize()
            wall[vote] = time.perf_counter() - t0
        zs = [out[p][0].clone() for p in cxs["persons"]]
        its = [int(out[p][1].max()) for p in cxs["persons"]]
        res[vote] = (zs, its)
        if vote:
            check(model.vote_collectives == T, f"{model.vote_collectives} vote collectives for {len(zs)} persons x {T} iterations: expected {T}")
    model.sampler_vote_group = None
    print(f"[rank {rank}] sampler of {len(cxs['persons'])} persons, {T} iterations: {1e3 * wall[False]:.2f} ms without the vote, "
          f"{1e3 * wall[True]:.2f} ms with it = {1e3 * (wall[True] - wall[False]) / T:.3f} ms per iteration for the "
          f"{dist.get_backend()} MAX all-reduce of {len(cxs['persons'])} flags (blocking, host-visible)", flush=True)
    for n, p in enumerate(cx["persons"]):
        same_vote = torch.equal(res[True][0][n], z_full[n][ids.cuda()])
        same_novote = torch.equal(res[False][0][n], z_full[n][ids.cuda()])
        print(f"[rank {rank}] person {p}: sampler iterations all-rays {it_full[n]}, my share without the vote {res[False][1][n]}, "
              f"with the vote {res[True][1][n]}; depths identical to the single-process rows: with vote {same_vote}, without "
              f"{same_novote}", flush=True)
        check(same_vote, f"person {p}: depths with the all-reduced vote differ from the single-process sampler")
        check(res[True][1][n] == it_full[n], f"person {p}: iteration count with the vote")
    # the test must be able to fail: on at least one rank and person the local vote stops earlier than the global one
    early = torch.tensor([int(any(res[False][1][n] < it_full[n] for n in range(len(it_full))))])
    dist.all_reduce(early, op=dist.ReduceOp.MAX)
    check(int(early) == 1, "no rank converged earlier than the whole call: the scene does not exercise the vote")
    flag = torch.tensor([0 if ok else 1])
    dist.all_reduce(flag)
    dist.destroy_process_group()
    sys.exit(1 if int(flag) else 0)


if __name__ == "__main__":
    main()
