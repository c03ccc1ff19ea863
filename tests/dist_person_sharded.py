"""2-rank check of parallel.render_person_sharded against the single-process render (run by tests/test_parallel_gpu.py
through torch.distributed.run; both ranks may share one GPU: backend gloo)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import parallel            # noqa: E402
from tests.test_render_gpu import build      # noqa: E402


def main():
    backend = os.environ.get("MP_DIST_BACKEND", "gloo")     # "nccl" (= RCCL): one GPU per rank, real device collectives
    dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"              # where the small control tensors of the collectives live
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(rank % torch.cuda.device_count())
    model, oracle, inp = build(P=int(os.environ.get("MP_TEST_PERSONS", "2")), H=16, W=16)
    R = inp["uv"].shape[1]
    gin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    model.convergence_group = R // world                      # groups must not straddle the ray slices
    whole = model(gin)
    torch.cuda.synchronize()
    part, (s0, s1) = parallel.render_person_sharded(model, gin)
    ok = True
    for k in ("rgb_values", "acc_map", "acc_person_list", "normal_values", "fg_rgb_values"):
        a, b = part[k], whole[k][s0:s1]
        same = torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
        d = (torch.nan_to_num(a) - torch.nan_to_num(b)).abs().max().item()
        print(f"[rank {rank}] {k}: rays {s0}:{s1} max |sharded - single| = {d:.3e} identical={same}", flush=True)
        ok = ok and d < 1e-6
    full = parallel.gather_rays(part["rgb_values"], R, world, R // world)
    ok = ok and (torch.nan_to_num(full) - torch.nan_to_num(whole["rgb_values"])).abs().max().item() < 1e-6
    flag = torch.tensor([1.0 if ok else 0.0], device=cdev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() > 0.5 else 1)


if __name__ == "__main__":
    main()
