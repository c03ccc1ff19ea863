"""4-rank check of parallel.render_hybrid (2 ray shards x 2 person slots; SURVEY.md section 8e: "4 person-groups x 2 ray-shards" for
8 GPUs) against the single-process render: every rank's rays bit-identical, the gathered image identical.  Run by
tests/test_parallel_gpu.py through torch.distributed.run; MP_DIST_BACKEND=nccl (one GPU per rank) runs the team exchange as a
real all_to_all_single over RCCL, the default gloo lets the ranks share the box's GPU(s)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiply_amd import parallel            # noqa: E402
from tests.test_render_gpu import build      # noqa: E402


def main():
    backend = os.environ.get("MP_DIST_BACKEND", "gloo")
    dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(rank % torch.cuda.device_count())
    slots = int(os.environ.get("MP_TEST_SLOTS", "2"))
    shards = world // slots
    P = int(os.environ.get("MP_TEST_PERSONS", "4"))
    model, oracle, inp = build(P=P, H=16, W=32)
    R = inp["uv"].shape[1]
    gin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    G = 32                                                     # convergence group: 16 groups dealt to the ray shards
    model.convergence_group = G
    whole = model(gin)
    torch.cuda.synchronize()
    part, ids = parallel.render_hybrid(model, gin, slots, shards, G, groups_per_row=None)
    ok = True
    for k in ("rgb_values", "acc_map", "acc_person_list", "normal_values", "fg_rgb_values"):
        a, b = part[k], whole[k][ids.to(whole[k].device)]
        same = torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
        print(f"[rank {rank} = shard {rank // slots} slot {rank % slots}] {k}: {len(ids)} rays identical={same}", flush=True)
        ok = ok and same
    dev = "cuda" if backend == "nccl" else "cpu"
    full = parallel.gather_hybrid(part["rgb_values"].to(dev), ids, R)
    ok = ok and torch.equal(torch.nan_to_num(full.cpu(), nan=-7.0), torch.nan_to_num(whole["rgb_values"].cpu(), nan=-7.0))
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() > 0.5 else 1)


if __name__ == "__main__":
    main()
