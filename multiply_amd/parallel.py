"""Ray/frame sharding helpers for one-process-per-GPU rendering (torch.distributed; backend "nccl" = RCCL on ROCm).

The render path needs no data-path collective: weights are replicated, rays are independent once the sampler's
convergence vote (reference ray_sampler.py:137) is taken per `group` of consecutive rays, so shards are cut at
multiples of the group size and every rank's pixels are bit-identical to the single-GPU call.  Collectives are only
used to reassemble an image (all_gather of the per-rank tiles) and to agree on timings (all_reduce MAX).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rays, world, group):
    """Contiguous [start, end) per rank, cut at multiples of `group` rays, balanced to within one group."""
    n_groups = (n_rays + group - 1) // group
    base, extra = divmod(n_groups, world)
    bounds, g0 = [], 0
    for r in range(world):
        g1 = g0 + base + (1 if r < extra else 0)
        bounds.append((min(g0 * group, n_rays), min(g1 * group, n_rays)))
        g0 = g1
    return bounds


def shard_input(inp, rank, world, group):
    """The rank's slice of a Multiply.forward input dict (only `uv` is per-ray)."""
    n = inp["uv"].shape[1]
    s, e = shard_bounds(n, world, group)[rank]
    out = dict(inp)
    out["uv"] = inp["uv"][:, s:e].contiguous()
    return out, (s, e)


def gather_rays(local, n_rays, world, group):
    """all_gather of per-ray outputs (R_local, ...) into (n_rays, ...) in ray order; ranks may hold different counts."""
    bounds = shard_bounds(n_rays, world, group)
    width = max(e - s for s, e in bounds)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:e - s] for b, (s, e) in zip(bufs, bounds)], 0)


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class GradientAllReduce:
    """Data-parallel training: every rank runs forward+backward on its own rays / frames, then ONE
    all-reduce (sum, then 1/world) of a single flat fp32 buffer holding every parameter gradient (~2.2 M floats =
    8.8 MB for two persons), instead of the reference's single-GPU step.  On the 8-GPU xGMI mesh RCCL turns one large
    message into reduce-scatter + all-gather over all links; many small per-tensor collectives would be latency-bound."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        n = sum(self.sizes)
        p0 = self.params[0]
        self.flat = torch.zeros(n, dtype=torch.float32, device=p0.device)

    def __call__(self):
        """gradients -> flat buffer (one concat) -> all_reduce -> averaged gradients written back (one foreach copy)"""
        world = dist.get_world_size() if dist.is_initialized() else 1
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        torch.cat([p.grad.reshape(-1) for p in self.params], out=self.flat)
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / world)
            torch._foreach_copy_([p.grad for p in self.params],
                                 [v.reshape(p.shape) for p, v in zip(self.params, self.flat.split(self.sizes))])
        return self.flat
