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


_PLANS = {}


def _interleaved_plan(n_rays, world, group, groups_per_row, device="cpu"):
    """Cached per (frame size, world, group, hint, device): the ascending ray ids of every rank, and for the gather the map
    ray -> row of the concatenated padded per-rank buffers.  Built once -- the frame loop calls this every frame, and a
    quarter-million-ray index computation per call costs more host time than an eighth of a frame costs GPU time."""
    key = (int(n_rays), int(world), int(group), int(groups_per_row or 0), str(device))
    plan = _PLANS.get(key)
    if plan is None:
        ids = torch.arange(n_rays)
        gid = ids // group
        c = world if not groups_per_row else int(groups_per_row)
        owner = (gid % c + gid // c) % world
        order = torch.argsort(owner, stable=True)                      # ranks one after the other, ray ids ascending inside
        counts = torch.bincount(owner, minlength=world).tolist()
        shards = list(torch.split(order, counts))
        width = max(counts) if counts else 0
        row_of_ray = torch.empty(n_rays, dtype=torch.long)
        for r, sh in enumerate(shards):
            row_of_ray[sh] = r * width + torch.arange(len(sh))
        plan = dict(shards=[sh.to(device) for sh in shards], width=width, row_of_ray=row_of_ray.to(device))
        _PLANS[key] = plan
    return plan


def shard_indices_interleaved(n_rays, world, group, groups_per_row=None):
    """Load-balanced sharding of ONE frame (body rays cost ~30x background-only rays, so contiguous slices
    of the image are badly balanced): the frame's convergence groups (`group` consecutive rays -- with the tile-ordered
    ray list of bench.py a 64x8-pixel block) are dealt out on a diagonal lattice: with `groups_per_row` = C groups per
    band of the image, group g = (band, column) = (g // C, g % C) goes to rank (band + column) % world, so every rank
    visits every column and every band equally often for ANY world size (a plain g % world hands each rank whole vertical
    stripes whenever C is a multiple of the world size -- 8 ranks on a 512-wide frame: the outer ranks would render
    background only).  Without the hint the deal is skewed by the world size, (g + g // world) % world, which is the same
    lattice when C == world.  Measured on the 512x512 two-person frame (tools/shard_latency.py): per-rank times within
    5 % of each other at 2, 4 and 8 ranks.
    Cutting at whole groups keeps the sampler's vote (ray_sampler.py:137) per group, so every pixel equals the
    single-process render with convergence_group = group.  Returns one ascending long tensor of ray ids per rank (cached)."""
    return _interleaved_plan(n_rays, world, group, groups_per_row)["shards"]


def shard_input_interleaved(inp, rank, world, group, groups_per_row=None):
    """The rank's interleaved share of a Multiply.forward input dict and the ray ids it holds."""
    idx = shard_indices_interleaved(inp["uv"].shape[1], world, group, groups_per_row)[rank]
    out = dict(inp)
    out["uv"] = inp["uv"][:, idx.to(inp["uv"].device)].contiguous()
    return out, idx


def gather_rays_interleaved(local, n_rays, world, group, groups_per_row=None):
    """all_gather of per-ray outputs of an interleaved sharding into (n_rays, ...) in the frame's ray order (every rank gets
    the whole image): ONE collective of (n_rays / world) rows per rank -- concatenate the outputs to keep along the last
    dimension before calling -- and one cached index to put the rows back in ray order."""
    plan = _interleaved_plan(n_rays, world, group, groups_per_row, local.device)
    width = plan["width"]
    if local.shape[0] == width:
        pad = local.contiguous()
    else:
        pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    flat = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather(list(flat.split(width)), pad)
    return flat[plan["row_of_ray"]]


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
        """world > 1: gradients -> flat buffer (one concat) -> all_reduce -> averaged gradients written back (one foreach copy);
        returns the flat buffer of AVERAGED gradients.  world == 1: nothing to average -- the parameters' .grad are final as they
        are, nothing is copied (it cost ~40 zero fills of never-touched gradients + a 9 MB concat per iteration in the single-GPU
        bench line) and None is returned: a caller that wants the flat vector (gradient norm, clipping) calls flat_gradients()."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        if world == 1:
            return None
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

    def flat_gradients(self):
        """the current .grad of every parameter as ONE flat fp32 vector (zeros where a parameter has no gradient), at any world size"""
        torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in self.params], out=self.flat)
        return self.flat


class BucketedGradientSync:
    """The same averaging, OVERLAPPED with the backward pass: the hand-written adjoint sweep
    (multiply_amd/train.py TrainGraph.backward) finishes one person's networks after the other, then the background nets.  Set
    as `model.grad_bucket_sync`, every retired group of gradients is copied into its own flat bucket and its all-reduce is
    launched at once with async_op=True -- on RCCL's own stream, beside the next person's backward kernels -- and the sweep's
    last act is to wait for the buckets and hand the AVERAGED gradients to autograd (p.grad is final when loss.backward()
    returns; no GradientAllReduce call afterwards).  Buckets: one per person (SDF + colour net, ~3.3 MB), one for the
    background nets, the frame code and density.beta (~2.3 MB)."""

    def __init__(self, group=None):
        self.group = group
        self.pending = []

    def retire(self, params, grads):
        """params / grads: the tensors of one retired group (same order on every rank)"""
        if not params:
            return
        flat = torch.cat([g.reshape(-1).float() for g in grads])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append((work, flat, list(params), [g.shape for g in grads]))

    def finish(self, grads_by_id):
        """waits for every bucket; -> grads_by_id with the retired entries replaced by their averages"""
        world = dist.get_world_size(self.group)
        for work, flat, params, shapes in self.pending:
            work.wait()
            flat.mul_(1.0 / world)
            o = 0
            for p, sh in zip(params, shapes):
                n = int(torch.Size(sh).numel())
                grads_by_id[id(p)] = flat[o:o + n].reshape(sh).to(grads_by_id[id(p)].dtype)
                o += n
        self.pending = []
        return grads_by_id


# ======================================================================================================================
# Person-sharded rendering (BASELINE.json configs[3]): rank g owns the networks' work of persons
# {p : p % world == g} for ALL rays of the call, then ONE exchange step turns "all rays of my persons" into "all persons
# of my rays", and compositing + background are ray-partitioned.
# ======================================================================================================================
def _exchange_by_rays(dense, world, backend_alltoall, group=None):
    """dense [world * n_slice, ...] (rows = rays of the whole call, slice-major) -> [world, n_slice, ...]: block g holds
    rank g's rows for MY ray slice.  all_to_all over RCCL; all_gather + select where the backend has no all_to_all.
    group: the process group the exchange runs in (None: the world); world = its size."""
    n_slice = dense.shape[0] // world
    if backend_alltoall:
        out = torch.empty_like(dense)
        dist.all_to_all_single(out, dense.contiguous(), group=group)
        return out.reshape(world, n_slice, *dense.shape[1:])
    rank = dist.get_rank(group)
    bufs = [torch.empty_like(dense) for _ in range(world)]
    dist.all_gather(bufs, dense.contiguous(), group=group)
    return torch.stack([b.reshape(world, n_slice, *dense.shape[1:])[rank] for b in bufs], 0)


def render_person_sharded(model, input, canonical_pose=False, group=None, exchange_events=None):
    """Eval-mode Multiply.forward with the persons sharded over the ranks (of `group`; None: all ranks).  Every rank returns the
    output dict for ITS ray slice [rank * ceil(R / world), ...) (use gather_rays to assemble the image).  Identical results to
    the single-process call with convergence groups that do not straddle a slice (the sampler's vote is per person and per
    group).  exchange_events (a list): (start, end) timing events around every exchange are appended (bench.py --mode person)."""
    import ctypes as C
    from . import hip
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    L = hip.lib()
    st = hip.stream()
    P = input["smpl_trans"].shape[1]
    mine = [p for p in range(P) if p % world == rank]
    n_local = (P + world - 1) // world                       # persons per rank, padded
    with torch.no_grad():
        parts = model._forward_eval(input, mine, canonical_pose, composite=False) if mine else {}
    last = model._last if mine else None
    dev = model.density.beta.device
    R = input["uv"].shape[1]
    n_slice = (R + world - 1) // world
    Rpad = n_slice * world
    rs = model.ray_sampler
    NZ = rs.N_samples + rs.N_samples_extra + 2
    S = NZ - 1
    width = NZ + S * 7 + 1                                    # z, sdf, rgb3, nrm3 per sample, + hit flag
    f32 = dict(dtype=torch.float32, device=dev)
    recv = []
    for j in range(n_local):                                  # one exchange per local person slot
        dense = torch.zeros(Rpad, width, **f32)
        if j < len(mine):
            d = parts[mine[j]]
            n, rows = d["n_hit"], d["hit_index"][:d["n_hit"]].long()
            dense[rows, :NZ] = d["z"][:n]
            dense[rows, NZ:NZ + S] = d["sdf"][:n * S].reshape(n, S)
            dense[rows, NZ + S:NZ + 4 * S] = d["rgb"][:n * S].reshape(n, 3 * S)
            dense[rows, NZ + 4 * S:NZ + 7 * S] = d["nrm"][:n * S].reshape(n, 3 * S)
            dense[rows, -1] = 1.0
        if exchange_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        recv.append(_exchange_by_rays(dense, world, dist.get_backend(group) == "nccl", group))      # [world, n_slice, width]
        if exchange_events is not None:
            e1.record()
            exchange_events.append((e0, e1))
    # my ray slice, all persons: person p = j * world + g
    s0 = rank * n_slice
    n_my = max(0, min(R, s0 + n_slice) - s0)
    persons = list(range(P))
    z_l, sdf_l, rgb_l, nrm_l, inv_l = [], [], [], [], []
    for p in persons:
        blk = recv[p // world][p % world][:n_my]
        hit = blk[:, -1] > 0.5
        inv = torch.where(hit, torch.arange(n_my, device=dev, dtype=torch.int32), torch.full((n_my,), -1, dtype=torch.int32, device=dev))
        z_l.append(blk[:, :NZ].contiguous()); sdf_l.append(blk[:, NZ:NZ + S].contiguous())
        rgb_l.append(blk[:, NZ + S:NZ + 4 * S].contiguous()); nrm_l.append(blk[:, NZ + 4 * S:NZ + 7 * S].contiguous())
        inv_l.append(inv.contiguous())
    table = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev)
    tabs = [table(t) for t in (inv_l, z_l, sdf_l, rgb_l, nrm_l)]
    beta = (model.density.beta.detach().abs() + model.density.beta_min).reshape(1).float().contiguous()
    # rays of my slice + background for them
    uv = input["uv"].to(dev).float().reshape(-1, 2)[s0:s0 + n_my].contiguous()
    K = input["intrinsics"].to(dev).float().reshape(16).contiguous()
    pose = input["pose"].to(dev).float().reshape(16).contiguous()
    dirs = torch.empty(n_my, 3, **f32); far = torch.empty(n_my, **f32)
    hip.check(L.mp_ray_setup(hip.ptr(uv), hip.ptr(K), hip.ptr(pose), n_my, C.c_float(model.sdf_bounding_sphere),
                             hip.ptr(dirs), hip.ptr(far), st), "mp_ray_setup")
    bg_rgb = None
    if input.get("idx", None) is not None:
        key = "image_id" if "image_id" in input else "idx"
        code = model.frame_latent_encoder.weight.detach()[int(torch.as_tensor(input[key]).reshape(-1)[0])]
        t = torch.linspace(0.0, 1.0, rs.N_samples_inverse_sphere, device=dev)
        z_bg = torch.flip(t * (1.0 / rs.scene_bounding_sphere), dims=[0]).contiguous()
        bg_rgb = hip.background(model.bg_implicit_network, model.bg_rendering_network, dirs,
                                pose.reshape(4, 4)[:3, 3].contiguous(), z_bg, code, radius=model.sdf_bounding_sphere)
    out = {k: torch.empty(n_my, 3, **f32) for k in ("rgb_values", "fg_rgb_values", "normal_values")}
    acc_map = torch.empty(n_my, **f32); acc_person = torch.empty(n_my, P, **f32); bg_T = torch.empty(n_my, **f32)
    hip.check(L.mp_composite(n_my, P, NZ, *[hip.ptr(t) for t in tabs], hip.ptr(beta),
                             hip.ptr(bg_rgb) if bg_rgb is not None else None, hip.ptr(out["rgb_values"]),
                             hip.ptr(out["fg_rgb_values"]), hip.ptr(out["normal_values"]), hip.ptr(acc_map),
                             hip.ptr(acc_person), hip.ptr(bg_T), st), "mp_composite")
    torch.cuda.synchronize()                                  # the pointer tables / blocks above must outlive the launch
    out.update(acc_map=acc_map, acc_person_list=acc_person)
    return out, (s0, s0 + n_my)


# ======================================================================================================================
# HYBRID rendering (last sentence of the person-sharded row; BASELINE.json configs[3] on an 8-GPU node:
# "4 person-groups x 2 ray-shards"): world = ray_shards x person_slots.  The frame's convergence groups are dealt to the
# ray shards on the diagonal lattice (shard_indices_interleaved), and every ray shard is rendered by a TEAM of person_slots
# ranks with render_person_sharded inside the team's own process group: persons {p : p % person_slots == slot} evaluated
# for the shard's rays, one all_to_all inside the team, compositing + background partitioned by the shard's ray slices.
# No collective crosses teams until the caller assembles the image.  rank r -> (shard r // person_slots, slot r % person_slots).
# ======================================================================================================================
_TEAMS = {}


def hybrid_teams(person_slots, ray_shards):
    """The team (process group) of this rank for a world of ray_shards x person_slots ranks.  Collective: EVERY rank creates
    every team's group, in the same order (torch.distributed.new_group's contract).  Cached per (world, split)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    assert world == person_slots * ray_shards, f"world {world} != {ray_shards} ray shards x {person_slots} person slots"
    key = (world, person_slots, ray_shards)
    if key not in _TEAMS:
        teams = [dist.new_group([s * person_slots + k for k in range(person_slots)]) for s in range(ray_shards)]
        _TEAMS[key] = teams
    return _TEAMS[key][rank // person_slots], rank // person_slots, rank % person_slots


def render_hybrid(model, input, person_slots, ray_shards, group_size, groups_per_row=None, canonical_pose=False,
                  exchange_events=None):
    """Eval-mode Multiply.forward on ray_shards x person_slots ranks.  Returns (out, ray_ids): the output dict of THIS rank's
    rays and their ids in the frame's ray order (ascending).  Bit-identical to the single-process call with
    convergence_group = group_size: the shards are whole convergence groups and the person teams composite exactly the rows
    the single process composites."""
    team, shard, slot = hybrid_teams(person_slots, ray_shards)
    sub, idx = shard_input_interleaved(input, shard, ray_shards, group_size, groups_per_row)
    out, (s0, s1) = render_person_sharded(model, sub, canonical_pose, group=team, exchange_events=exchange_events)
    return out, idx[s0:s1]


def gather_hybrid(local, ray_ids, n_rays):
    """all_gather of a hybrid render's per-ray outputs into (n_rays, ...) in ray order on every rank: ONE collective of the
    padded (rows | ray id) blocks; ranks hold different counts."""
    world = dist.get_world_size()
    n = torch.tensor([local.shape[0]], device=local.device)
    counts = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    width = int(max(int(c) for c in counts))
    flat = local.reshape(local.shape[0], -1).float()
    pad = torch.zeros(width, flat.shape[1] + 1, dtype=torch.float32, device=local.device)
    pad[:local.shape[0], :-1] = flat
    pad[:local.shape[0], -1] = ray_ids.to(local.device).float()          # exact below 2^24 rays
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    full = torch.zeros((n_rays, flat.shape[1]), dtype=torch.float32, device=local.device)
    for b, c in zip(bufs, counts):
        c = int(c)
        full[b[:c, -1].long()] = b[:c, :-1]
    return full.reshape((n_rays,) + tuple(local.shape[1:])).to(local.dtype)


# ======================================================================================================================
# Person-sharded TRAINING: rank g owns the networks and SMPL state of persons {p : p % world == g}.
#   forward : each rank samples + evaluates its persons for all rays; ONE all_gather moves the per-sample rows
#             [ray, z, sdf, rgb, normal] (+ eikonal gradients, in/off-surface flags) of every person to every rank
#             (2.6 MB per person at 512 rays x 161 samples); every rank then composites all 512 rays and evaluates the
#             loss, so the compositing adjoint needs no second exchange; the background branch is sliced by rays and its
#             colours all-gathered (6 KB).
#   backward: the compositing adjoint gives d(sdf, rgb) for all persons on every rank; each rank back-propagates its
#             own persons.  Person networks are rank-private (no collective); density.beta's gradient is identical on
#             all ranks; the background networks' and the frame code's gradients are partial sums over the rank's ray
#             slice -> PersonShardedGradSync (one flat all-reduce, 0.58 M floats).
# ======================================================================================================================
def train_person_sharded(model, input, cond_zero_shit=False, canonical_pose=False, draws=None):
    """Training-mode Multiply.forward with the persons sharded over the ranks; same output dict on every rank."""
    from . import train
    assert model.training
    return train.forward_train(model, input, -1, cond_zero_shit, canonical_pose, draws=draws,
                               shard=(dist.get_world_size(), dist.get_rank()))


class PersonShardedGradSync:
    """After loss.backward() of a person-sharded step: sums the gradients of the parameters every rank holds a partial
    gradient for (background networks, frame latent codes) in ONE flat all-reduce, and leaves alone what is rank-private
    (the person networks; other ranks' copies get no gradient) or already identical (density.beta)."""

    def __init__(self, model):
        self.params = [p for m in (model.bg_implicit_network, model.bg_rendering_network, model.frame_latent_encoder)
                       for p in m.parameters() if p.requires_grad]
        self.flat = None

    def __call__(self):
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        if self.flat is None or self.flat.numel() != n:
            self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        o = 0
        for p in self.params:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            self.flat[o:o + p.numel()] = g.reshape(-1)
            o += p.numel()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].reshape(p.shape).clone()
            o += p.numel()


# ======================================================================================================================
# Frame-sharded data parallelism and the epoch-level stages under it (last paragraph).
#
# One optimiser step of the reference handles ONE frame (multiply_model.py:131-222: batch size 1, 512 pixels).  With N ranks a
# step handles N frames, one per rank: every rank runs forward + loss + backward on its own frame -- the per-frame rows of the
# BodyModelParams tables included -- and ONE flat all-reduce averages the gradients of the shared networks AND of the body-model
# tables (a rank's table gradient is non-zero in its own frame's rows only; the tables are small -- frames x 85 floats per
# person -- so they ride dense in the same flat buffer instead of a sparse exchange).  Every rank then takes the same optimiser
# step on the same averaged gradients: the replicas stay bit-identical without a parameter broadcast.
#
# The stages between epochs -- canonical-mesh extraction every 20 epochs, instance masks / SAM prompts every 50
# (multiply_model.py:489-518) -- are single-GPU work whose PRODUCTS every rank needs: rank `src` runs them, the products (a few
# tensors of data-dependent shape) are broadcast.
# ======================================================================================================================
_DTYPES = [torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int64, torch.int32, torch.int16, torch.int8,
           torch.uint8, torch.bool]


def broadcast_tensors(tensors, src=0, group=None, device=None):
    """A list of tensors whose NUMBER, SHAPES and DTYPES only rank `src` knows (the other ranks pass None) -> the same list on
    every rank.  Two collectives: one int64 header (count; per tensor dtype code, ndim, up to 8 extents), one byte payload."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(tensors)
    rank = dist.get_rank(group)
    MAXT, MAXD = 64, 8
    if rank == src:
        tensors = [t.detach().contiguous() for t in tensors]
        assert len(tensors) <= MAXT and all(t.dim() <= MAXD for t in tensors)
        device = tensors[0].device if tensors and device is None else device
        hdr = torch.zeros(1 + MAXT * (2 + MAXD), dtype=torch.int64)
        hdr[0] = len(tensors)
        for i, t in enumerate(tensors):
            o = 1 + i * (2 + MAXD)
            hdr[o], hdr[o + 1] = _DTYPES.index(t.dtype), t.dim()
            for d, e in enumerate(t.shape):
                hdr[o + 2 + d] = e
    else:
        hdr = torch.zeros(1 + MAXT * (2 + MAXD), dtype=torch.int64)
    if device is None:
        device = torch.device("cpu")
    hdr = hdr.to(device)
    dist.broadcast(hdr, src=src, group=group)
    h = hdr.cpu().tolist()
    metas = []
    for i in range(h[0]):
        o = 1 + i * (2 + MAXD)
        metas.append((_DTYPES[h[o]], tuple(h[o + 2:o + 2 + h[o + 1]])))
    nbytes = [torch.empty(0, dtype=dt).element_size() * int(torch.Size(sh).numel()) for dt, sh in metas]
    # every piece starts on a 16-byte boundary of the payload so that its typed view is aligned
    offs, total = [], 0
    for n in nbytes:
        offs.append(total)
        total += (n + 15) // 16 * 16
    payload = torch.zeros(max(total, 16), dtype=torch.uint8, device=device)
    if rank == src:
        for t, o, n in zip(tensors, offs, nbytes):
            if n:
                payload[o:o + n] = t.to(device).reshape(-1).view(torch.uint8) if t.dtype != torch.bool else t.to(device).reshape(-1).to(torch.uint8)
    dist.broadcast(payload, src=src, group=group)
    out = []
    for (dt, sh), o, n in zip(metas, offs, nbytes):
        raw = payload[o:o + n]
        out.append((raw.to(torch.bool) if dt == torch.bool else raw.view(dt)).reshape(sh).clone())
    return out


def refresh_canonical_meshes_broadcast(model, conds=None, res_up=2, src=0, group=None, extract=None):
    """The every-20-epochs stage under data parallelism (multiply_model.py:497-506): rank `src` re-extracts every person's
    canonical mesh (multiply_amd.mesh.refresh_canonical_meshes: MISE + marching cubes on the device), the vertex and face arrays
    are broadcast, EVERY rank re-assigns mesh_v_cano_list / mesh_f_cano_list / mesh_face_vertices_list (read by the in / off-
    surface flags of the next 20 epochs' training forwards, multiply.py:153-167).  `extract(model)` -> (vertex list, face list)
    replaces the extractor (tests)."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    dev = model.density.beta.device
    items = None
    if rank == src:
        if extract is None:
            from .mesh import refresh_canonical_meshes
            vs, fs = refresh_canonical_meshes(model, conds, res_up)
        else:
            vs, fs = extract(model)
        items = [t for v, f in zip(vs, fs) for t in (v, f)]
    items = broadcast_tensors(items, src=src, group=group, device=dev)
    vs, fs = [t.to(dev) for t in items[0::2]], [t.to(dev) for t in items[1::2]]
    model.mesh_v_cano_list, model.mesh_f_cano_list = vs, fs
    model.mesh_face_vertices_list = [v[0][f][None] for v, f in zip(vs, fs)]
    return vs, fs


def frame_instance_masks_broadcast(model, inputs, use_smpl_mesh, res_up=2, src=0, group=None, produce=None):
    """The every-50-epochs stage (multiply_model.py:741-939, one frame of get_instance_mask): rank `src` rasterises the persons'
    meshes (multiply_amd.mesh_losses.frame_instance_masks), every rank receives the instance masks (P,H,W) bool, the depth maps
    and the 27 projected key points per person -- what the SAM prompts and the data set's mask refresh are built from."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    dev = model.density.beta.device
    items = None
    if rank == src:
        if produce is None:
            from .mesh_losses import frame_instance_masks
            masks, depth, kps = frame_instance_masks(model, inputs, use_smpl_mesh, res_up)
        else:
            masks, depth, kps = produce(model, inputs)
        items = [masks, torch.stack(list(depth), 0), kps]
    masks, depth, kps = broadcast_tensors(items, src=src, group=group, device=dev)
    return masks, list(depth), kps


class FrameShardedTrainer:
    """One data-parallel optimiser step over N frames, one per rank (see the section comment).  `params` = the shared networks'
    parameters AND the BodyModelParams tables that are being optimised; `optimizers` are stepped by every rank on the averaged
    gradients.  step(inputs, targets, frame_idx) -> the loss dict of this rank's frame."""

    def __init__(self, model, body_model_list, loss_fn, optimizers, vote_group=None):
        self.model, self.body, self.loss_fn, self.opts = model, list(body_model_list), loss_fn, list(optimizers)
        params = [p for p in model.parameters() if p.requires_grad]
        params += [p for bm in self.body for p in bm.parameters() if p.requires_grad]
        self.sync = GradientAllReduce(params)
        self.vote_group = vote_group

    def step(self, inputs, targets, frame_idx, sync=True):
        """sync=False: no collective and no optimiser step -- the plain single-process gradient of this frame (tests)"""
        from .mesh_losses import body_model_inputs
        inp = dict(inputs)
        if self.body:
            idx = torch.as_tensor(frame_idx, device=self.model.density.beta.device).reshape(1).long()
            inp["smpl_trans"], inp["smpl_shape"], inp["smpl_pose"] = body_model_inputs(self.body, idx)
            last = torch.clamp(idx - 1, min=0)                                  # multiply_model.py:170-178
            with torch.no_grad():
                _, _, inp["smpl_pose_last"] = body_model_inputs(self.body, last)
        self.model.train()
        out = self.model(inp)
        lo = self.loss_fn(out, targets)
        for p in self.sync.params:
            p.grad = None
        lo["loss"].backward()
        if sync:
            self.sync()                                                          # ONE flat all-reduce: networks + body tables
            for o in self.opts:
                o.step()
        return lo
