"""world_size-2 gloo test of the ray-sharding helpers used by the multi-GPU render path (runs on CPU)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multiply_amd import parallel


def test_shard_bounds_partition_and_alignment():
    for n, world, group in [(262144, 8, 512), (1000, 2, 512), (513, 4, 512), (512, 2, 512), (7, 3, 2)]:
        b = parallel.shard_bounds(n, world, group)
        assert b[0][0] == 0 and b[-1][1] == n
        for (s0, e0), (s1, e1) in zip(b, b[1:]):
            assert e0 == s1
        for s, e in b:
            assert s % group == 0 or s == n
        ngroups = [-(-(e - s) // group) for s, e in b]      # balanced to within one group
        assert max(ngroups) - min(ngroups) <= 1


def _worker(rank, world, port, n, group, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    inp = {"uv": torch.arange(n * 2, dtype=torch.float32).reshape(1, n, 2), "pose": torch.eye(4)[None]}
    sub, (s, e) = parallel.shard_input(inp, rank, world, group)
    assert sub["uv"].shape[1] == e - s and sub["pose"] is inp["pose"]
    local = sub["uv"][0].sum(-1, keepdim=True) * 2.0          # a per-ray "render"
    full = parallel.gather_rays(local, n, world, group)
    want = inp["uv"][0].sum(-1, keepdim=True) * 2.0
    ok = torch.equal(full, want)
    tmax = parallel.max_over_ranks(1.0 + rank, "cpu")
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, tmax))


@pytest.mark.parametrize("n,group", [(1300, 512), (1024, 512)])
def test_two_rank_gather(n, group):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, group, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(abs(t - 2.0) < 1e-6 for _, _, t in res)


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    unused = torch.nn.Parameter(torch.ones(4))                    # a parameter that received no gradient on this rank
    params = list(net.parameters()) + [unused]
    x = torch.full((2, 5), float(rank + 1))
    net(x).sum().backward()
    mine = [p.grad.clone() for p in net.parameters()]
    ar = parallel.GradientAllReduce(params)
    flat = ar()
    # expected: mean over ranks of the local gradients (inputs differ per rank by a known factor)
    gathered = [None] * world
    dist.all_gather_object(gathered, [g.numpy() for g in mine])
    ok = True
    for i, p in enumerate(net.parameters()):
        want = sum(torch.tensor(g[i]) for g in gathered) / world
        ok = ok and torch.allclose(p.grad, want, atol=1e-6)
    ok = ok and float(unused.grad.abs().sum()) == 0.0 and flat.numel() == sum(p.numel() for p in params)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_gradient_allreduce_at_world_size_one_copies_nothing_and_says_so():
    """world == 1 (no process group): nothing to average -- __call__ returns None instead of a stale flat buffer, the gradients stay
    where they are, and flat_gradients() builds the flat vector on request (zeros for parameters without a gradient)"""
    torch.manual_seed(0)
    net = torch.nn.Linear(4, 3)
    unused = torch.nn.Parameter(torch.ones(2))
    net(torch.ones(2, 4)).sum().backward()
    ar = parallel.GradientAllReduce(list(net.parameters()) + [unused])
    assert ar() is None and unused.grad is None
    flat = ar.flat_gradients()
    want = torch.cat([net.weight.grad.reshape(-1), net.bias.grad.reshape(-1), torch.zeros(2)])
    assert torch.equal(flat, want)


def test_two_rank_gradient_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    groups = [[torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5))],       # "person 0"
              [torch.nn.Parameter(torch.zeros(2, 2))],                                            # "person 1"
              []]                                                                                  # an empty group: no collective
    grads = {}
    sync = parallel.BucketedGradientSync()
    for gi, params in enumerate(groups):             # groups retire one after the other, as the adjoint sweep finishes them
        gs = [torch.full(p.shape, float((rank + 1) * (gi + 1) * (i + 1)), dtype=torch.float64 if i else torch.float32)
              for i, p in enumerate(params)]
        for p, g in zip(params, gs):
            grads[id(p)] = g
        sync.retire(params, gs)
    extra = torch.nn.Parameter(torch.zeros(2))       # a gradient that never went through a bucket stays as it is
    grads[id(extra)] = torch.full((2,), 5.0 + rank)
    pending = len(sync.pending)
    out = sync.finish(grads)
    ok = pending == 2 and sync.pending == []
    for gi, params in enumerate(groups):
        for i, p in enumerate(params):
            want = torch.full(p.shape, 1.5 * (gi + 1) * (i + 1))          # mean over the two ranks of (rank + 1) * ...
            ok = ok and out[id(p)].shape == p.shape and torch.allclose(out[id(p)].float(), want, atol=1e-6)
            ok = ok and out[id(p)].dtype == (torch.float64 if i else torch.float32)        # handed back in the gradient's dtype
    ok = ok and torch.equal(out[id(extra)], torch.full((2,), 5.0 + rank))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_bucketed_gradient_sync_averages_every_retired_group():
    """parallel.BucketedGradientSync as TrainGraph.backward drives it: one asynchronous all-reduce per retired group, finish()
    waits and hands the AVERAGED gradients back by parameter identity (shape and dtype kept), untouched entries pass through"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _sharded_sync_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)

    class M(torch.nn.Module):          # the attribute surface PersonShardedGradSync touches
        def __init__(self):
            super().__init__()
            self.bg_implicit_network = torch.nn.Linear(4, 3)
            self.bg_rendering_network = torch.nn.Linear(3, 2)
            self.frame_latent_encoder = torch.nn.Embedding(5, 3)
            self.foreground_implicit_network_list = torch.nn.ModuleList([torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)])
    m = M()
    # partial gradients of the shared parameters (each rank's ray slice); the person networks are rank-private
    for i, p in enumerate(list(m.bg_implicit_network.parameters()) + list(m.bg_rendering_network.parameters())):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    if rank == 0:
        m.frame_latent_encoder.weight.grad = torch.ones(5, 3)          # rank 1 has none: treated as zeros
    mine = m.foreground_implicit_network_list[rank]
    mine.weight.grad = torch.full((2, 2), 7.0 + rank)
    parallel.PersonShardedGradSync(m)()
    ok = True
    for i, p in enumerate(list(m.bg_implicit_network.parameters()) + list(m.bg_rendering_network.parameters())):
        ok = ok and torch.equal(p.grad, torch.full_like(p, 3.0 * (i + 1)))           # SUM over the two ranks
    ok = ok and torch.equal(m.frame_latent_encoder.weight.grad, torch.ones(5, 3))
    ok = ok and torch.equal(mine.weight.grad, torch.full((2, 2), 7.0 + rank))        # untouched
    ok = ok and m.foreground_implicit_network_list[1 - rank].weight.grad is None     # the other rank's person: none
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_person_sharded_grad_sync_sums_only_the_shared_parameters():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sharded_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_lattice_deal_partitions_the_frame_and_balances_a_body_shaped_cost_map():
    """parallel.shard_indices_interleaved: every ray exactly once, whole convergence groups only, equal group counts, and --
    on a 512x512 frame in 8x8-tile order whose cost is concentrated in two vertical body-shaped blobs (body rays ~25x
    background rays) -- per-rank costs within 3 % of each other for every world size up to 8, where a plain round robin of
    the groups gives the two outer ranks of 8 background only."""
    import numpy as np
    import torch
    from multiply_amd import parallel
    H = W = 512
    yy, xx = np.mgrid[:H, :W]
    cost_px = np.ones((H, W))
    for cx in (200, 312):
        cost_px += 25.0 * ((((xx - cx) / 70.0) ** 2 + ((yy - 270) / 190.0) ** 2) < 1)
    order = np.arange(H * W).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)      # bench.py's tile order
    cost = torch.from_numpy(cost_px.reshape(-1)[order])
    for world in range(1, 9):
        shards = parallel.shard_indices_interleaved(H * W, world, 512, groups_per_row=8)
        allids = torch.cat(shards)
        assert allids.numel() == H * W and torch.equal(allids.sort().values, torch.arange(H * W))
        assert all(bool((s.reshape(-1, 512)[:, 0] % 512 == 0).all()) and bool((s.reshape(-1, 512).diff(dim=1) == 1).all()) for s in shards)
        counts = [s.numel() // 512 for s in shards]
        assert max(counts) - min(counts) <= 3
        load = torch.stack([cost[s].sum() for s in shards])
        assert float(load.max() / load.mean()) < 1.03, (world, load.tolist())
    plain = [cost[(torch.arange(H * W) // 512) % 8 == r].sum() for r in range(8)]
    assert float(max(plain) / (sum(plain) / 8)) > 2.0                                           # the stripe problem it avoids
    # without the hint the deal is skewed by the world size: still a partition into whole groups
    shards = parallel.shard_indices_interleaved(5000, 3, 64)
    assert torch.equal(torch.cat(shards).sort().values, torch.arange(5000))


def _interleaved_worker(rank, world, port, n, group, gpr, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    inp = {"uv": torch.arange(n * 2, dtype=torch.float32).reshape(1, n, 2), "pose": torch.eye(4)[None]}
    sub, ids = parallel.shard_input_interleaved(inp, rank, world, group, gpr)
    assert sub["uv"].shape[1] == len(ids) and torch.equal(sub["uv"][0], inp["uv"][0][ids])
    ok = True
    for _ in range(2):                                               # second call: the cached plan
        local = torch.cat([sub["uv"][0].sum(-1, keepdim=True) * 2.0, sub["uv"][0] + 1.0], dim=1)     # three "outputs" in one tensor
        full = parallel.gather_rays_interleaved(local, n, world, group, gpr)
        want = torch.cat([inp["uv"][0].sum(-1, keepdim=True) * 2.0, inp["uv"][0] + 1.0], dim=1)
        ok = ok and torch.equal(full, want)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("n,group,gpr,world", [(4096, 512, 4, 2), (1300, 64, None, 3), (2048, 512, 1, 2)])
def test_interleaved_shards_gather_back_in_ray_order(n, group, gpr, world):
    """shard_input_interleaved + gather_rays_interleaved over gloo: one collective reassembles several outputs in the frame's
    ray order, also when the ranks hold different numbers of rays (ragged last group, 3 ranks)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_interleaved_worker, args=(r, world, port, n, group, gpr, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _stage_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    # ---- tensors whose number / shapes / dtypes only the source knows
    g = torch.Generator().manual_seed(3)
    src_list = [torch.randn(1, 1234, 3, generator=g), torch.randint(0, 1234, (2466, 3), generator=g), torch.rand(2, 17, 19, generator=g) > 0.5,
                torch.zeros(0, 3), torch.randn(5, generator=g).double(), torch.randint(0, 100, (2, 27, 2), generator=g).to(torch.int32)]
    got = parallel.broadcast_tensors(src_list if rank == 1 else None, src=1)
    ok = ok and len(got) == len(src_list) and all(a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b) for a, b in zip(got, src_list))

    # ---- the every-20-epochs stage: rank 0 "extracts" (stand-in extractor), every rank ends up with the same three lists
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.density = torch.nn.Module()
            self.density.beta = torch.nn.Parameter(torch.tensor(0.1))
            self.mesh_v_cano_list, self.mesh_f_cano_list, self.mesh_face_vertices_list = [], [], []
    m = M()
    calls = []

    def extract(model):
        calls.append(rank)
        gg = torch.Generator().manual_seed(11)
        vs = [torch.randn(1, n, 3, generator=gg) for n in (700, 913)]
        fs = [torch.randint(0, n, (f, 3), generator=gg) for n, f in ((700, 1396), (913, 1822))]
        return vs, fs
    vs, fs = parallel.refresh_canonical_meshes_broadcast(m, extract=extract)
    ok = ok and calls == ([0] if rank == 0 else [])                                  # only the source ran the extractor
    want_v, want_f = extract(None)
    ok = ok and all(torch.equal(a, b) for a, b in zip(m.mesh_v_cano_list, want_v)) and all(torch.equal(a, b) for a, b in zip(m.mesh_f_cano_list, want_f))
    ok = ok and [tuple(t.shape) for t in m.mesh_face_vertices_list] == [(1, 1396, 3, 3), (1, 1822, 3, 3)]
    ok = ok and torch.equal(m.mesh_face_vertices_list[1][0], want_v[1][0][want_f[1]])

    # ---- the every-50-epochs stage: masks, depth maps, key points
    def produce(model, inputs):
        gg = torch.Generator().manual_seed(5)
        depth = [torch.rand(24, 32, generator=gg) for _ in range(2)]
        return torch.stack([d > 0.5 for d in depth]), depth, torch.randint(0, 32, (2, 27, 2), generator=gg).to(torch.int32)
    masks, depth, kps = parallel.frame_instance_masks_broadcast(m, {}, True, produce=produce)
    wm, wd, wk = produce(None, None)
    ok = ok and masks.dtype == torch.bool and torch.equal(masks, wm) and all(torch.equal(a, b) for a, b in zip(depth, wd)) and torch.equal(kps, wk)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_epoch_stage_products_are_broadcast_from_the_producing_rank():
    """frame-sharded data parallelism: canonical-mesh refresh and instance masks run on ONE rank, their
    products -- tensors of data-dependent number, shape and dtype -- reach every rank bit for bit (gloo, 2 ranks)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_stage_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _exchange_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_slice, width = 5, 7
    # rank g's dense block: rows = all rays of the call (slice-major), value encodes (producer rank, ray, column)
    dense = (1000.0 * rank + torch.arange(world * n_slice, dtype=torch.float32)[:, None] * 10 + torch.arange(width)[None, :]).contiguous()
    a2a = parallel._exchange_by_rays(dense, world, True)        # the RCCL branch: all_to_all_single (gloo implements it for CPU tensors)
    gat = parallel._exchange_by_rays(dense, world, False)       # the fallback: all_gather + select
    want = torch.stack([1000.0 * g + (torch.arange(rank * n_slice, (rank + 1) * n_slice, dtype=torch.float32)[:, None] * 10
                                      + torch.arange(width)[None, :]) for g in range(world)], 0)
    ok = a2a.shape == (world, n_slice, width) and torch.equal(a2a, want) and torch.equal(gat, want)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("world", [2, 3])
def test_person_sharded_exchange_all_to_all_branch(world):
    """parallel._exchange_by_rays: the all_to_all_single branch (what runs over RCCL) and the all_gather fallback deliver the same
    [producer rank][my ray slice] blocks -- 'all rays of my persons' becomes 'all persons of my rays'"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _hybrid_worker(rank, world, port, slots, q):
    """hybrid mode (ray shards x person teams) on CPU: the teams' process groups, the exchange INSIDE a team (both branches) and
    the image gather, with a stand-in 'renderer' whose value encodes (ray, person)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = world // slots
    team, shard, slot = parallel.hybrid_teams(slots, shards)
    ok = dist.get_world_size(team) == slots and dist.get_rank(team) == slot and shard == rank // slots
    n, group, P = 96, 8, 4                                   # 12 convergence groups dealt to the ray shards
    inp = {"uv": torch.arange(n * 2, dtype=torch.float32).reshape(1, n, 2)}
    sub, ids = parallel.shard_input_interleaved(inp, shard, shards, group)
    # every rank "evaluates" its persons {p : p % slots == slot} for ALL rays of its shard ...
    R = sub["uv"].shape[1]
    n_slice = (R + slots - 1) // slots
    mine = [p for p in range(P) if p % slots == slot]
    per_person = []
    for j in range((P + slots - 1) // slots):
        dense = torch.zeros(n_slice * slots, 2)
        if j < len(mine):
            dense[:R, 0] = ids.float() * 10 + mine[j]        # value = (ray id, person)
            dense[:R, 1] = 1.0
        a = parallel._exchange_by_rays(dense, slots, True, team)
        b = parallel._exchange_by_rays(dense, slots, False, team)
        ok = ok and torch.equal(a, b)
        per_person.append(a)
    # ... and after the exchange holds ALL persons of its own ray slice of the shard
    s0 = slot * n_slice
    n_my = max(0, min(R, s0 + n_slice) - s0)
    my_ids = ids[s0:s0 + n_my]
    local = torch.zeros(n_my, P)
    for p in range(P):
        blk = per_person[p // slots][p % slots][:n_my]
        ok = ok and bool((blk[:, 1] == 1.0).all()) and torch.equal(blk[:, 0], my_ids.float() * 10 + p)
        local[:, p] = blk[:, 0]
    full = parallel.gather_hybrid(local, my_ids, n)
    want = torch.arange(n, dtype=torch.float32)[:, None] * 10 + torch.arange(P, dtype=torch.float32)[None, :]
    ok = ok and torch.equal(full, want)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("world,slots", [(4, 2), (6, 3)])
def test_hybrid_teams_exchange_inside_a_team_and_gather(world, slots):
    """SURVEY.md section 8e ("4 person-groups x 2 ray-shards" on 8 GPUs), on gloo at 2 x 2 and 2 x 3: every rank's team is its ray
    shard's person slots, the person exchange stays inside the team, and the gathered image is in the frame's ray order"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_hybrid_worker, args=(r, world, port, slots, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
