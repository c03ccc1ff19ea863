#!/usr/bin/env python
"""Headline benchmark: rays/sec rendering full synthetic 512x512 frames of a 2-person scene on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (Multiply.forward, eval mode, all persons, with background) over one batch of
synthetic input = one full 512x512 frame = 262,144 rays (BASELINE.json configs[1]: 2 persons, 128 importance
samples/ray, half-precision MFMA MLPs: f16 operands, fp32 accumulate).  Inputs (rays' uv, camera, SMPL parameters,
weights) are resident in HBM before the timed region.

N > 1 (BASELINE.json configs[2]): one process per GPU.  `python bench.py --gpus N` without a launcher
re-executes itself under torch.distributed.run; under a launcher it asserts WORLD_SIZE == N.  The headline line is STRONG
scaling: ONE frame per step, its convergence groups (64x8-pixel blocks) dealt round robin to the ranks (body rays cost ~30x
background rays), every rank renders its share and ONE RCCL all_gather per output reassembles the image on every rank, inside
the timed region; time = max over ranks between barriers.  The frame-per-rank weak-scaling throughput (no data-path
collective at all) is reported beside it under "weak".  Training: 512 / N rays per rank + one flat gradient all-reduce.

The JSON line also carries
  roofline     : the dominant kernel's algorithmic FLOP/s (HIP events inside the timed region) vs the dense 16-bit MFMA peak
  cpu_baseline : the CPU oracle (fp32 torch restatement of the reference path) timed on this host on a bounded sample
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

M_IMP, M_REN, M_BGIMP, M_BGREN = 542208, 266496, 532736, 40704      # MACs / point
PEAK_BF16_TFLOPS = 2500.0                                           # MI355X dense bf16 / f16 MFMA (MI355X_MICROARCH.md)


def build_model(n_samples, seed=0, H=512, W=512, P=2, tile=8):
    import warnings
    warnings.filterwarnings("ignore")
    from multiply_amd.config import load_config
    from multiply_amd.multiply import Multiply
    from multiply_amd.synthetic import make_scene, make_smpl_tables
    tables = make_smpl_tables(0)
    sc = make_scene(P, seed=seed, H=H, W=W)
    opt = load_config()
    opt.ray_sampler.N_samples = n_samples
    opt.ray_sampler.N_samples_eval = max(128, n_samples)
    torch.manual_seed(0)
    model = Multiply(opt, sc["smpl_params"][0, :, 76:], smpl_tables=tables).eval()
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    sp = t(sc["smpl_params"])
    uv = sc["uv"]
    if tile and H % tile == 0 and W % tile == 0:
        # ray order inside the batch is the caller's choice (the reference takes any uv list): emit the frame tile by tile
        # (tile x tile pixels contiguous) so that 64 consecutive rays are spatial neighbours -> coherent nearest-vertex
        # searches; a convergence group of 512 rays is then a 64x8-pixel block instead of a 512x1 row.
        idx = np.arange(H * W).reshape(H // tile, tile, W // tile, tile).transpose(0, 2, 1, 3).reshape(-1)
        uv = uv[:, idx]
    inp = dict(uv=t(uv), intrinsics=t(sc["intrinsics"]), pose=t(sc["pose"]), smpl_params=sp,
               smpl_pose=sp[:, :, 4:76], smpl_shape=sp[:, :, 76:], smpl_trans=sp[:, :, 1:4], idx=torch.tensor([3]))
    return model, inp, tables, sc


def person_mode_renderer(model, gin, mode, world, rank, group, chunk, person_slots, keep):
    """BASELINE.json configs[3]: ONE frame per step on `world` ranks with the PERSONS sharded --
    mode 'person': rank g evaluates persons {p : p % world == g} for all rays, one all_to_all per person slot turns "all rays of
    my persons" into "all persons of my rays", compositing + background ray-partitioned (parallel.render_person_sharded);
    mode 'hybrid': world = ray_shards x person_slots, a team of person_slots ranks per ray shard (parallel.render_hybrid).
    The frame goes through in chunks of `chunk` rays (whole convergence groups; the dense exchange block of a chunk is
    chunk x (NZ + 7 S + 1) floats per person slot), the kept outputs are all-gathered ONCE per frame (parallel.gather_hybrid).
    Returns render() -> {key: (R, 3)} and the event lists [(start, end)] of the exchanges and of the image gather."""
    from multiply_amd import parallel as PX
    R = gin["uv"].shape[1]
    step = max(group * world, (chunk // (group * world)) * group * world)
    ex_evs, gather_evs = [], []
    ray_shards = world // person_slots if mode == "hybrid" else 1

    def render(_gin=None):
        rows, ids, shaded, sdf_evals = [], [], [], []
        for c0 in range(0, R, step):
            sub = dict(gin)
            sub["uv"] = gin["uv"][:, c0:c0 + step].contiguous()
            if mode == "hybrid":
                out, rid = PX.render_hybrid(model, sub, person_slots, ray_shards, group, exchange_events=ex_evs)
                rid = rid.to(gin["uv"].device) + c0
            else:
                out, (s0, s1) = PX.render_person_sharded(model, sub, exchange_events=ex_evs)
                rid = torch.arange(c0 + s0, c0 + s1, device=gin["uv"].device)
            rows.append(torch.cat([out[k] for k in keep], dim=1))
            ids.append(rid)
            shaded += list(model.last_stats.get("n_shaded", []))
            sdf_evals += list(model.last_stats.get("n_sdf_evals", []))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        full = PX.gather_hybrid(torch.cat(rows, 0), torch.cat(ids, 0), R)
        e1.record()
        gather_evs.append((e0, e1))
        return dict(zip(keep, full.split([3] * len(keep), dim=1))), shaded, sdf_evals

    return render, ex_evs, gather_evs


def to_dev(inp):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}


def host_cpu():
    """(threads torch uses, physical cores, model name) of this host"""
    model, cores = "unknown", set()
    try:
        phys, core = None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip() and phys is not None:
                    cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    return torch.get_num_threads(), (len(cores) or None), model


def cpu_baseline(model, inp, tables, sc, n_samples, n_rays=8192, budget_s=330.0, min_rays=4096):
    """Times the CPU oracle on a bounded sample of the same frame: `n_rays` rays = a centred block of image rows that crosses
    both bodies, every pixel of those rows (>= 16 k rays or 3 frames asked, 8 192 taken by default = ~8 min of
    oracle so that the whole bench.py run stays inside the driver's timeout; extrapolated and labelled).  Also returns the
    GPU-vs-oracle pixel error on that sample.
    WALL-TIME GUARD: the oracle walks the sample one convergence group (512 rays) at a time and stops at the first group
    boundary past `budget_s` seconds once `min_rays` rays are done (a slower host must not turn the headline run into a driver
    timeout); the rate, the parity figures and the `sample` label refer to the rays actually processed."""
    from oracle import multiply_oracle as O
    H = W = int(round(np.sqrt(inp["uv"].shape[1])))
    rows = max(1, n_rays // W)
    r0 = H // 2 - rows // 2
    uvx, uvy = inp["uv"][0, :, 0], inp["uv"][0, :, 1]
    sel = torch.nonzero((uvy >= r0) & (uvy < r0 + rows)).flatten()
    sel = sel[torch.argsort(uvy[sel] * W + uvx[sel])]
    sub = dict(inp)
    sub["uv"] = inp["uv"][:, sel]
    # the CPU path does the REFERENCE's work: every ray that meets a person's box is sampled (multiply.py:256-266), without
    # the eval-mode near-body refinement of the device path (which leaves the pixels unchanged)
    near_cull, model.near_cull = model.near_cull, False
    got = model(to_dev(sub))
    torch.cuda.synchronize()
    model.near_cull = near_cull
    hit = [model._last["per"][p]["hit_index"][:n].long().cpu() for p, n in zip(model._last["persons"], model.last_stats["n_hit"])]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = O.SamplerCfg(N_samples=n_samples, N_samples_eval=max(128, n_samples))
    oracle = O.MultiplyOracle(sd, tables, sc["smpl_params"][0, :, 76:], cfg)
    # the oracle in chunks of one convergence group (512 rays = the reference's pixel_per_batch chunk: the sampler's convergence
    # vote spans a call, multiply.py / ray_sampler.py:137), exactly the device's groups; bounded memory at any sample size
    G = int(model.convergence_group or len(sel))
    parts = []
    t0 = time.time()
    done = 0
    for c0 in range(0, len(sel), G):
        chunk = dict(sub)
        chunk["uv"] = sub["uv"][:, c0:c0 + G]
        hg = [h[(h >= c0) & (h < c0 + G)] - c0 for h in hit]
        hg = [h if len(h) else torch.zeros(1, dtype=torch.long) for h in hg]        # multiply.py:262-263 per chunk
        parts.append(oracle.forward_eval(chunk, hg)["rgb_values"])
        done = min(len(sel), c0 + G)
        if time.time() - t0 > budget_s and done >= min(min_rays, len(sel)):
            break
    dt = time.time() - t0
    want = {"rgb_values": torch.cat(parts, 0)}
    err = (got["rgb_values"].cpu()[:done] - want["rgb_values"]).abs()
    err = err[~err.isnan()]
    threads, phys, name = host_cpu()
    hit_done = [int((h < done).sum()) for h in hit]
    cut = "" if done == len(sel) else f" (stopped by the {budget_s:.0f} s wall-time guard: {done} of the {len(sel)} rays asked for)"
    return dict(value=done / dt, unit="rays/s", cores=threads, physical_cores=phys, cpu_model=name, kind="port",
                rays=done, seconds=dt,
                sample=f"{done} rays{cut} (image rows {r0}.. of the same {H}x{W} frame, row-major from row {r0}, hit rays "
                       f"{hit_done}), fp32 torch oracle on {threads} threads, {dt:.1f} s; the frame rate "
                       f"is this rate extrapolated", parity_rgb_max_abs=float(err.max()), parity_rgb_mean_abs=float(err.mean()))


def train_iterations(model, gin, steps, warmup, dist, barrier, seed=0, rays=512):
    """ms/train-iter (BASELINE.json's second metric): forward (training mode) + Loss + backward + Adam step on `rays`
    random pixels of the frame per rank (confs/dataset: 512 pixels per iteration), current_epoch 301 (no in/off-surface
    flags, temporal loss on, pose conditioning on).  N > 1: one flat RCCL all-reduce of the gradients per step."""
    from multiply_amd.config import load_config
    from multiply_amd.loss import Loss
    from multiply_amd.parallel import GradientAllReduce
    dev = gin["uv"].device
    g = torch.Generator(device="cpu").manual_seed(seed)
    R = gin["uv"].shape[1]
    loss_fn = Loss(load_config().loss)
    # multiply_model.py configure_optimizers: Adam, lr 5e-4; torch's multi-tensor ("fused") implementation of the same update
    # on the GPU (one launch instead of ~20 element-wise ones per step)
    opt = torch.optim.Adam(model.parameters(), lr=5.0e-4, fused=dev.type == "cuda")
    # N > 1: the gradient buckets are all-reduced WHILE the backward sweep goes on (parallel.BucketedGradientSync); the
    # "allreduce" phase below is then what is left after loss.backward() returned
    from multiply_amd.parallel import BucketedGradientSync
    overlapped = bool(dist) and os.environ.get("MP_FLAT_ALLREDUCE", "0") != "1"
    model.grad_bucket_sync = BucketedGradientSync() if overlapped else None
    ar = (lambda: None) if overlapped else GradientAllReduce(model.parameters())
    model.train()
    # ray-sharded DP: the sampler's convergence vote spans the rays of ALL ranks (one MAX all-reduce per sampler iteration), so
    # that the N-rank step samples like the single-process 512-ray step (ray_sampler.py:137; tests/dist_sampler_vote.py)
    model.sampler_vote_group = True if dist else None
    model.async_setup = True       # the inputs below are resident before the loop: the setup's host sync need not wait for
    evs = []                       # the previous iteration's backward (Multiply._setup)
    acc = [0.0] * 4
    # every iteration's pixels and targets are drawn and moved to HBM BEFORE the loop (the contract: inputs resident when
    # the timed region starts; in the trainer a prefetching data loader delivers them)
    batches = []
    for _ in range(warmup + steps):
        sel = torch.randperm(R, generator=g)[:rays].to(dev)
        tin = dict(gin)
        tin["uv"] = gin["uv"][:, sel].contiguous()
        tin.update(current_epoch=301, index_outside=torch.zeros(rays, dtype=torch.bool, device=dev),
                   smpl_pose_last=gin["smpl_pose"] + 0.01)
        batches.append((tin, {"rgb": torch.rand(1, rays, 3, generator=g).to(dev)}))
    torch.cuda.synchronize()
    it_no = [0]
    host_s = [0.0]

    def one(timed):
        tin, gt = batches[it_no[0]]
        it_no[0] += 1
        h0 = time.perf_counter()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        out = model(tin)
        with contextlib.redirect_stdout(sys.stderr):     # Loss prints "Nan: bce_loss" like the reference (loss.py:125)
            lo = loss_fn(out, gt)
        ev[1].record()
        opt.zero_grad(set_to_none=True)
        lo["loss"].backward()
        ev[2].record()
        ar()
        ev[3].record()
        opt.step()
        ev[4].record()
        if timed:                  # read after the loop: no host wait inside the timed region
            evs.append(ev)
            host_s[0] += time.perf_counter() - h0      # the host's own time to enqueue the iteration (incl. the setup's one sync)
        return lo

    for _ in range(warmup):
        one(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        lo = one(True)
    barrier()
    dt = time.perf_counter() - t0
    for ev in evs:
        for i in range(4):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
    model.eval()
    model.async_setup = False
    model.sampler_vote_group = None
    model.grad_bucket_sync = None
    return dt, [a / max(steps, 1) for a in acc], float(lo["loss"]), model.last_stats, 1e3 * host_s[0] / max(steps, 1)


def train_cpu_baseline(model, gin, inp, tables, sc, n_samples, rays=512, iters=3, seed=0):
    """`iters` training iterations of the CPU oracle (fp32 torch autograd incl. the double backward through the normals and
    the eikonal term) on `rays` random pixels of the frame (the GPU figure's 512 rays).  The sampler's depths
    and the random draws are taken from a GPU call on the same pixels (the sampler runs without gradients in the
    reference)."""
    from oracle import multiply_oracle as O
    from multiply_amd.config import load_config
    from multiply_amd.loss import Loss
    dev = gin["uv"].device
    g = torch.Generator().manual_seed(seed)
    R = gin["uv"].shape[1]
    sel = torch.randperm(R, generator=g)[:rays]
    tin = dict(gin)
    tin["uv"] = gin["uv"][:, sel.to(dev)].contiguous()
    tin.update(current_epoch=301, index_outside=torch.zeros(rays, dtype=torch.bool, device=dev),
               smpl_pose_last=gin["smpl_pose"] + 0.01)
    loss_fn = Loss(load_config().loss)
    gt = {"rgb": torch.rand(1, rays, 3, generator=g)}
    model.train()
    model.zero_grad(set_to_none=True)
    with contextlib.redirect_stdout(sys.stderr):
        out = model(tin)
        lo_gpu = loss_fn(out, {"rgb": gt["rgb"].to(dev)})
    lo_gpu["loss"].backward()                       # the device's own iteration on these pixels: what the oracle is compared with
    torch.cuda.synchronize()
    graph = model._last_train
    model.eval()
    grads_gpu = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    cx = graph.cx
    hit = [cx["per"][p]["hit_index"][:graph.fg[p]["Rp"]].long().cpu() for p in cx["persons"]]
    z_given = [graph.fg[p]["zfinal"].cpu() for p in cx["persons"]]
    draws = {"person": {p: {k: v.cpu() for k, v in d.items()} for p, d in graph.draws["person"].items()},
             "bg_rand": graph.draws["bg_rand"].cpu()}
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    oracle = O.MultiplyOracle(sd, tables, sc["smpl_params"][0, :, 76:], O.SamplerCfg(N_samples=n_samples,
                                                                                    N_samples_eval=max(128, n_samples)))
    oracle.sd = sd
    for pp in oracle.persons:
        pp.sd = sd
    oin = dict(inp)
    oin["uv"] = inp["uv"][:, sel]
    tl = torch.mean(torch.square((inp["smpl_pose"] + 0.01) - inp["smpl_pose"])).reshape(())      # multiply.py:242-243
    names = [k for k, v in sd.items() if v.requires_grad]
    # The SAMPLER of this iteration against the oracle's own (round 6): the gradient comparison below hands the device's depths to the
    # oracle (`z_given`: the sampler runs without gradients in the reference, ray_sampler.py:86-87), which certifies the differentiable
    # path only -- so the depths themselves are compared here: the oracle's ErrorBoundSampler on the same rays with the same recorded
    # draws (fp32 network queries) vs the device's (sampler_sdf_mode, default near-fp32).
    z_parity = {}
    try:
        t0 = time.time()
        dirs_all, cam1 = O.get_camera_rays(oin["uv"][0], inp["pose"][0], inp["intrinsics"][0])
        zmax, zsum, zcnt, zbad, zrays = 0.0, 0.0, 0, 0, 0
        with torch.no_grad():
            for n, p in enumerate(cx["persons"]):
                so = oracle.servers[p].forward(inp["smpl_params"][0, p, 0], inp["smpl_trans"][0, p], inp["smpl_pose"][0, p],
                                               inp["smpl_shape"][0, p])
                cond = inp["smpl_pose"][0, p, 3:] / np.pi
                fn = lambda pts, p=p, so=so, cond=cond: oracle.persons[p].sdf_func(pts, cond, so["smpl_tfs"], so["smpl_verts"],
                                                                                   eval_mode=False)[0]
                d = draws["person"][p]
                z_or, _ = O.error_bound_sample(oracle.cfg, dirs_all[hit[n]], cam1[None].expand(len(hit[n]), -1), fn, oracle.beta().detach(),
                                               dict(t_rand=d["t_rand"], u_final=d["u_final"], extra_idx=d["extra_idx"].long()))
                e = (z_given[n] - z_or).abs()
                zmax, zsum, zcnt = max(zmax, float(e.max())), zsum + float(e.sum()), zcnt + e.numel()
                zbad, zrays = zbad + int((e.max(1).values > 3e-3).sum()), zrays + e.shape[0]
        # (the maximum is a single depth on a flat stretch of a CDF -- no weight there -- whenever it is large: the mean and the number
        # of rays with a depth off by more than 3e-3 are the figures to read; tests/tolerances.py Z_VALS_PRECISE)
        z_parity = {"parity_sampler_depth_max_abs": zmax, "parity_sampler_depth_mean_abs": zsum / max(zcnt, 1),
                    "parity_sampler_depth_rays_above_3e-3": zbad, "parity_sampler_depth_rays": zrays,
                    "parity_sampler_seconds": round(time.time() - t0, 1), "sampler_sdf": model.resolved_sampler_sdf_mode(0)}
    except Exception as ex:                                   # never lose the bench line to the extra check
        z_parity = {"parity_sampler_depth_note": f"not computed: {type(ex).__name__}: {ex}"}
    times, parity = [], {}
    for i in range(iters + 1):                       # the first (cold: allocator, thread pool) is compared but not timed
        t0 = time.time()
        want = oracle.forward_train(oin, hit, z_given, draws)
        want.update(fg_rgb_values_each_person_list=[], index_in_surface=None, epoch=301, temporal_loss=tl,
                    smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1))
        with contextlib.redirect_stdout(sys.stderr):
            lo = loss_fn(want, gt)
        # The reference replaces a NaN bce term by zero (loss.py:124-128): log(1 - acc + 1e-6) is NaN as soon as ONE opacity exceeds
        # 1 + 1e-6 -- a rounding-level event (the device's and the oracle's acc_map agree to ~7e-6, and a ray through several opaque
        # bodies sums to 1 within that).  When the guard fires on one side only, the two iterations differ by the whole bce term and
        # no gradient is comparable: the oracle's loss is then taken without the term as well (and the event is recorded).
        guard_gpu, guard_or = float(lo_gpu["bce_loss"]) == 0.0, float(lo["bce_loss"]) == 0.0
        bce_note = None
        loss_unadjusted = float(lo["loss"])
        if guard_gpu and not guard_or:
            lo = dict(lo)
            lo["loss"] = lo["loss"] - loss_fn.bce_weight * lo["bce_loss"]
            bce_note = (f"the reference's NaN guard on bce_loss (loss.py:124-128) fired on the device (acc_map max "
                        f"{float(out['acc_map'].max()):.8f} > 1 + 1e-6) and not on the oracle (max {float(want['acc_map'].max()):.8f}): "
                        f"the oracle's loss is compared without the bce term too")
        elif guard_or and not guard_gpu:
            bce_note = ("the reference's NaN guard on bce_loss fired on the oracle and not on the device: the parity figures of this "
                        "iteration include the whole bce term")
        gw = torch.autograd.grad(lo["loss"], [sd[k] for k in names], allow_unused=True)
        times.append(time.time() - t0)
        if i == 0:
            # parity of the SAME iteration at the benchmarked workload: loss, and every parameter gradient (relative L2 per
            # tensor; tensors whose gradient is below 1e-7 everywhere are compared absolutely and left out of the maximum)
            worst, worst_name, n_cmp, n_tiny, n_nograd = 0.0, "", 0, 0, 0
            for k, gwk in zip(names, gw):
                if k not in grads_gpu or gwk is None:
                    n_nograd += k not in grads_gpu
                    continue
                a, b = grads_gpu[k].reshape(-1), gwk.double().reshape(-1)
                if float(b.abs().max()) < 1e-7 and float((a - b).abs().max()) < 1e-7:
                    n_tiny += 1
                    continue
                rel = float((a - b).norm() / (b.norm() + 1e-12))
                n_cmp += 1
                if rel > worst:
                    worst, worst_name = rel, k
            fwd = {k: float((out[k].detach().cpu() - want[k].detach()).abs().nan_to_num().max())
                   for k in ("rgb_values", "acc_map", "acc_person_list", "normal_values")}
            parity = {"parity_loss_abs": abs(float(lo_gpu["loss"]) - float(lo["loss"])), "loss_gpu": float(lo_gpu["loss"]),
                      "loss_oracle": float(lo["loss"]), "parity_grad_rel_worst": worst, "parity_grad_worst_tensor": worst_name,
                      "parity_grad_tensors": n_cmp, "parity_grad_tensors_below_1e-7": n_tiny,
                      "parity_state_entries_without_gradient": n_nograd, "parity_forward_max_abs": fwd, "parity_note": bce_note,
                      # structured form of the note (advisor, round 5): did the reference's NaN guard on the bce term fire on one side only?
                      # `parity_loss_abs_unadjusted` is the difference of the two losses as they were, bce term included on the oracle's side
                      "bce_guard": {"device": guard_gpu, "oracle": guard_or, "mismatch": guard_gpu != guard_or},
                      "parity_loss_abs_unadjusted": abs(float(lo_gpu["loss"]) - loss_unadjusted), **z_parity}
        del gw
    dt = float(np.mean(times[1:])) if len(times) > 1 else float(times[0])     # iters = 0 (tests): the one compared iteration, cold
    threads, phys, name = host_cpu()
    return {"value": 1e3 * dt, "unit": "ms/train-iter", "cores": threads, "physical_cores": phys, "cpu_model": name,
            "kind": "port", "rays": rays, "iters": iters, "discarded_warmup_iters": 1, **parity,
            "sample": f"mean of {iters} iterations (after one discarded cold iteration, {round(times[0], 1)} s) on {rays} rays "
                      f"(hit rays {[int(len(h)) for h in hit]}): forward from the sampler's depths + loss + autograd on the fp32 "
                      f"torch oracle, {threads} threads; per iteration {[round(t, 1) for t in times[1:]]} s"}


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the sources of the eval-path kernels: what a committed PMC measurement is tied to"""
    import hashlib
    h = hashlib.sha256()
    for f in ("mlp.hip", "mlp_core.hpp", "geom.hip", "sampler.hip", "composite.hip", "common.hpp"):
        with open(os.path.join(REPO, "multiply_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: one process per GPU through torch.distributed.run (the driver starts the
    same module itself for its multi-GPU runs)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def timed_frames(model, gin, steps, barrier, after=None, frame_fn=None):
    """EXACTLY `steps` forward passes between two barriers; after(out) runs inside the timed region (the all_gather of the
    strong-scaling mode).  frame_fn(gin) replaces model(gin) (person / hybrid modes: a frame is several calls; it returns the
    frame's (n_shaded, n_sdf_evals) lists itself).  Returns elapsed seconds and the per-step statistics."""
    shaded, sdf_evals = [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        if frame_fn is not None:
            out, sh, se = frame_fn(gin)
            shaded.append(sh)
            sdf_evals.append(se)
        else:
            out = model(gin)
            shaded.append(model.last_stats["n_shaded"])
            sdf_evals.append(model.last_stats["n_sdf_evals"])
        if after is not None:
            after(out)
    barrier()
    return time.perf_counter() - t0, shaded, sdf_evals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=128, help="importance samples per ray (N_samples)")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--tile", type=int, default=8, help="emit the frame's rays in tile x tile pixel blocks (0 = row-major)")
    ap.add_argument("--train-steps", type=int, default=10, help="timed training iterations for ms/train-iter (0 = skip)")
    ap.add_argument("--train-warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=8192, help="rays of the CPU oracle's render sample (~1 min per 1000 on the box's host)")
    ap.add_argument("--cpu-train-iters", type=int, default=2, help="timed oracle training iterations (one more, cold, is run first and discarded)")
    ap.add_argument("--cpu-train-rays", type=int, default=512, help="rays of the CPU oracle's training iterations and of the device iteration "
                    "they are compared with (default: the reference's 512 pixels per iteration; the test suite uses fewer)")
    ap.add_argument("--cpu-budget-s", type=float, default=330.0, help="wall-time guard of the CPU oracle's render sample: it stops at the "
                    "first 512-ray group boundary past this many seconds (once 4096 rays are done)")
    ap.add_argument("--persons", type=int, default=2, help="persons of the synthetic scene (BASELINE.json configs[3]: --persons 4 --samples 256)")
    ap.add_argument("--mode", choices=("ray", "person", "hybrid"), default="ray",
                    help="N > 1: how ONE frame is split -- ray: convergence groups dealt to the ranks (configs[2], default); person: persons "
                         "sharded, one all_to_all (configs[3] on <= P ranks); hybrid: person teams x ray shards (configs[3] on 8 ranks)")
    ap.add_argument("--person-slots", type=int, default=0, help="--mode hybrid: ranks per team (default: min(persons, world))")
    ap.add_argument("--chunk-rays", type=int, default=16384, help="--mode person / hybrid: rays per exchange (whole convergence groups)")
    ap.add_argument("--sampler-sdf", choices=("auto", "f16", "f16x2", "bf16x3"), default=os.environ.get("MP_SAMPLER_SDF", "auto"),
                    help="arithmetic of the sampler's network queries: auto (default) = bf16x3 (near-fp32: mp_tf_sdf_val; depths within "
                         "1e-3 of the fp32 reference instead of 2e-2) | f16x2 (split activations) | f16 (fused half-precision kernel, "
                         "the round 1-5 default: ~5 ms per frame less)")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the frame-per-rank weak-scaling leg")
    ap.add_argument("--breakdown", action="store_true", help="print per-phase GPU times to stderr")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args.gpus)            # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    dist = world > 1
    # one rank per GPU.  (MP_BENCH_BACKEND=gloo lets the multi-rank code path be smoke-tested on a box with fewer GPUs
    # than ranks: ranks then share devices and collectives go through the host.)
    backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    td = None
    ranks_seen = None
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend)
        assert td.get_world_size() == args.gpus, "process group size differs from --gpus"
        # what the driver's SCALE record can be checked against: an all_reduce of ones over the data-path backend (nccl = RCCL)
        one = torch.ones(1, device="cuda")
        td.all_reduce(one)
        torch.cuda.synchronize()
        ranks_seen = int(round(float(one.item())))
        assert ranks_seen == td.get_world_size(), f"all_reduce of ones saw {ranks_seen} ranks, the group has {td.get_world_size()}"
    if args.mode != "ray" and not dist:
        raise SystemExit("bench.py: --mode person / hybrid split ONE frame over several ranks: use --gpus N > 1 (the single-GPU line of "
                         "configs[3] is `--persons 4 --samples 256`)")

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist:
            return x
        tt = torch.tensor([x], device="cuda", dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        return float(tt.item())

    GROUP = 512                              # the reference renders frames in chunks of pixel_per_batch = 512 rays
    # strong scaling: every rank holds the SAME frame (seed 0) and renders its round-robin share of the convergence groups
    model, inp, tables, sc = build_model(args.samples, seed=0, H=args.res, W=args.res, P=args.persons, tile=args.tile)
    model.convergence_group = GROUP
    model.sampler_sdf_mode = args.sampler_sdf
    R = inp["uv"].shape[1]
    exchange_evs, frame_fn = None, None
    if dist and args.mode != "ray":
        slots = args.person_slots or min(args.persons, world)
        if args.mode == "hybrid" and (slots < 1 or world % slots):
            raise SystemExit(f"bench.py: --mode hybrid needs world ({world}) = ray shards x person slots ({slots})")
        gin = to_dev(inp)                     # every rank holds the whole frame's rays; persons (and ray shards) split the work
        frame_fn, exchange_evs, gather_evs = person_mode_renderer(model, gin, args.mode, world, rank, GROUP, args.chunk_rays, slots,
                                                                  ("rgb_values", "normal_values", "fg_rgb_values"))
        assemble = None                       # the image all_gather is part of the frame function
    elif dist:
        from multiply_amd.parallel import gather_rays_interleaved, shard_input_interleaved
        # groups per band of the image: with tile x tile ray order a group of 512 rays is a (512 / tile)-pixel wide block
        GPR = max(1, (args.res * args.tile) // GROUP) if args.tile else max(1, args.res // GROUP)
        share, my_ids = shard_input_interleaved(inp, rank, world, GROUP, GPR)
        gin = to_dev(share)
        image = {}

        KEEP = ("rgb_values", "normal_values", "fg_rgb_values")

        gather_evs = []                      # (start, end) events around every image all_gather of the timed region

        def assemble(out):                   # ONE all_gather of the outputs the caller keeps (multiply_model.py:1045-1069)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            full = gather_rays_interleaved(torch.cat([out[k] for k in KEEP], dim=1), R, world, GROUP, GPR)
            for k, part in zip(KEEP, full.split([out[k].shape[1] for k in KEEP], dim=1)):
                image[k] = part
            e1.record()
            gather_evs.append((e0, e1))
    else:
        gin, assemble = to_dev(inp), None

    model.profile = True                     # the warm-up also creates the pool of timing events the phases use
    # the frame's inputs are resident before the loop: the setup (SMPL, culls) and its one host sync run on a side stream, so that
    # the host keeps enqueuing ahead of the GPU from frame to frame (Multiply._setup, async_setup)
    model.async_setup = os.environ.get("MP_BENCH_ASYNC_EVAL", "1") == "1"
    for _ in range(args.warmup):
        o = frame_fn(gin)[0] if frame_fn is not None else model(gin)
        if assemble is not None:
            assemble(o)
    torch.cuda.synchronize()
    model.phase_events = {}
    if dist:
        gather_evs.clear()
    if exchange_evs is not None:
        exchange_evs.clear()
    elapsed, shaded, sdf_evals = timed_frames(model, gin, args.steps, barrier, assemble, frame_fn)
    model.profile = False
    model.async_setup = False
    per_rank = None
    if dist:
        # what a scaling run is diagnosed with: every rank's own wall time per frame, its share of the rays and the time it
        # spent inside the image all_gather (which includes waiting for the slowest rank)
        mine = torch.tensor([1e3 * elapsed / args.steps, float(gin["uv"].shape[1]),
                             sum(a.elapsed_time(b) for a, b in gather_evs) / max(len(gather_evs), 1)], device="cuda", dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        td.all_gather(allr, mine)
        per_rank = {"ms_per_step": [float(t[0]) for t in allr], "rays": [int(t[1]) for t in allr],
                    "all_gather_ms_per_step": [float(t[2]) for t in allr]}
        if exchange_evs is not None:         # person / hybrid: the all_to_all exchanges' own time (incl. waiting for the team's slowest rank)
            mine = torch.tensor([sum(a.elapsed_time(b) for a, b in exchange_evs) / args.steps, len(exchange_evs) / args.steps],
                                device="cuda", dtype=torch.float64)
            allx = [torch.empty_like(mine) for _ in range(world)]
            td.all_gather(allx, mine)
            per_rank["exchange_ms_per_step"] = [float(t[0]) for t in allx]
            per_rank["exchanges_per_step"] = [float(t[1]) for t in allx]
    elapsed = max_over_ranks(elapsed)
    render_stats = model.last_stats
    phases = model.phase_times_ms()

    weak = None
    if dist and not args.no_weak and args.mode == "ray":            # frame-per-rank: rank r renders frame r of the synthetic sequence, no collective
        wmodel, winp, _, _ = build_model(args.samples, seed=rank, H=args.res, W=args.res, P=args.persons, tile=args.tile)
        wmodel.convergence_group = GROUP
        wgin = to_dev(winp)
        for _ in range(args.warmup):
            wmodel(wgin)
        wel, _, _ = timed_frames(wmodel, wgin, args.steps, barrier)
        wel = max_over_ranks(wel)
        weak = {"value": R * args.steps * world / wel, "unit": "rays/s", "ms_per_step": 1e3 * wel / args.steps, "scaling": "weak",
                "parallelism": f"frame-sharded dp{world}: every rank renders its own frame, no data-path collective"}
        del wmodel, wgin

    train = None
    if args.train_steps > 0:
        from multiply_amd import train as _T
        if _T.TRAIN_PRECISION == "f32":
            train_dtype, train_peak = "f32", 157.3
            train_note = "exact-fp32 matrix instruction (v_mfma_f32_16x16x4_f32), MP_TRAIN_PRECISION=f32"
            train_peak_note = "the fp32-matrix peak, 157.3 TFLOP/s"
        else:
            train_dtype, train_peak = "bf16x3", PEAK_BF16_TFLOPS / 3.0
            train_note = ("fp32 tensors and fp32 accumulation; every GEMM operand is split into two bfloat16 halves and a product is "
                          "three 16-bit MFMAs (~2^-16 relative per product, fp32 range) -- not a reduced-precision training step: "
                          "parity_* below compare it with the fp32 oracle")
            train_peak_note = "the dense 16-bit MFMA peak / 3 (three MFMAs per product) = 833 TFLOP/s"
        rays_rank = 512 // world             # strong scaling: the reference's 512 pixels per iteration split over the ranks
        full = to_dev(inp)
        tdt, tph, tloss, tstats, thost = train_iterations(model, full, args.train_steps, args.train_warmup, dist, barrier, seed=rank,
                                                   rays=rays_rank)
        tdt = max_over_ranks(tdt)
        # exact op count of the iteration's differentiable MLP work: fg SDF net 6 GEMM passes over
        # (hit-ray samples + eikonal points) rows, colour net 3, background nets 3 over 32 samples per ray (fp32 MFMA)
        S = args.samples + 33
        hit_rows = sum(int(h) for h in tstats["n_hit"]) * S
        eik_rows = 512 * len(tstats["n_hit"])
        tflop = (6 * (hit_rows + eik_rows) * 2 * M_IMP + 3 * hit_rows * 2 * M_REN + 3 * 32 * rays_rank * 2 * (M_BGIMP + M_BGREN))
        train = {"metric": "ms/train-iter (forward + loss + backward + gradient all-reduce + Adam step)",
                 "ms_per_iter": 1e3 * tdt / args.train_steps, "steps": args.train_steps, "warmup": args.train_warmup,
                 "rays_per_iter_per_gpu": rays_rank, "rays_per_iter": rays_rank * world, "scaling": "strong",
                 "dtype": train_dtype, "dtype_note": train_note, "obb_mode": model.obb_mode,
                 "gpu_ms": {"forward+loss": tph[0], "backward": tph[1], "allreduce": tph[2], "adam": tph[3]},
                 "host_ms_per_iter": thost,
                 "hit_rays": tstats["n_hit"], "last_loss": tloss,
                 # the foreground SDF net's per-point stash (csrc/tfuse.hip): 46 fp32 tensors of 256 values + the encoded inputs.  Round 6
                 # built and measured 16-bit storage of V(l) / U(l) (profiles/r06_stash16_ab.txt): slower and not accurate enough -- reverted
                 "stash_bytes_per_point": 46 * 1024 + 3 * 39 * 4,
                 "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": train_peak, "flop_per_iter_this_rank": tflop,
                              "achieved": tflop / (tdt / args.train_steps) / 1e12,
                              "frac": tflop / (tdt / args.train_steps) / 1e12 / train_peak,
                              "note": "algorithmic GEMM FLOP of the differentiable path (the sampler's queries are not counted) "
                                      "over the whole iteration's wall time; peak = " + train_peak_note + ".  The iteration is not "
                                      "matrix-bound: its three dominant kernels (k_tf_sdf_fwd / _bwd, k_gemm_tn_b3w: 5.1 of 9.4 ms) move "
                                      "the SDF net's per-point stash at 2.8-4.3 TB/s (profiles/r05_train_pmc_traffic.txt, r06_train_kernels.txt)"}}

    n_shaded = float(sum(int(w.sum()) for s_ in shaded for w in s_)) / args.steps          # per frame (this rank's share)
    n_sdf = float(sum(int(w[:-1].sum()) for s_ in sdf_evals for w in s_)) / args.steps
    R_rank = gin["uv"].shape[1] if frame_fn is None else (R + world - 1) // world     # person / hybrid: background is ray-partitioned
    flops = {"mlp_shade": n_shaded * 4 * M_IMP, "sampler_mlp_sdf": n_sdf * 2 * M_IMP, "mlp_color": n_shaded * 2 * M_REN,
             "background": R_rank * 32 * 2 * (M_BGIMP + M_BGREN)}
    dom = max(flops, key=lambda k: phases.get(k, (0, 0.0))[1])
    n_launch, ms = phases[dom]
    per_launch_s = ms / 1e3 / max(n_launch, 1)
    launches_per_frame = n_launch / args.steps
    achieved = flops[dom] / launches_per_frame / per_launch_s / 1e12
    if args.breakdown and rank == 0:
        tot = sum(v[1] for v in phases.values())
        for k, (n, m) in sorted(phases.items(), key=lambda kv: -kv[1][1]):
            print(f"  {k:18s} {n:5d} launches {m / args.steps:9.2f} ms/frame {100 * m / tot:5.1f}%", file=sys.stderr)
        print(f"  shaded points/frame {n_shaded:.0f}  sdf evals/frame {n_sdf:.0f}  hit rays {render_stats['n_hit']}",
              file=sys.stderr)

    # HBM traffic of the dominant kernel: rocprofv3 PMC passes of this same command (FETCH_SIZE and WRITE_SIZE in
    # separate runs, gfx950 half-count correction on reads) are kept under profiles/; the counters cannot be read from
    # inside the process, so the committed measurement is attached when it was taken on this workload.
    traffic = None
    kernels = {"mlp_shade": ["k_mlp_fwdsave", "k_mlp_grad"] if model.shade_mode == "reverse" else ["k_mlp_shade"],
               "mlp_color": ["k_mlp_color"], "background": ["k_background"], "sampler_mlp_sdf": ["k_mlp_sdf"]}[dom]
    pmc_file = next((os.path.join(REPO, "profiles", f) for f in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json")
                     if os.path.exists(os.path.join(REPO, "profiles", f))), None)
    traffic_note = None
    if pmc_file and args.res == 512 and args.samples == 128 and args.persons == 2 and world == 1:
        with open(pmc_file) as f:
            pmc = json.load(f)                      # per-dispatch averages of ONE frame (bench.py --steps 1 --warmup 0)
        # a committed measurement is attached only when it is a measurement of THIS tree on THIS workload (tools/write_profiles.py
        # records both): the hash of the kernels' sources and the algorithmic work per launch of the profiled run
        meta = pmc.pop("_meta", None)
        want_hash = kernel_source_hash()
        if meta is None:
            traffic_note, pmc = f"{os.path.basename(pmc_file)} carries no _meta record (tree / workload of the profiled run unknown): not attached", {}
        elif meta.get("kernel_source_sha16") != want_hash:
            traffic_note, pmc = (f"{os.path.basename(pmc_file)} was measured on kernel sources {meta.get('kernel_source_sha16')}, this tree "
                                 f"is {want_hash}: not attached"), {}
        elif abs(meta.get("algorithmic_flop_per_launch", 0.0) / (flops[dom] / launches_per_frame) - 1.0) > 0.01 or \
                meta.get("dominant") != dom:
            traffic_note, pmc = (f"{os.path.basename(pmc_file)}: the profiled run's work per launch ({meta.get('dominant')}, "
                                 f"{meta.get('algorithmic_flop_per_launch', 0.0):.4g} FLOP) differs from this run's: not attached"), {}
        rd = wr = 0.0
        for name, e in pmc.items():
            if any(k in name for k in kernels):
                rd += e["hbm_read_bytes_corrected"] * e["dispatches"]
                wr += e["hbm_write_bytes_uncalibrated"] * e["dispatches"]
        if rd + wr > 0:
            traffic = {"bytes_per_launch": (rd + wr) / launches_per_frame, "read": rd / launches_per_frame,
                       "write": wr / launches_per_frame,
                       "source": os.path.relpath(pmc_file, REPO).replace(".json", ".txt") +
                                 " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; Infinity-Cache hits are "
                                 "counted: an upper bound on HBM bytes)"}

    if rank == 0:
        out = {
            "metric": "rays/sec rendering full 512x512 frames (eval forward, all persons, with background)",
            "value": R * args.steps / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f16",
            "dtype_note": "MLP operands IEEE half (f16: the bf16 operand width and MFMA rate BASELINE.json's configs[1] names, 3 more "
                          "mantissa bits), fp32 accumulation; everything else fp32.  The reference is fp32; stated tolerances vs the "
                          "fp32 oracle: tests/tolerances.py, cpu_baseline.parity_* below",
            "collective_backend": (backend if dist else None), "rccl_ranks_seen": ranks_seen, "data": "synthetic",
            "config": {"workload": f"{args.persons}-person synthetic SMPL scene, {args.res}x{args.res} rays/frame, N_samples="
                                   f"{args.samples} (+32 extra +2 bounds = {args.samples + 33} composited samples/ray/"
                                   f"person), N_samples_eval={max(128, args.samples)}, 32 background samples, "
                                   f"convergence groups of 512 rays (reference pixel_per_batch), geometric-init weights",
                       "rays_per_step": R, "frames": args.steps,
                       "persons": args.persons, "sampler_sdf": model.resolved_sampler_sdf_mode(0),
                       "parallelism": ("single GPU" if not dist else
                                       f"ray-sharded dp{world}: one frame per step, convergence groups dealt on a diagonal lattice, "
                                       f"all_gather of the image on every rank" if args.mode == "ray" else
                                       f"person-sharded x{world}: rank g evaluates persons p % {world} == g for all rays, one all_to_all per "
                                       f"person slot and {args.chunk_rays}-ray chunk, compositing + background ray-partitioned, one all_gather "
                                       f"of the image" if args.mode == "person" else
                                       f"hybrid: {world // (args.person_slots or min(args.persons, world))} ray shards x "
                                       f"{args.person_slots or min(args.persons, world)} person slots (teams exchange inside, one all_gather "
                                       f"of the image over the world)")},
            "roofline": {"bound": "mfma", "kernel": dom + " = " + " + ".join(kernels), "achieved": achieved, "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                         "avg_launch_ms": 1e3 * per_launch_s, "launches_per_step": launches_per_frame,
                         "algorithmic_flop_per_launch": flops[dom] / launches_per_frame,
                         "note": "rank 0's share of the frame" if dist else None},
            "phases_ms_per_step": {k: v[1] / args.steps for k, v in phases.items()},
            "lib_source_sha16": __import__("multiply_amd.hip", fromlist=["x"]).lib_source_sha16(),     # content hash of csrc/* + headers + flags the loaded library was built from
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if weak is not None:
            out["weak"] = weak
        if train is not None:
            out["train_iter"] = train
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, inp, tables, sc, args.samples, n_rays=args.cpu_rays, budget_s=args.cpu_budget_s)
            if train is not None:
                train["cpu_baseline"] = train_cpu_baseline(model, to_dev(inp), inp, tables, sc, args.samples,
                                                           rays=args.cpu_train_rays, iters=args.cpu_train_iters)
        print(json.dumps(out))
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
