"""Training-mode Multiply.forward + Loss + hand-written backward on the GPU vs the CPU oracle under torch autograd.

The sampler runs without gradients in the reference (ray_sampler.py:86-87): its depths are taken from the GPU run and
handed to the oracle, together with the same random draws, so that everything downstream -- outputs, loss and the
gradient of EVERY parameter -- is compared on identical inputs.  The differentiable path is fp32 on both sides
(exact-f32 MFMA GEMMs); tolerances are stated at the assertions."""
import numpy as np
import pytest
import torch

from oracle import multiply_oracle as O
from tests import tolerances as TOL
from tests.test_render_gpu import build, report

pytestmark = pytest.mark.gpu


def _cpu(d):
    if torch.is_tensor(d):
        return d.detach().cpu()
    if isinstance(d, dict):
        return {k: _cpu(v) for k, v in d.items()}
    return d


def _train_setup(H=11, W=11, epoch=301):      # odd size: no ray through the sphere centre (NaN, see test_render_gpu.report)
    from multiply_amd.loss import Loss
    from multiply_amd.config import load_config
    from multiply_amd import train
    model, oracle, inp = build(H=H, W=W)
    model.train()
    R = inp["uv"].shape[1]
    gin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    gin.update(current_epoch=epoch, index_outside=torch.zeros(R, dtype=torch.bool),
               smpl_pose_last=gin["smpl_pose"] + 0.01)
    g = torch.Generator().manual_seed(5)
    gt = {"rgb": torch.rand(1, R, 3, generator=g)}
    gin["sam_mask"] = (torch.randn(1, R, 2, generator=g) * 4).cuda()
    loss_fn = Loss(load_config().loss)
    return model, oracle, inp, gin, gt, loss_fn, train


def test_training_sampler_matches_oracle_with_same_draws():
    """ErrorBoundSampler in training mode (stratified t_rand, random u, randperm extras): same draws -> same depths
    up to the half-precision SDF of the sampler's network queries."""
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(R)]
    cx = model._setup({**gin, "hit_index": hit}, -1, False)
    draws = train.make_draws(model, cx, None)
    for n, p in enumerate(cx["persons"]):
        zfinal, iters, _ = model._sample_person(cx, n, p, draws["person"][p])
        torch.cuda.synchronize()
        so = oracle.servers[p].forward(inp["smpl_params"][0, p, 0], inp["smpl_trans"][0, p], inp["smpl_pose"][0, p],
                                       inp["smpl_shape"][0, p])
        cond = inp["smpl_pose"][0, p, 3:] / np.pi
        dirs, cam1 = O.get_camera_rays(inp["uv"][0], inp["pose"][0], inp["intrinsics"][0])
        cam = cam1[None].expand(R, -1)
        fn = lambda pts: oracle.persons[p].sdf_func(pts, cond, so["smpl_tfs"], so["smpl_verts"], eval_mode=False)[0]
        d = _cpu(draws["person"][p])
        z, it_o = O.error_bound_sample(oracle.cfg, dirs, cam, fn, oracle.beta().detach(),
                                       dict(t_rand=d["t_rand"], u_final=d["u_final"], extra_idx=d["extra_idx"].long()))
        print("[info] iterations oracle", it_o, "gpu", iters.tolist())
        assert model.resolved_sampler_sdf_mode(p) == "bf16x3"          # the default: near-fp32 queries (mp_tf_sdf_val)
        assert TOL.within(report(f"train z_vals person {p}", zfinal, z), TOL.TRAIN_Z_VALS)
        # the opt-out, the half-precision sampler kernel, same draws: an order of magnitude farther
        model.sampler_sdf_mode = "f16"
        zp, it_p, _ = model._sample_person(cx, n, p, draws["person"][p])
        model.sampler_sdf_mode = "auto"
        torch.cuda.synchronize()
        assert TOL.within(report(f"train z_vals person {p}, sampler_sdf_mode f16", zp, z), TOL.TRAIN_Z_VALS_F16)


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_training_forward_loss_and_all_parameter_gradients(precision, monkeypatch):
    """both arithmetic modes of the training GEMMs (multiply_amd.train.TRAIN_PRECISION): split-bfloat16 products (default) and the
    exact-fp32 matrix instruction (the cross-check), each against the oracle at its stated tolerance"""
    from multiply_amd import train as T
    monkeypatch.setattr(T, "TRAIN_PRECISION", precision)
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(R)]
    out = model({**gin, "hit_index": hit})
    assert len(out) == 20 and isinstance(out["t_list"], list) and out["index_in_surface"] is None
    assert out["grad_theta"].shape == (1, 1024, 3) and out["points"].shape == (R, 97, 3)
    lo = loss_fn(out, gt)
    model.zero_grad()
    lo["loss"].backward()
    torch.cuda.synchronize()
    graph = model._last_train

    # oracle on the same depths and draws
    for v in oracle.sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    z_given = [graph.fg[p]["zfinal"].cpu() for p in range(2)]
    want = oracle.forward_train(inp, hit, z_given, _cpu(graph.draws))
    # multiply.py:242-243, from the INPUTS (not from the device's output): mean((smpl_pose_last - smpl_pose)^2)
    tl = torch.mean(torch.square((inp["smpl_pose"] + 0.01) - inp["smpl_pose"]))
    want.update(fg_rgb_values_each_person_list=[], index_in_surface=None, epoch=301,
                temporal_loss=tl.reshape(()), smpl_surface_loss=torch.zeros(1),
                zero_pose_loss=torch.zeros(1), sam_mask=gin["sam_mask"].squeeze().cpu())
    lw = loss_fn(want, gt)
    names = [k for k, v in oracle.sd.items() if v.requires_grad]
    gw = torch.autograd.grad(lw["loss"], [oracle.sd[k] for k in names], allow_unused=True)

    # forward: fp32 vs fp32, different summation orders only
    for k, tol in TOL.train_fwd().items():
        mx, mean = report("train " + k, out[k], want[k].detach())
        assert mx < tol, k
    for k in ("loss", "rgb_loss", "eikonal_loss", "bce_loss", "sam_mask_loss", "temporal_loss"):
        a, b = float(lo[k]), float(lw[k])
        print(f"[parity] loss term {k}: gpu {a:.6f} oracle {b:.6f}")
        assert abs(a - b) < 1e-4 * max(1.0, abs(b)), k

    # backward: every parameter.  Relative error in the L2 sense per tensor; the colour nets' ReLU masks and the
    # |sdf| of the background density can flip for single samples between summation orders, hence 2e-2 there.
    worst = _compare_parameter_gradients(model, oracle, names, gw)
    print(f"[parity] worst relative parameter-gradient error {worst:.3e} over {len(names)} tensors")


def _compare_parameter_gradients(model, oracle, names, gw):
    got = dict(model.named_parameters())
    worst = 0.0
    for k, g in zip(names, gw):
        if k not in got:          # registered buffers (the SMPL tables) are part of the state dict, not parameters
            continue
        a = got[k].grad
        if g is None:
            assert a is None or float(a.abs().max()) == 0.0, k
            continue
        assert a is not None, f"no gradient for {k}"
        a = a.detach().cpu().double().reshape(-1)
        b = g.double().reshape(-1)
        rel = float((a - b).norm() / (b.norm() + 1e-12))
        worst = max(worst, rel)
        tol = TOL.TRAIN_GRAD_REL_RENDERING if "rendering" in k else TOL.TRAIN_GRAD_REL
        assert rel < tol or float((a - b).abs().max()) < 1e-7, f"{k}: rel {rel:.3e} |g| {float(b.norm()):.3e}"
    return worst


def test_training_forward_and_gradients_from_the_oracles_own_sampler_depths():
    """The other tests hand the DEVICE's depths to the oracle.  Here the depths come from the ORACLE's sampler (fp32 SDF
    queries, the same recorded draws) and are handed to the device (`z_given`): everything downstream of the sampler --
    warp, both networks, normals, eikonal term, compositing, background, loss, every parameter gradient -- is compared from
    depths the device never produced."""
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(R)]
    cx = model._setup({**gin, "hit_index": hit}, -1, False)
    draws = train.make_draws(model, cx, None)
    dirs, cam1 = O.get_camera_rays(inp["uv"][0], inp["pose"][0], inp["intrinsics"][0])
    cam = cam1[None].expand(R, -1)
    z_oracle = []
    for p in cx["persons"]:
        so = oracle.servers[p].forward(inp["smpl_params"][0, p, 0], inp["smpl_trans"][0, p], inp["smpl_pose"][0, p],
                                       inp["smpl_shape"][0, p])
        cond = inp["smpl_pose"][0, p, 3:] / np.pi
        fn = lambda pts: oracle.persons[p].sdf_func(pts, cond, so["smpl_tfs"], so["smpl_verts"], eval_mode=False)[0]
        d = _cpu(draws["person"][p])
        with torch.no_grad():
            z, _ = O.error_bound_sample(oracle.cfg, dirs, cam, fn, oracle.beta().detach(),
                                        dict(t_rand=d["t_rand"], u_final=d["u_final"], extra_idx=d["extra_idx"].long()))
        z_oracle.append(z.float())
        draws["person"][p]["z_given"] = z.float().cuda()
    out = train.forward_train(model, {**gin, "hit_index": hit}, draws=draws)
    lo = loss_fn(out, gt)
    model.zero_grad()
    lo["loss"].backward()
    torch.cuda.synchronize()
    for p in cx["persons"]:
        assert torch.equal(model._last_train.fg[p]["zfinal"].cpu(), z_oracle[p])          # the device did not sample
    for v in oracle.sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    want = oracle.forward_train(inp, hit, z_oracle, _cpu(draws))
    tl = torch.mean(torch.square((inp["smpl_pose"] + 0.01) - inp["smpl_pose"]))
    want.update(fg_rgb_values_each_person_list=[], index_in_surface=None, epoch=301, temporal_loss=tl.reshape(()),
                smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1), sam_mask=gin["sam_mask"].squeeze().cpu())
    lw = loss_fn(want, gt)
    names = [k for k, v in oracle.sd.items() if v.requires_grad]
    gw = torch.autograd.grad(lw["loss"], [oracle.sd[k] for k in names], allow_unused=True)
    for k, tol in TOL.train_fwd().items():
        mx, mean = report("train (oracle depths) " + k, out[k], want[k].detach())
        assert mx < tol, k
    assert abs(float(lo["loss"]) - float(lw["loss"])) < 1e-4 * max(1.0, abs(float(lw["loss"])))
    worst = _compare_parameter_gradients(model, oracle, names, gw)
    print(f"[parity] from the oracle's depths: loss gpu {float(lo['loss']):.6f} oracle {float(lw['loss']):.6f}, worst relative "
          f"parameter-gradient error {worst:.3e}")


def test_training_parity_at_the_benchmarked_workload():
    """BASELINE.json configs[1] as bench.py's `train_iter` runs it: 512 random pixels of the 512x512 two-person frame,
    N_samples = 128 (161 composited samples per ray and person), epoch 301.  One iteration of the device (forward + loss +
    hand-written backward) against one iteration of the oracle under torch autograd on the same pixels, depths and draws --
    the comparison bench.py records in its JSON line (`train_iter.cpu_baseline.parity_*`)."""
    import bench
    model, inp, tables, sc = bench.build_model(128)
    model.convergence_group = 512
    gin = bench.to_dev(inp)
    res = bench.train_cpu_baseline(model, gin, inp, tables, sc, 128, rays=512, iters=0)       # ONE oracle iteration: the compared one
    print("[parity] bench workload:", {k: v for k, v in res.items() if k.startswith(("parity", "loss_"))})
    # round 6: the sampler of that iteration against the oracle's own sampler on the same draws (tests/tolerances.py Z_VALS_PRECISE)
    assert res["parity_sampler_depth_mean_abs"] < TOL.Z_VALS_PRECISE["mean"]
    assert res["parity_sampler_depth_rays_above_3e-3"] <= max(TOL.TRAIN_DEPTH_RAYS["floor"], TOL.TRAIN_DEPTH_RAYS["frac"] * res["parity_sampler_depth_rays"])
    assert res["parity_grad_tensors"] >= 60
    assert res["parity_loss_abs"] < 1e-4 * max(1.0, abs(res["loss_oracle"]))
    assert res["parity_grad_rel_worst"] < TOL.TRAIN_GRAD_REL_RENDERING, res["parity_grad_worst_tensor"]
    for k, v in res["parity_forward_max_abs"].items():
        # measured at this workload: rgb 1.3e-6, acc 2.9e-6, normals 1.2e-4 (82 k shaded samples per person: the worst one
        # sits where |grad sdf| is small and the normalisation amplifies the fp32 summation-order difference)
        assert v < (6e-4 if k == "normal_values" else 3 * TOL.train_fwd().get(k, 8e-6)), (k, v)


def test_training_step_with_the_near_fp32_sampler_mode():
    """sampler_sdf_mode = 'bf16x3' inside a training forward (the sampler's queries on the iteration's shared, already resolved
    weights): the step runs, and its depths differ from the f16 sampler's by no more than the f16 tolerance"""
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(R)]
    zs = {}
    for mode in ("f16", "bf16x3"):
        model.sampler_sdf_mode = mode
        torch.manual_seed(9)
        out = model({**gin, "hit_index": hit})
        lo = loss_fn(out, gt)
        model.zero_grad()
        lo["loss"].backward()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(lo["loss"]).all())
        zs[mode] = [model._last_train.fg[p]["zfinal"].clone() for p in range(2)]
    model.sampler_sdf_mode = "auto"
    for p in range(2):
        assert TOL.within(report(f"train z_vals person {p}: bf16x3 vs f16 sampler", zs["bf16x3"][p], zs["f16"][p].cpu()), TOL.TRAIN_Z_VALS_F16)


def test_training_step_reduces_loss():
    """a few Adam steps (multiply_model.py:130-139 optimiser over model.parameters()) on a fixed batch"""
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    torch.manual_seed(0)
    losses = []
    for it in range(6):
        out = model(gin)
        lo = loss_fn(out, gt)
        opt.zero_grad()
        lo["loss"].backward()
        opt.step()
        losses.append(float(lo["loss"]))
    print("[info] losses", losses)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_two_live_training_graphs_keep_their_gradients_apart():
    """Two training forwards alive before one backward (loss = f(A) + f(B)): every sweep owns its accumulators and gradient
    buffer (TrainState.start_backward), so the summed backward equals the sum of the two separate backwards -- and a retained
    graph swept twice accumulates exactly twice its gradient."""
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(R)]
    in_a = {**gin, "hit_index": hit}
    in_b = {**gin, "hit_index": hit, "uv": gin["uv"].flip(1).contiguous()}
    gt_b = {"rgb": gt["rgb"].flip(1)}

    def fwd(i, g, seed):
        torch.manual_seed(seed)          # the draws of a forward come from torch's global generator
        return loss_fn(model(i), g)["loss"]

    def grads():
        return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    sep = []
    for i, g, seed in ((in_a, gt, 11), (in_b, gt_b, 12)):
        model.zero_grad(set_to_none=True)
        fwd(i, g, seed).backward()
        sep.append(grads())
    model.zero_grad(set_to_none=True)
    la, lb = fwd(in_a, gt, 11), fwd(in_b, gt_b, 12)          # both graphs alive
    (la + lb).backward()
    both = grads()
    worst = 0.0
    for k, g in both.items():
        want = sep[0][k] + sep[1][k]
        rel = float((g - want).norm() / (want.norm() + 1e-12))
        worst = max(worst, rel)
        assert rel < 1e-4 or float((g - want).abs().max()) < 1e-7, f"{k}: {rel:.3e}"
    print(f"[parity] two live graphs vs separate backwards: worst relative difference {worst:.2e} over {len(both)} tensors")
    # a retained graph swept twice: param.grad accumulates g + g
    model.zero_grad(set_to_none=True)
    l1 = fwd(in_a, gt, 11)
    l1.backward(retain_graph=True)
    l1.backward()
    twice = grads()
    for k, g in twice.items():
        want = 2 * sep[0][k]
        rel = float((g - want).norm() / (want.norm() + 1e-12))
        assert rel < 1e-4 or float((g - want).abs().max()) < 1e-7, f"{k} (double sweep): {rel:.3e}"


def test_smpl_surface_and_zero_pose_regularisers_match_the_oracle():
    """The two optional regularisers of the training forward (multiply.py:336-394; weight 0 in the shipped configs, so a reachable
    branch rather than a used one): value, and the gradient of every parameter through loss terms that are ONLY these two (all
    other weights irrelevant: the loss here is their weighted sum + the render loss), against the oracle's restatement under torch
    autograd on the same vertex draws.  The vertex segmentation the reference reads from an asset it does not ship
    (outputs/smpl_vert_segmentation.json) is a synthetic one here; parity of these two terms with the reference itself is unpinned."""
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    model.smpl_surface_weight, model.zero_pose_weight = 1.0, 0.5
    loss_fn.smpl_surface_weight, loss_fn.zero_pose_weight = 1.0, 0.5
    nv = model.smpl_server_list[0].verts_c.reshape(-1, 3).shape[0]
    ids = list(range(nv))
    model.smpl_vertex_part = {"head": ids[:300], "rightHand": ids[300:400], "leftHand": ids[400:500], "rightFoot": ids[500:560],
                              "leftFoot": ids[560:620], "leftHandIndex1": ids[620:640], "rightHandIndex1": ids[640:660]}
    torch.manual_seed(21)
    with torch.no_grad():         # the geometric initialisation zeroes layer 0's conditioning columns (the outputs would not depend on
        for net in model.foreground_implicit_network_list:     # the pose at all: zero_pose_loss == 0): perturb them
            net.lin0.weight_v[:, 39:] += 0.05 * torch.randn_like(net.lin0.weight_v[:, 39:])
            net.lin8.bias[0] += 0.03
    oracle.sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for pp in oracle.persons:
        pp.sd = oracle.sd
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(R)]
    out = model({**gin, "hit_index": hit, "current_epoch": 30})       # epoch 30: both schedules are on, the pose conditioning too
    lo = loss_fn(out, gt)
    model.zero_grad()
    lo["loss"].backward()
    torch.cuda.synchronize()
    graph = model._last_train
    assert float(out["smpl_surface_loss"]) > 0 and float(out["zero_pose_loss"]) > 0
    for v in oracle.sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    draws = _cpu(graph.draws)
    assert draws["person"][0]["surf_idx"].min() >= 660            # the excluded parts are never drawn
    z_given = [graph.fg[p]["zfinal"].cpu() for p in range(2)]
    want = oracle.forward_train(inp, hit, z_given, draws)
    for k in ("smpl_surface_loss", "zero_pose_loss"):
        a, b = float(out[k]), float(want[k])
        print(f"[parity] {k}: gpu {a:.6f} oracle {b:.6f}")
        assert abs(a - b) < 2e-5 * max(1.0, abs(b)), k
    tl = torch.mean(torch.square((inp["smpl_pose"] + 0.01) - inp["smpl_pose"]))
    want.update(fg_rgb_values_each_person_list=[], index_in_surface=graph_index_in_surface(out), epoch=30,
                temporal_loss=tl.reshape(()), sam_mask=gin["sam_mask"].squeeze().cpu())
    lw = loss_fn(want, gt)
    for k in ("loss", "rgb_loss", "eikonal_loss", "bce_loss", "in_shape_loss", "sam_mask_loss", "smpl_surface_loss", "zero_pose_loss"):
        print(f"[parity] loss term {k}: gpu {float(lo[k]):.6f} oracle {float(lw[k]):.6f}")
    # (the perturbed conditioning columns make the network rougher than the geometric initialisation: 3e-4 instead of the 1e-4 of
    # the other training tests; the terms under test agree to 1e-6 above)
    assert abs(float(lo["loss"]) - float(lw["loss"])) < 3e-4 * max(1.0, abs(float(lw["loss"])))
    names = [k for k, v in oracle.sd.items() if v.requires_grad]
    gw = torch.autograd.grad(lw["loss"], [oracle.sd[k] for k in names], allow_unused=True)
    worst = _compare_parameter_gradients(model, oracle, names, gw)
    print(f"[parity] with the two regularisers on: worst relative parameter-gradient error {worst:.3e}")


def graph_index_in_surface(out):
    v = out["index_in_surface"]
    return None if v is None else v.cpu()


def test_eval_after_a_fused_optimizer_step_uses_the_updated_weights():
    """torch's fused Adam updates the parameters WITHOUT bumping their version counters, which the packed-weight caches of the
    f16 kernels are keyed on: the eval render after such a step must not come from the stale pack (Multiply.train() drops the
    caches, training-mode forwards always repack).  The eval pixels after the step must differ from those before it and equal a
    render of a FRESH model loaded with the same state dict."""
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    ein = {k: v for k, v in gin.items() if k not in ("current_epoch", "index_outside", "smpl_pose_last")}
    model.eval()
    with torch.no_grad():
        before = model(ein)["rgb_values"].clone()
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, fused=True)
    for _ in range(2):
        lo = loss_fn(model(gin), gt)
        opt.zero_grad()
        lo["loss"].backward()
        opt.step()
    model.eval()
    with torch.no_grad():
        after = model(ein)["rgb_values"].clone()
    fresh, _, _ = build(H=11, W=11)      # a model that never saw the optimizer: no cache to be stale
    fresh.load_state_dict(model.state_dict())
    fresh.eval()
    with torch.no_grad():
        want = fresh(ein)["rgb_values"]
    print("[info] eval pixels moved by", float((after - before).abs().max()), "; vs fresh model", float((after - want).abs().max()))
    assert float((after - before).abs().max()) > 1e-4
    assert float((after - want).abs().max()) == 0.0


def test_smpl_pose_backward_matches_autograd(smpl_tables):
    """mp_smpl_pose_bwd: adjoint of the bone transforms w.r.t. [scale, transl, thetas, betas] vs torch autograd on the
    oracle's SMPLServer restatement (incl. the |theta + 1e-8| Rodrigues quirk and a zero rotation)."""
    from multiply_amd import hip
    from multiply_amd.smpl import SMPLServer
    betas = np.linspace(-0.5, 0.5, 10).astype(np.float32)
    server = SMPLServer(betas=betas, smpl_tables=smpl_tables)
    so = O.SMPLServerOracle(smpl_tables, betas)
    g = torch.Generator().manual_seed(7)
    prm = torch.zeros(86)
    prm[0] = 1.1
    prm[1:4] = torch.randn(3, generator=g) * 0.3
    prm[4:76] = torch.randn(72, generator=g) * 0.3
    prm[4 + 3 * 5:4 + 3 * 6] = 0.0                      # one joint with exactly zero rotation
    prm[76:] = torch.tensor(betas) + 0.1
    a = torch.randn(24, 4, 4, generator=g)
    a[:, 3, :] = 0
    pg = prm.clone().requires_grad_(True)
    out = so.forward(pg[0], pg[1:4], pg[4:76], pg[76:])
    want = torch.autograd.grad((out["smpl_tfs"] * a).sum(), pg)[0]
    d = prm.cuda()
    verts = torch.empty(6890, 3, device="cuda"); tfs = torch.empty(24, 4, 4, device="cuda"); jn = torch.empty(24, 3, device="cuda")
    server.pose_into(d, verts, tfs, jn)
    assert (tfs.cpu() - out["smpl_tfs"].detach()).abs().max() < 1e-5
    dprm = torch.empty(86, device="cuda")
    da = a.cuda().reshape(24, 16).contiguous()
    rj = server.rest_joints()
    hip.check(hip.lib().mp_smpl_pose_bwd(hip.ptr(server.tables.parents), hip.ptr(d), hip.ptr(server.tfs_c_inv), hip.ptr(rj),
                                         hip.ptr(server.tables.j_shapedirs), hip.ptr(da), hip.ptr(dprm), hip.stream()), "bwd")
    got = dprm.cpu()
    for name, sl in (("scale", slice(0, 1)), ("transl", slice(1, 4)), ("thetas", slice(4, 76)), ("betas", slice(76, 86))):
        e = (got[sl] - want[sl]).abs().max().item() / (want[sl].abs().max().item() + 1e-12)
        print(f"[grad parity] smpl backward {name}: rel-to-max err {e:.3e}")
        assert e < 2e-4, name


def test_training_gradients_to_body_model_params():
    """smpl_pose / smpl_trans / smpl_shape (BodyModelParams in the reference's trainer) receive gradients through the
    canonical warp, the normals' Jacobian and the pose conditioning."""
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(R)]
    for k in ("smpl_pose", "smpl_trans", "smpl_shape"):
        gin[k] = gin[k].clone().requires_grad_(True)
    gin["smpl_pose_last"] = gin["smpl_pose"].detach() + 0.01
    out = model({**gin, "hit_index": hit})
    lo = loss_fn(out, gt)
    lo["loss"].backward()
    torch.cuda.synchronize()
    graph = model._last_train
    oin = dict(inp)
    for k in ("smpl_pose", "smpl_trans", "smpl_shape"):
        oin[k] = inp[k].clone().requires_grad_(True)
    z_given = [graph.fg[p]["zfinal"].cpu() for p in range(2)]
    want = oracle.forward_train(oin, hit, z_given, _cpu(graph.draws))
    tl = torch.mean(torch.square(inp["smpl_pose"] + 0.01 - oin["smpl_pose"]))
    want.update(fg_rgb_values_each_person_list=[], index_in_surface=None, epoch=301, temporal_loss=tl,
                smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1), sam_mask=gin["sam_mask"].squeeze().cpu())
    lw = loss_fn(want, gt)
    gw = torch.autograd.grad(lw["loss"], [oin[k] for k in ("smpl_pose", "smpl_trans", "smpl_shape")])
    assert abs(float(lo["loss"]) - float(lw["loss"])) < 1e-4
    for k, w in zip(("smpl_pose", "smpl_trans", "smpl_shape"), gw):
        a = gin[k].grad.cpu()
        e = (a - w).abs().max().item() / (w.abs().max().item() + 1e-12)
        print(f"[grad parity] d loss / d {k}: rel-to-max err {e:.3e} (|want|max {w.abs().max().item():.3e})")
        assert e < 5e-3, k


def test_zero_pose_regulariser_reaches_the_pose_under_optimisation():
    """zero_pose_weight > 0 while the body-model inputs are being optimised (round 6; the branch raised before): the term
    reaches smpl_pose through the pose conditioning of the network evaluations (multiply.py:377-394, cond = pose[3:] / pi).
    d loss / d smpl_pose against the oracle under torch autograd; the surface term in the same situation still refuses loudly."""
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    model.zero_pose_weight = loss_fn.zero_pose_weight = 0.5
    torch.manual_seed(21)
    with torch.no_grad():         # (see the test above: the geometric initialisation's conditioning columns are zero)
        for net in model.foreground_implicit_network_list:
            net.lin0.weight_v[:, 39:] += 0.05 * torch.randn_like(net.lin0.weight_v[:, 39:])
    oracle.sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for pp in oracle.persons:
        pp.sd = oracle.sd
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(R)]
    gin["smpl_pose"] = gin["smpl_pose"].clone().requires_grad_(True)
    gin["smpl_pose_last"] = gin["smpl_pose"].detach() + 0.01
    out = model({**gin, "hit_index": hit, "current_epoch": 30})
    lo = loss_fn(out, gt)
    lo["loss"].backward()
    torch.cuda.synchronize()
    graph = model._last_train
    assert float(out["zero_pose_loss"]) > 0
    oin = dict(inp)
    oin["smpl_pose"] = inp["smpl_pose"].clone().requires_grad_(True)
    z_given = [graph.fg[p]["zfinal"].cpu() for p in range(2)]
    want = oracle.forward_train(oin, hit, z_given, _cpu(graph.draws))
    # (epoch 30: no temporal term -- the reference adds it from epoch 251 on, multiply.py:242-243)
    want.update(fg_rgb_values_each_person_list=[], index_in_surface=graph_index_in_surface(out), epoch=30, temporal_loss=torch.zeros(1),
                smpl_surface_loss=torch.zeros(1), sam_mask=gin["sam_mask"].squeeze().cpu())
    lw = loss_fn(want, gt)
    assert abs(float(out["zero_pose_loss"]) - float(want["zero_pose_loss"])) < 2e-5
    # the regulariser's own share of the pose gradient (everything else is covered by the test above)
    (g_all,) = torch.autograd.grad(lw["loss"], [oin["smpl_pose"]], retain_graph=True)
    (g_zp,) = torch.autograd.grad(want["zero_pose_loss"].sum(), [oin["smpl_pose"]])
    a = gin["smpl_pose"].grad.cpu()
    e = (a - g_all).abs().max().item() / (g_all.abs().max().item() + 1e-12)
    share = g_zp.abs().max().item() * 0.5 / (g_all.abs().max().item() + 1e-12)
    print(f"[grad parity] d loss / d smpl_pose with the zero-pose term: rel-to-max err {e:.3e}; the term's share of the gradient {share:.2e}")
    assert e < 5e-3 and share > 5 * e
    model.smpl_surface_weight = 1.0
    nv = model.smpl_server_list[0].verts_c.reshape(-1, 3).shape[0]
    model.smpl_vertex_part = {"head": list(range(300)), "rightHand": [], "leftHand": [], "rightFoot": [], "leftFoot": [],
                              "leftHandIndex1": [], "rightHandIndex1": []}
    with pytest.raises(NotImplementedError):
        model({**gin, "hit_index": hit, "current_epoch": 30})


def test_forward_mode_and_reverse_mode_training_agree(monkeypatch):
    """The two hand-written differentiation schemes of the SDF net (reverse-over-reverse, default; forward-mode + reverse)
    are independent implementations: same draws -> same outputs and the same gradient for every parameter."""
    from multiply_amd import train as T
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup()
    R = inp["uv"].shape[1]
    hit = [torch.arange(R), torch.arange(R)]
    res = {}
    draws = None
    for mode in ("reverse", "forward"):
        monkeypatch.setattr(T, "SDF_TRAIN_MODE", mode)
        cx = model._setup({**gin, "hit_index": hit}, -1, False)
        if draws is None:
            draws = T.make_draws(model, cx, None)
        out = T.forward_train(model, {**gin, "hit_index": hit}, draws=draws)
        lo = loss_fn(out, gt)
        model.zero_grad()
        lo["loss"].backward()
        torch.cuda.synchronize()
        res[mode] = (float(lo["loss"]), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    assert abs(res["reverse"][0] - res["forward"][0]) < 1e-6
    worst = 0.0
    for k, g in res["reverse"][1].items():
        h = res["forward"][1][k]
        rel = float((g - h).norm() / (h.norm() + 1e-12))
        worst = max(worst, rel)
        assert rel < 2e-3 or float((g - h).abs().max()) < 1e-7, k
    print(f"[parity] forward-mode vs reverse-mode training gradients: worst relative difference {worst:.3e}")
