"""Pins the TRAINING-mode arithmetic of the CPU oracle against vectors produced by the reference's own modules with
model.training = True and every random draw recorded (tests/golden/make_train_golden.py -> reference_train.npz):
the sampler with training draws, the eikonal points, the create_graph normals / colours, and -- through the reference's
dense compositing, its background branch and its Loss -- the gradient of a training loss w.r.t. EVERY parameter.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import multiply_oracle as O
from tests.util import seeded_networks, state_checksum, t32

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def tg():
    return np.load(os.path.join(HERE, "golden", "reference_train.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def setup(tg, smpl_tables):
    m, opt = seeded_networks(2, 0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    assert state_checksum(sd) == float(tg["weights_checksum"])       # the reference's initial weights, bit for bit
    sp = t32(tg["scene_smpl_params"])
    oracle = O.MultiplyOracle(sd, smpl_tables, sp[0, :, 76:].numpy())
    from multiply_amd.synthetic import make_scene
    sc = make_scene(2, seed=0, H=64, W=64)
    assert np.array_equal(sc["smpl_params"], tg["scene_smpl_params"])
    dirs, cam1 = O.get_camera_rays(t32(sc["uv"])[0], t32(sc["pose"])[0], t32(sc["intrinsics"])[0])
    sel = torch.as_tensor(tg["sel"]).long()
    d, c = dirs[sel], cam1[None].expand(len(sel), -1)
    so = oracle.servers[0].forward(sp[0, 0, 0], sp[0, 0, 1:4], sp[0, 0, 4:76], sp[0, 0, 76:])
    cond = sp[0, 0, 7:76] / np.pi
    return oracle, sp, d, c, so, cond


def close(a, b, atol, rtol=0.0, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b) - rtol * np.abs(b)
    assert err.max() <= atol, f"{what}: max err {np.abs(a - b).max():.3e} (atol {atol}, rtol {rtol})"


def test_t1_sampler_training_draws(tg, setup):
    """ErrorBoundSampler.get_z_vals with model.training: stratified t_rand, random u for the final inverse CDF, randperm
    for the extra samples, jittered inverse-sphere depths -- fed the reference's recorded draws in its call order."""
    oracle, sp, d, c, so, cond = setup
    person = oracle.persons[0]
    fn = lambda pts: person.sdf_func(pts, cond, so["smpl_tfs"], so["smpl_verts"], eval_mode=False)[0]
    draws = dict(t_rand=t32(tg["t1_t_rand"]), u_final=t32(tg["t1_u_final"]),
                 extra_idx=torch.as_tensor(tg["t1_perm"]).long()[:oracle.cfg.N_samples_extra])
    with torch.no_grad():
        z, iters = O.error_bound_sample(oracle.cfg, d, c, fn, oracle.beta(), draws)
    assert iters * oracle.cfg.N_samples_eval == len(tg["t1_perm"])       # same number of sampler iterations (randperm length)
    close(z, tg["t1_z"], 2e-4, what="training z_vals")     # as the eval sampler pin (G6): the inverse CDF amplifies fp32 reordering
    close(O.bg_depths(oracle.cfg, d.shape[0], t32(tg["t1_bg_rand"])), tg["t1_zbg"], 1e-7, what="jittered inverse depths")


def _train_pieces(tg, setup):
    oracle, sp, d, c, so, cond = setup
    for v in oracle.sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    person = oracle.persons[0]
    zz = t32(tg["t1_z"])
    zmax, z = zz[:, -1], zz[:, :-1]
    S = z.shape[1]
    pts = (c[:, None, :] + z[:, :, None] * d[:, None, :]).reshape(-1, 3)
    with torch.no_grad():
        x_c, _ = O.deform_inverse(pts, so["smpl_tfs"], so["smpl_verts"], person.server.weights)
    rgb, nrm, sdf = oracle.shade(person, x_c, cond, so["smpl_tfs"], create_graph=True)
    eik_idx = torch.as_tensor(tg["t2_eik_perm"]).long()[:512]
    xe = (person.server.verts_c[eik_idx] + t32(tg["t2_eik_noise"]) * 0.01).detach().requires_grad_(True)
    se = person.implicit(xe, cond)[:, :1]
    gth = torch.autograd.grad(se, xe, torch.ones_like(se), create_graph=True)[0]
    return dict(z=z, zmax=zmax, S=S, x_c=x_c, rgb=rgb, nrm=nrm, sdf=sdf, xe=xe, gth=gth)


def test_t2_per_sample_training_arithmetic(tg, setup):
    """sdf_func_with_smpl_deformer (no outlier override), eikonal points + gradient(create_graph), forward_gradient /
    get_rbg_value with create_graph=True (multiply.py:137-151, 322-331, 600-661)."""
    t = _train_pieces(tg, setup)
    close(t["x_c"], tg["t2_xc"], 2e-5, what="canonical points")
    close(t["sdf"].detach(), tg["t2_sdf"], 2e-5, what="sdf")
    close(t["xe"].detach(), tg["t2_eik_points"], 1e-6, what="eikonal points")
    close(t["gth"].detach(), tg["t2_grad_theta"], 5e-5, what="grad_theta")
    close(t["nrm"].detach(), tg["t2_nrm"], 3e-4, what="normals")
    close(t["rgb"].detach(), tg["t2_rgb"], 5e-5, what="rgb")


def test_t3_training_loss_and_every_parameter_gradient(tg, setup):
    """Dense compositing + background + Loss on the reference side, the same assembled from the oracle's pieces and
    multiply_amd.loss here: loss terms and the gradient of the loss w.r.t. every parameter (norm and a seeded projection
    per tensor; full tensors for the small ones)."""
    from multiply_amd.config import load_config
    from multiply_amd.loss import Loss
    oracle, sp, d, c, so, cond = setup
    t = _train_pieces(tg, setup)
    S = t["S"]
    w, bgT = O.dense_volume_rendering(t["z"], t["zmax"], t["sdf"], oracle.beta())
    fg = (w[:, :, None] * t["rgb"].reshape(-1, S, 3)).sum(1)
    acc = w.sum(-1)
    code = oracle.sd["frame_latent_encoder.weight"][5]
    bg = oracle.background_train(d, c, code, t32(tg["t1_bg_rand"]))
    rgb_values = fg + bgT[:, None] * bg
    close(acc.detach(), tg["t3_acc"], 5e-5, what="acc")
    close(bgT.detach(), tg["t3_bgT"], 5e-5, what="bg transmittance (dense path)")
    close(bg.detach(), tg["t3_bg_rgb"], 5e-5, what="background colour")
    close(rgb_values.detach(), tg["t3_rgb_values"], 1e-4, what="rgb_values")
    loss_fn = Loss(load_config().loss)
    mo = dict(fg_rgb_values_each_person_list=[], rgb_values=rgb_values, grad_theta=t["gth"][None], acc_map=acc,
              index_in_surface=None, index_off_surface=None, epoch=301, temporal_loss=torch.zeros(1),
              smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1), sam_mask=t32(tg["t3_sam"]),
              acc_person_list=acc[:, None])
    lo = loss_fn(mo, {"rgb": t32(tg["t3_gt_rgb"])})
    for k in ("loss", "rgb_loss", "eikonal_loss", "bce_loss", "sam_mask_loss"):
        a, b = float(lo[k].detach()), float(tg["t3_loss_" + k][0])
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (k, a, b)
    keys = [str(k) for k in tg["t3_grad_keys"]]
    allk = [k for k, v in oracle.sd.items() if v.requires_grad]
    allg = dict(zip(allk, torch.autograd.grad(lo["loss"], [oracle.sd[k] for k in allk], allow_unused=True)))
    grads = [allg[k] for k in keys]
    prj = torch.Generator().manual_seed(99)
    worst = 0.0
    for k, g, n_ref, p_ref in zip(keys, grads, tg["t3_grad_norm"], tg["t3_grad_proj"]):
        assert g is not None, k
        r = torch.randn(oracle.sd[k].shape, generator=prj)
        n, p = float(g.double().norm()), float((g.double() * r.double()).sum())
        # fp32 graphs in a different operation order: 2e-3 relative on the norm; the projection is compared against the norm
        assert abs(n - n_ref) <= 2e-3 * n_ref + 1e-9, (k, n, n_ref)
        assert abs(p - p_ref) <= 2e-3 * n_ref * float(r.double().norm()) / max(np.sqrt(r.numel()), 1.0) + 1e-9, (k, p, p_ref)
        worst = max(worst, abs(n - n_ref) / (n_ref + 1e-12))
        full = "t3_grad_full_" + k
        if full in tg.files:
            close(g, tg[full], 2e-3 * float(np.abs(tg[full]).max()) + 1e-9, what="gradient of " + k)
    # every parameter the reference's loss reaches is reached here, and no other
    assert {k for k, g in allg.items() if g is not None} == set(keys)
    print(f"worst relative gradient-norm deviation {worst:.2e} over {len(keys)} tensors")
