"""Pins the CPU oracle (oracle/multiply_oracle.py) against golden vectors produced by the reference's own
modules (tests/golden/make_golden.py -> reference_eval.npz).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import multiply_oracle as O
from tests.util import seeded_networks, state_checksum, t32


@pytest.fixture(scope="module")
def nets(golden):
    m, opt = seeded_networks(2, 0)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    # proves our construction yields exactly the reference's initial weights (same RNG consumption)
    assert sorted(sd.keys()) == list(golden["state_keys"])
    assert state_checksum(sd) == float(golden["weights_checksum"])
    return sd


@pytest.fixture(scope="module")
def scene(golden, smpl_tables):
    sp = t32(golden["scene_smpl_params"])
    T = O.SMPLTables(smpl_tables)
    servers = [O.SMPLServerOracle(T, sp[0, p, 76:].numpy()) for p in range(2)]
    outs = [servers[p].forward(sp[0, p, 0], sp[0, p, 1:4], sp[0, p, 4:76], sp[0, p, 76:]) for p in range(2)]
    return sp, servers, outs


def close(a, b, atol, rtol=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b) - rtol * np.abs(b)
    assert err.max() <= atol, f"max err {np.abs(a - b).max():.3e} (atol {atol}, rtol {rtol})"


def test_g1_camera_rays(golden):
    d, c = O.get_camera_rays(t32(golden["g1_uv"])[0], t32(golden["g1_pose"])[0], t32(golden["g1_K"])[0])
    close(d, golden["g1_dirs"][0], 2e-7)
    close(c, golden["g1_cam"][0], 0)


def test_g2_smpl_server(golden, smpl_tables):
    v = golden["g2_scale_trans_betas"]
    sv = O.SMPLServerOracle(smpl_tables, v[76:])
    close(sv.verts_c, golden["g2_verts_c"], 1e-6)
    close(sv.tfs_c_inv, golden["g2_tfs_c_inv"], 1e-5)
    for i in range(3):
        o = sv.forward(t32(v[0] * (1.0 + 0.1 * i)), t32(v[1:4]), t32(golden[f"g2_thetas{i}"]), t32(v[76:]))
        close(o["smpl_verts"], golden[f"g2_smpl_verts{i}"], 2e-6)
        close(o["smpl_tfs"], golden[f"g2_smpl_tfs{i}"], 1e-5)
        close(o["smpl_jnts"], golden[f"g2_smpl_jnts{i}"], 2e-6)


def test_g3_deformer(golden, scene):
    sp, servers, outs = scene
    x = t32(golden["g3_x"])
    xc, outl = O.deform_inverse(x, outs[0]["smpl_tfs"], outs[0]["smpl_verts"], servers[0].weights)
    assert (outl.numpy() == golden["g3_outlier"]).all()
    close(xc, golden["g3_xc"], 2e-5)
    w_c, _, _ = O.query_weights(xc, servers[0].verts_c, servers[0].weights)
    xd = O.skinning(xc, w_c, outs[0]["smpl_tfs"], inverse=False)
    close(xd, golden["g3_xd"], 2e-5)


def test_g4_networks(golden, nets):
    x, cond = t32(golden["g4_x"]), t32(golden["g4_cond"])
    out = O.implicit_forward(nets, "foreground_implicit_network_list.0.", x, cond, multires=6)
    close(out, golden["g4_imp"], 2e-5)
    xg = x.clone().requires_grad_(True)
    s = O.implicit_forward(nets, "foreground_implicit_network_list.0.", xg, cond, multires=6)[:, :1]
    g = torch.autograd.grad(s, xg, torch.ones_like(s))[0]
    close(g, golden["g4_grad"], 2e-5)
    rgb = O.rendering_forward_pose_no_view(nets, "foreground_rendering_network_list.0.", x, t32(golden["g4_nrm"]),
                                           cond, out[:, 1:])
    close(rgb, golden["g4_rgb"], 1e-5)
    code = nets["frame_latent_encoder.weight"][3]
    bo = O.implicit_forward(nets, "bg_implicit_network.", t32(golden["g4_bg_x"]), code, multires=10)
    close(bo, golden["g4_bg_imp"], 2e-5)
    brgb = O.rendering_forward_nerf_frame(nets, "bg_rendering_network.", t32(golden["g4_bg_view"]), bo[:, 1:], code)
    close(brgb, golden["g4_bg_rgb"], 1e-5)


def test_g5_density(golden):
    s = t32(golden["g5_sdf"])
    for i, b in enumerate([0.1, 0.01, 1e-3]):
        close(O.laplace_density(s, torch.tensor(b)), golden[f"g5_sigma{i}"], 0.0, rtol=1e-6)


def test_g6_sampler(golden, nets, scene):
    sp, servers, outs = scene
    cfg = O.SamplerCfg()
    # the golden rays come from a 64x64 camera (make_golden.py G6)
    from multiply_amd.synthetic import make_scene
    sc = make_scene(2, seed=0, H=64, W=64)
    dirs, cam = O.get_camera_rays(t32(sc["uv"])[0], t32(sc["pose"])[0], t32(sc["intrinsics"])[0])
    sel = torch.tensor(golden["g6_sel"])
    beta0 = nets["density.beta"].abs() + 1e-4
    for p in range(2):
        person = O.PersonOracle(nets, p, servers[p])
        cond = sp[0, p, 7:76] / np.pi
        fn = lambda pts: person.sdf_func(pts, cond, outs[p]["smpl_tfs"], outs[p]["smpl_verts"], True)[0]
        z, iters = O.error_bound_sample(cfg, dirs[sel], cam[None].expand(len(sel), -1), fn, beta0)
        close(z, golden[f"g6_z{p}"], 2e-4)
        close(O.bg_depths(cfg, len(sel)), golden[f"g6_zbg{p}"], 1e-7)


def test_g7_g8_shading_and_compositing(golden, nets, scene, smpl_tables):
    sp, servers, outs = scene
    model = O.MultiplyOracle(nets, smpl_tables, sp[0, :, 76:].numpy())
    p = 0
    cond = sp[0, p, 7:76] / np.pi
    person = model.persons[p]
    pts = t32(golden["g7_pts"])
    with torch.no_grad():
        sdf, xc, _ = person.sdf_func(pts, cond, outs[p]["smpl_tfs"], outs[p]["smpl_verts"], True)
    close(sdf, golden["g7_sdf"], 2e-5)
    close(xc, golden["g7_xc"], 2e-5)
    rgb, nrm, _ = model.shade(person, t32(golden["g7_xc"]), cond, outs[p]["smpl_tfs"])
    close(rgb.detach(), golden["g7_rgb"], 5e-5)
    close(nrm.detach(), golden["g7_nrm"], 2e-4)
    # compositing: the oracle's packed (nerfacc-style) path vs the reference's dense path for P=1, all rays hit
    z = t32(golden["g6_z0"])
    zz, zmax = z[:, :-1], z[:, -1]
    S = zz.shape[1]
    R = zz.shape[0]
    beta = model.beta()
    w, bgT = O.dense_volume_rendering(zz, zmax, t32(golden["g7_sdf"]), beta)
    close(w, golden["g8_w"], 2e-6)
    close(bgT, golden["g8_bgT"], 2e-6)
    fg, nv, acc, accp, bg_T = O.packed_composite(
        R, [torch.arange(R)], [zz], [zmax], [t32(golden["g7_sdf"]).reshape(R, S)],
        [t32(golden["g7_rgb"]).reshape(R, S, 3)], [t32(golden["g7_nrm"]).reshape(R, S, 3)], beta, [0])
    close(fg, golden["g8_fg"], 5e-6)
    close(nv, golden["g8_nrm"], 5e-6)
    close(acc, golden["g8_acc"], 5e-6)
    close(accp[:, 0], golden["g8_acc"], 5e-6)
    # nerfacc quirk (multiply.py:457-463): bg transmittance omits the last sample's alpha
    sd_last = O.laplace_density(t32(golden["g7_sdf"]).reshape(R, S)[:, -1], beta) * (zmax - zz[:, -1])
    close(bg_T * torch.exp(-sd_last), golden["g8_bgT"], 1e-5)
    # background branch
    d_s = None
    from multiply_amd.synthetic import make_scene
    sc = make_scene(2, seed=0, H=64, W=64)
    dirs, cam = O.get_camera_rays(t32(sc["uv"])[0], t32(sc["pose"])[0], t32(sc["intrinsics"])[0])
    sel = torch.tensor(golden["g6_sel"])
    zbg = torch.flip(O.bg_depths(model.cfg, R), dims=[-1])
    bp = O.depth2pts_outside(cam[None, None].expand(R, 32, -1), dirs[sel][:, None].expand(-1, 32, -1), zbg)
    close(bp, golden["g8_bg_pts"], 2e-6)
    bg = model.background(dirs[sel], cam[None].expand(R, -1), nets["frame_latent_encoder.weight"][5])
    close(bg, golden["g8_bg_rgb"], 2e-5)
    close(fg + t32(golden["g8_bgT"])[:, None] * bg, golden["g10_rgb_dense"], 3e-5)


@pytest.mark.parametrize("P", [2, 3])
def test_multi_person_packing_against_a_ray_by_ray_loop(P):
    """The oracle's packed multi-person compositing (multiply.py:425-480; nerfacc is an absent third party, the reference itself
    holds no dense path for more than one person) against a statement from first principles written as plain loops in float64:
    for every ray, the intervals [t_start, t_end) of ALL persons that hit it -- each person's intervals come from ITS OWN depths
    (multiply.py:428-431) -- are visited in order of t_end; alpha = 1 - exp(-sigma dt); the transmittance in front of an interval
    is the product of (1 - alpha) of the intervals visited before it; weights = T alpha; the background sees the EXCLUSIVE
    transmittance of the ray's last interval (multiply.py:457-463) and 1 on rays without samples.  Ragged hit sets, persons
    interleaved in depth, rays hit by one / all / no person."""
    import math
    from oracle import multiply_oracle as O
    g = torch.Generator().manual_seed(40 + P)
    R, S, beta = 37, 9, 0.07
    hit, zs, zmaxs, sdfs, rgbs, nrms = [], [], [], [], [], []
    for p in range(P):
        keep = torch.rand(R, generator=g) < (0.75 if p else 0.6)
        keep[3] = False                                                 # ray 3: nobody
        keep[5] = True                                                  # ray 5: everybody
        h = torch.nonzero(keep).flatten()
        z = torch.sort(0.5 + 3.0 * torch.rand(len(h), S, generator=g), dim=1).values
        hit.append(h); zs.append(z); zmaxs.append(z[:, -1] + 0.05 + torch.rand(len(h), generator=g))
        sdfs.append(0.3 * torch.randn(len(h), S, generator=g)); rgbs.append(torch.rand(len(h), S, 3, generator=g))
        nrms.append(torch.randn(len(h), S, 3, generator=g))
    got = O.packed_composite(R, hit, zs, zmaxs, sdfs, rgbs, nrms, torch.tensor(beta), list(range(P)))
    want_rgb, want_nrm = torch.zeros(R, 3, dtype=torch.float64), torch.zeros(R, 3, dtype=torch.float64)
    want_acc, want_accp, want_T = torch.zeros(R, dtype=torch.float64), torch.zeros(R, P, dtype=torch.float64), torch.ones(R, dtype=torch.float64)
    for r in range(R):
        ivs = []
        for p in range(P):
            rows = torch.nonzero(hit[p] == r).flatten()
            if len(rows) == 0:
                continue
            k = int(rows[0])
            zz = torch.cat([zs[p][k], zmaxs[p][k:k + 1]]).double()
            for s in range(S):
                sd = float(sdfs[p][k, s])
                sigma = (1.0 / beta) * (0.5 + 0.5 * math.copysign(1.0, sd) * math.expm1(-abs(sd) / beta)) if sd != 0 else 0.5 / beta
                ivs.append((float(zz[s + 1]), float(zz[s]), sigma, rgbs[p][k, s].double(), nrms[p][k, s].double(), p))
        ivs.sort(key=lambda t: t[0])
        T = 1.0
        for n, (te, ts, sigma, c, nv, p) in enumerate(ivs):
            a = 1.0 - math.exp(-sigma * (te - ts))
            w = T * a
            want_rgb[r] += w * c; want_nrm[r] += w * nv; want_acc[r] += w; want_accp[r, p] += w
            if n == len(ivs) - 1:
                want_T[r] = T                                           # exclusive: the last interval's own alpha is not applied
            T *= 1.0 - a
    names = ("rgb", "normal", "acc", "acc_person", "bg_T")
    for nme, a, b in zip(names, got, (want_rgb, want_nrm, want_acc, want_accp, want_T)):
        err = float((a.double() - b).abs().max())
        assert err < 2e-6, (nme, err)
    assert float(got[4][3]) == 1.0 and float(got[2][3]) == 0.0          # the ray nobody hits
