"""GPU parity of the geometry kernels (SMPL posing, exact nearest-vertex warp, rays) against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import multiply_oracle as O
from tests.util import t32

pytestmark = pytest.mark.gpu


def err(name, got, want):
    e = (torch.as_tensor(got).double().cpu() - torch.as_tensor(want).double()).abs().max().item()
    print(f"[parity] {name}: max abs err {e:.3e}")
    return e


@pytest.fixture(scope="module")
def server(smpl_tables):
    from multiply_amd.smpl import SMPLServer
    from multiply_amd.synthetic import make_scene
    sc = make_scene(2, seed=0, H=16, W=16)
    sp = t32(sc["smpl_params"])
    return SMPLServer(gender="male", betas=sp[0, 0, 76:].numpy(), smpl_tables=smpl_tables), sp


def test_smpl_server(server, smpl_tables):
    sv, sp = server
    ref = O.SMPLServerOracle(smpl_tables, sp[0, 0, 76:].numpy())
    assert err("verts_c", sv.verts_c[0], ref.verts_c) < 2e-6
    assert err("tfs_c_inv", sv.tfs_c_inv, ref.tfs_c_inv) < 2e-5
    rng = np.random.RandomState(0)
    for i in range(3):
        th = t32(rng.normal(0, 0.3 * i, 72))
        scale, tr = t32(1.0 + 0.1 * i), sp[0, 0, 1:4]
        want = ref.forward(scale, tr, th, sp[0, 0, 76:])
        got = sv(scale.cuda(), tr.cuda(), th.cuda(), sp[0, 0, 76:].cuda())
        assert err(f"smpl_verts[{i}]", got["smpl_verts"][0], want["smpl_verts"]) < 5e-6
        assert err(f"smpl_tfs[{i}]", got["smpl_tfs"][0], want["smpl_tfs"]) < 2e-5
        assert err(f"smpl_jnts[{i}]", got["smpl_jnts"][0], want["smpl_jnts"]) < 5e-6


def test_deformer_inverse_and_jacobian(server, smpl_tables):
    from multiply_amd.deformer import SMPLDeformer
    sv, sp = server
    ref = O.SMPLServerOracle(smpl_tables, sp[0, 0, 76:].numpy())
    want_pose = ref.forward(sp[0, 0, 0], sp[0, 0, 1:4], sp[0, 0, 4:76], sp[0, 0, 76:])
    got_pose = sv(sp[0, 0, 0].cuda(), sp[0, 0, 1:4].cuda(), sp[0, 0, 4:76].cuda(), sp[0, 0, 76:].cuda())
    d = SMPLDeformer(betas=sp[0, 0, 76:].numpy(), gender="male", server=sv)
    rng = np.random.RandomState(2)
    pick = rng.randint(0, 6890, 5000)
    x = want_pose["smpl_verts"][pick] + t32(rng.normal(0, 0.08, (5000, 3)))
    x[:200] += 1.5      # far-away points: exact search must still find the true nearest vertex
    xc_w, out_w = O.deform_inverse(x, want_pose["smpl_tfs"], want_pose["smpl_verts"], ref.weights)
    xc_g, out_g = d.forward(x.cuda(), got_pose["smpl_tfs"], return_weights=False, inverse=True,
                            smpl_verts=got_pose["smpl_verts"])
    assert (out_g.cpu() == out_w).all()
    assert err("x_c", xc_g, xc_w) < 2e-5
    w_c, _, _ = O.query_weights(xc_w, ref.verts_c, ref.weights)
    J = torch.einsum("pn,nij->pij", w_c, want_pose["smpl_tfs"])[:, :3, :3]
    jinv = d.forward_skinning_jacobian_inverse(xc_w.cuda(), got_pose["smpl_tfs"])
    assert err("J^-1", jinv, J.inverse()) < 5e-5


def test_nearest_vertex_ties_resolve_to_the_lowest_id_in_every_search_mode():
    """deformer.py:39 (knn_points, K = 1) is an argmin over the vertices in their original order: of several vertices at EXACTLY the
    same fp32 distance the lowest id wins.  A vertex set on a coarse dyadic lattice (many exact duplicates and mirror pairs; all the
    arithmetic is exact in fp32, so the brute-force reference does not depend on the order of its operations) drives the training search
    (mode 0: unbounded, coarse clusters), the eval search (mode 1: 0.1 cap, fine clusters) and the seeded canonical search of the
    Jacobian kernel through the C ABI.  The blend table is built so that x_c reveals which vertex was taken (x_c.x = x.x + id)."""
    import ctypes as C
    from multiply_amd import hip
    from multiply_amd.smpl import knn_cluster_perm
    L = hip.lib()
    rng = np.random.RandomState(7)
    V = 6890
    verts = (rng.randint(-8, 9, (V, 3)) / 8.0).astype(np.float32)                 # 17^3 = 4913 lattice sites for 6890 vertices
    n = 20000
    base = verts[rng.randint(0, V, n)]
    x = (base + rng.randint(-2, 3, (n, 3)) / 32.0).astype(np.float32)              # within 0.11 of a vertex: eval's capped search finds it
    x[:4000] = (rng.randint(-40, 41, (4000, 3)) / 16.0).astype(np.float32)         # and points far outside (unbounded search only)
    dev = "cuda"
    vt, xt = torch.from_numpy(verts).to(dev), torch.from_numpy(x).to(dev)
    d2 = ((xt[:, None, :] - vt[None, :, :]) ** 2).sum(-1)                          # exact
    dmin = d2.min(1, keepdim=True).values
    ids = torch.arange(V, device=dev)[None, :].expand_as(d2)
    want = torch.where(d2 == dmin, ids, torch.full_like(ids, V)).min(1).values
    n_tied = int(((d2 == dmin).sum(1) > 1).sum())
    assert n_tied > 2000                                                            # the case under test is well populated
    perm = torch.from_numpy(knn_cluster_perm(verts)).to(dev)
    vs = torch.empty(hip.KNN_NC * hip.KNN_CLUSTER, 4, dtype=torch.float32, device=dev)
    cb = torch.empty(hip.KNN_CB_ROWS, 4, dtype=torch.float32, device=dev)
    hip.check(L.mp_knn_build(hip.ptr(vt), hip.ptr(perm), hip.ptr(vs), hip.ptr(cb), hip.stream()), "mp_knn_build")
    btab = torch.zeros(V, 3, 4, dtype=torch.float32, device=dev)                   # x_c = I (x - c): I = identity, c = (-id, 0, 0)
    btab[:, 0, 0] = btab[:, 1, 1] = btab[:, 2, 2] = 1.0
    btab[:, 0, 3] = -torch.arange(V, device=dev, dtype=torch.float32)
    for mode in (0, 1):
        xc = torch.full((n, 3), float("nan"), dtype=torch.float32, device=dev)
        outl = torch.empty(n, dtype=torch.uint8, device=dev)
        sdf = torch.zeros(n, dtype=torch.float32, device=dev)
        hip.check(L.mp_warp_inverse(hip.ptr(xt), None, None, None, None, None, 0, 1, n, hip.ptr(vs), hip.ptr(cb), hip.ptr(btab),
                                    mode, None, None, hip.ptr(xc), hip.ptr(outl), hip.ptr(sdf), None, None, None, hip.stream()),
                  "mp_warp_inverse")
        torch.cuda.synchronize()
        got = (xc[:, 0] - xt[:, 0]).round().long()
        inside = outl == 0 if mode == 1 else torch.ones(n, dtype=torch.bool, device=dev)
        want_out = dmin[:, 0].clamp(max=4.0).sqrt() > 0.1                           # deformer.py:41-49
        assert torch.equal(outl.bool(), want_out), f"mode {mode}: outlier flags"
        assert int(inside.sum()) > 10000
        bad = (got != want) & inside
        assert not bool(bad.any()), f"mode {mode}: {int(bad.sum())} of {int(inside.sum())} points took another vertex than the lowest id at the minimum distance"
    # the seeded search of the Jacobian kernel (explicit points): seed = any vertex at the minimum distance, not the lowest id
    seed = torch.where(d2 == dmin, ids, torch.full_like(ids, -1)).max(1).values.int()
    jinv = torch.empty(n, 9, dtype=torch.float32, device=dev)
    nn = torch.empty(n, dtype=torch.int32, device=dev)
    hip.check(L.mp_warp_jacobian(hip.ptr(xt), None, None, 0, 0, n, hip.ptr(vs), hip.ptr(cb), hip.ptr(btab), hip.ptr(jinv), hip.ptr(nn),
                                 hip.ptr(seed), hip.ptr(vt), hip.stream()), "mp_warp_jacobian")
    torch.cuda.synchronize()
    assert torch.equal(nn.long(), want)


def test_rays(golden):
    import ctypes as C
    from multiply_amd import hip
    uv, pose, K = t32(golden["g1_uv"])[0], t32(golden["g1_pose"])[0], t32(golden["g1_K"])[0]
    R = uv.shape[0]
    dirs = torch.empty(R, 3, device="cuda"); far = torch.empty(R, device="cuda")
    uv_d, K_d, pose_d = uv.cuda().contiguous(), K.cuda().reshape(16).contiguous(), pose.cuda().reshape(16).contiguous()
    hip.check(hip.lib().mp_ray_setup(hip.ptr(uv_d), hip.ptr(K_d), hip.ptr(pose_d), R, C.c_float(3.0), hip.ptr(dirs),
                                     hip.ptr(far), hip.stream()), "mp_ray_setup")
    assert err("ray dirs vs reference golden", dirs, golden["g1_dirs"][0]) < 5e-7
    d, c = O.get_camera_rays(uv, pose, K)
    assert err("far", far, O.sphere_far(c[None].expand(R, -1), d, 3.0)[:, 0]) < 5e-6


def test_general_k_weight_query_and_skinning(smpl_tables):
    """deformer.K = 7 (the trainer's mesh-export setting): weights, forward and inverse skinning vs the oracle."""
    from multiply_amd.smpl import SMPLServer
    from multiply_amd.deformer import SMPLDeformer
    betas = np.zeros(10, dtype=np.float32)
    server = SMPLServer(betas=betas, smpl_tables=smpl_tables)
    d = SMPLDeformer(betas=betas, server=server)
    so = O.SMPLServerOracle(smpl_tables, betas)
    g = torch.Generator().manual_seed(11)
    idx = torch.randint(0, 6890, (3000,), generator=g)
    pts = so.verts_c[idx] + torch.randn(3000, 3, generator=g) * 0.03
    th = torch.randn(72, generator=g) * 0.2
    out = so.forward(torch.tensor(1.0), torch.tensor([0.1, -0.2, 0.05]), th, torch.tensor(betas))
    tfs = out["smpl_tfs"]
    for K in (1, 3, 7):
        d.K = K
        w_want, out_want = O.query_weights_k(pts, so.verts_c, so.weights, K)
        w = d.query_weights(pts.cuda())
        assert w.shape == (1, 3000, 24)
        e = (w[0].cpu() - w_want).abs().max().item()
        xf = d.forward_skinning(pts.cuda()[None], None, tfs.cuda()[None])[0].cpu()
        xf_want = O.skinning(pts, w_want, tfs, inverse=False)
        xi, outl = d.forward(pts.cuda(), tfs.cuda()[None], return_weights=False, inverse=True, smpl_verts=server.verts_c)
        xi_want = O.skinning(pts, w_want, tfs, inverse=True)
        print(f"[parity] K={K}: weights {e:.2e} fwd skinning {(xf - xf_want).abs().max().item():.2e} "
              f"inverse {(xi.cpu() - xi_want).abs().max().item():.2e}")
        assert e < 1e-5 and (xf - xf_want).abs().max() < 1e-5 and (xi.cpu() - xi_want).abs().max() < 1e-5
        assert torch.equal(outl.cpu(), out_want)


def test_hull_box_cull_hit_sets_match_the_published_algorithm():
    """The cull box of TRAINING mode (obb_mode 'auto' -> 'hull'; no outlier override there, the hit set decides which samples
    exist: multiply.py:142-143, 208-214, 256-266) is the minimum-volume oriented box the reference asks trimesh for: hull on
    the device (round 4: mp_obb_hull_device; rounds 2-3: Qhull on the host), candidate search on the device.  On 20 random poses the hit sets equal the brute-force
    restatement's (oracle/obb_oracle.py: float64, explicit 2-D hulls) on the oracle's posed vertices -- except rays that graze
    the box within 1e-4 -- and the device's box has the volume of the host statement's (multiply_amd/obb.py)."""
    import torch
    from multiply_amd.obb import min_volume_obb
    from oracle import multiply_oracle as O
    from oracle.obb_oracle import min_volume_obb_bruteforce, rays_hitting_box
    from tests.test_render_gpu import build
    model, oracle, inp = build(H=40, W=40)
    assert model.obb_mode == "auto" and model._obb_mode_now() == "pca"      # eval: the device-only box (identical pixels)
    model.train()
    assert model._obb_mode_now() == "hull"
    dirs, cam = O.get_camera_rays(inp["uv"][0], inp["pose"][0], inp["intrinsics"][0])
    g = torch.Generator().manual_seed(123)
    n_graze = n_rays = 0
    for trial in range(20):
        cur = dict(inp)
        if trial:
            cur["smpl_pose"] = inp["smpl_pose"] + 0.35 * torch.randn(inp["smpl_pose"].shape, generator=g)
            cur["smpl_trans"] = inp["smpl_trans"] + 0.05 * torch.randn(inp["smpl_trans"].shape, generator=g)
            sp = inp["smpl_params"].clone()
            sp[:, :, 4:76], sp[:, :, 1:4] = cur["smpl_pose"], cur["smpl_trans"]
            cur["smpl_params"] = sp
        gin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in cur.items()}
        cx = model._setup(gin, -1, False)
        torch.cuda.synchronize()
        # round 4: the hull itself is built on the device (gift wrapping, csrc/geom.hip k_hull_wrap) -- no host copy of the vertices;
        # the host-side hull (Qhull) stays as the fall-back and as the cross-check here: same box volume, same hit sets
        hs = cx["hull_status"].cpu().numpy()
        assert hs.shape == (len(cx["persons"]), 8) and (hs[:, 3] == 0).all(), hs
        assert (hs[:, 1] == 2 * hs[:, 0] - 4).all() and (2 * hs[:, 2] == 3 * hs[:, 1]).all(), hs     # Euler: F = 2 H - 4, E = 3 F / 2
        if trial < 5:
            cx_h = model._setup(gin, -1, False, host_hull=True)
            torch.cuda.synchronize()
            assert cx_h["hull_status"] is None
            for n, p in enumerate(cx["persons"]):
                bd, bh = cx["per"][p]["obb"].cpu().numpy(), cx_h["per"][p]["obb"].cpu().numpy()
                assert abs(np.prod(bd[12:15]) / np.prod(bh[12:15]) - 1.0) < 1e-5, (trial, p, bd, bh)
                a = set(cx["per"][p]["hit_index"][:cx["n_hit"][n]].tolist())
                b = set(cx_h["per"][p]["hit_index"][:cx_h["n_hit"][n]].tolist())
                assert len(a.symmetric_difference(b)) <= 2, (trial, p, len(a), len(b))
            if trial == 0:
                print(f"[parity] device hulls: {hs[:, :3].tolist()} (vertices, facets, edges); rounds, pivot / insert / total clocks x16: {hs[:, 4:].tolist()}")
        if trial == 0:
            model.obb_mode = "pca"
            cx_pca = model._setup(gin, -1, False)
            model.obb_mode = "auto"
        sp = cur["smpl_params"]
        for n, p in enumerate(cx["persons"]):
            got = set(cx["per"][p]["hit_index"][:cx["n_hit"][n]].tolist())
            so = oracle.servers[p].forward(sp[0, p, 0], cur["smpl_trans"][0, p], cur["smpl_pose"][0, p], cur["smpl_shape"][0, p])
            verts = so["smpl_verts"].reshape(-1, 3).numpy()
            c, a, h, vol = min_volume_obb_bruteforce(verts)
            box = cx["per"][p]["obb"].cpu().numpy()
            assert abs(np.prod(box[12:15]) / 1.2 ** 3 * 8 - vol) < 1e-4 * vol, (trial, p)        # the same minimal volume (fp32 vertices)
            hc, ha, hh = min_volume_obb(verts)
            assert abs(np.prod(hh) * 8 - vol) < 1e-9 * vol                                         # host statement = brute force
            want, margin = rays_hitting_box(cam.numpy(), dirs.numpy(), c, a, 1.2 * h)
            diff = got.symmetric_difference(set(want.tolist()))
            assert all(abs(margin[r]) < 1e-4 for r in diff), (trial, p, len(diff), [float(margin[r]) for r in list(diff)[:5]])
            n_graze += len(diff)
            n_rays += len(got)
            if trial == 0:
                pca = set(cx_pca["per"][p]["hit_index"][:cx_pca["n_hit"][n]].tolist())
                print(f"[parity] person {p}: hull box {len(got)} rays (oracle {len(want)}, {len(diff)} grazing), PCA box {len(pca)} rays")
                assert 0 < len(got) < len(pca)    # the minimum-volume box is the tighter of the two (neither contains the other)
    print(f"[parity] 20 poses x 2 persons: {n_rays} hit rays, {n_graze} grazing differences (|margin| < 1e-4)")


def test_device_hull_give_up_path_falls_back_to_the_host_hull():
    """The device hull's workgroups of a body spin on each other and ABANDON the wrap when they are not all resident (another
    process, or the main stream, holding the CUs): status[3] = 1 and the setup repeats with the host-side hull, with a warning.
    The hook mp_debug_hull_abandon starts the wraps in that state: same box, same hit sets as the undisturbed device hull."""
    import warnings
    import torch
    from multiply_amd import hip
    from tests.test_render_gpu import build
    model, oracle, inp = build(H=24, W=24)
    model.train()
    gin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    cx = model._setup(gin, -1, False)
    assert cx["hull_status"] is not None and (cx["hull_status"][:, 3] == 0).all()
    L = hip.lib()
    assert L.mp_debug_hull_abandon(1) == 0
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            cx_f = model._setup(gin, -1, False)
        torch.cuda.synchronize()
    finally:
        assert L.mp_debug_hull_abandon(0) == 1
    assert any("falling back to the host-side hull" in str(x.message) for x in w), [str(x.message) for x in w]
    assert cx_f["hull_status"] is None                      # the repeated setup took the host hull
    for n, p in enumerate(cx["persons"]):
        bd, bh = cx["per"][p]["obb"].cpu().numpy(), cx_f["per"][p]["obb"].cpu().numpy()
        assert abs(np.prod(bd[12:15]) / np.prod(bh[12:15]) - 1.0) < 1e-5, (p, bd, bh)
        a = set(cx["per"][p]["hit_index"][:cx["n_hit"][n]].tolist())
        b = set(cx_f["per"][p]["hit_index"][:cx_f["n_hit"][n]].tolist())
        assert len(a.symmetric_difference(b)) <= 2, (p, len(a), len(b))
    cx2 = model._setup(gin, -1, False)                      # hook off again: the device hull is back
    assert cx2["hull_status"] is not None and (cx2["hull_status"][:, 3] == 0).all()
