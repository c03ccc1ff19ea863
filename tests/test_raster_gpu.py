"""The HIP z-buffer (csrc/raster.hip through multiply_amd.render.Renderer) against the float64 restatement, the mesh-space
losses against their restatement, gradients against finite differences, and the rasterised canonical surface against the
volume renderer's own silhouette."""
import time

import numpy as np
import pytest
import torch

from oracle import raster_oracle as RO
from tests.test_raster_cpu import uv_sphere

pytestmark = pytest.mark.gpu


def make_renderer(K, R, T, H, W):
    from multiply_amd.render import Renderer
    r = Renderer(img_size=[H, W], cam_intrinsic=K)
    r.set_camera(torch.tensor(R)[None].float(), torch.tensor(T)[None].float())
    return r


def stable_pixels(v, f, R, T, K, H, W, delta=2e-3):
    """pixels whose winning face does not change when the camera's principal point moves by +-delta pixels: away from
    edges and depth ties, where float32 and float64 must agree on the face"""
    base = RO.rasterize(v, f, R, T, K[0, 0], K[1, 1], K[0, 2], K[1, 2], H, W)
    ok = np.ones((H, W), bool)
    for dx, dy in ((delta, 0), (-delta, 0), (0, delta), (0, -delta)):
        p = RO.rasterize(v, f, R, T, K[0, 0], K[1, 1], K[0, 2] + dx, K[1, 2] + dy, H, W)[1]
        ok &= p == base[1]
    return base, ok


def test_zbuffer_matches_the_restatement_on_random_triangle_soup():
    rs = np.random.RandomState(3)
    H, W = 40, 56
    K = np.array([[70.0, 0, 27.3], [0, 64.0, 19.1], [0, 0, 1.0]])
    ang = 0.2
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    T = np.array([0.05, -0.02, 3.0])
    ctr = rs.uniform(-1.4, 1.4, (600, 1, 3)) * [1, 1, 0.8]
    size = np.where(rs.uniform(size=(600, 1, 1)) < 0.05, 1.5, 0.12)                 # a few faces with big pixel boxes
    v = (ctr + rs.normal(size=(600, 3, 3)) * size).reshape(-1, 3)
    v[:9, 2] -= 6.0                                                                  # behind / straddling the camera
    v[9:12] = v[9]                                                                   # degenerate
    f = np.arange(1800).reshape(600, 3)
    (zb, p2f, bary), ok = stable_pixels(v, f, R, T, K, H, W)
    frag = make_renderer(K, R, T, H, W).rasterize(torch.tensor(v).float().cuda(), torch.tensor(f).cuda())
    torch.cuda.synchronize()
    gz, gf, gb = frag.zbuf[0, :, :, 0].cpu().numpy(), frag.pix_to_face[0, :, :, 0].cpu().numpy(), frag.bary_coords[0, :, :, 0].cpu().numpy()
    assert ok.mean() > 0.97 and (p2f >= 0).mean() > 0.5
    assert (gf[ok] == p2f[ok]).all()
    hit = ok & (p2f >= 0)
    ez = np.abs(gz[hit] - zb[hit]).max()
    eb = np.abs(gb[hit] - bary[hit]).max()
    print(f"[parity] z-buffer vs float64 restatement: {hit.sum()} stable covered pixels, max |dz| {ez:.2e}, max |dbary| {eb:.2e}; "
          f"{(~ok).sum()} edge pixels, face differs on {(gf[~ok] != p2f[~ok]).sum()}")
    assert ez < 2e-5 and eb < 2e-4                                                   # float32 projection of coordinates ~50 px
    assert (gz[gf < 0] == -1).all() and (gb[gf < 0] == -1).all()
    # on the unstable (edge / tie) pixels the depth still is that of SOME face through the pixel, close to the restatement's
    # unless the pixel sits on a silhouette edge
    assert (gf[~ok] != p2f[~ok]).mean() < 0.5


def test_zbuffer_of_a_sphere_at_image_scale_and_its_speed():
    H, W = 940, 1280                                                                 # Hi4D frame size
    # off-axis on purpose: with the sphere on the optical axis the 45-degree meridians pass exactly through pixel centres,
    # which the strict w > 0 rule assigns to neither neighbour
    K = np.array([[1400.0, 0, 640.3], [0, 1400.0, 469.8], [0, 0, 1.0]])
    R, T = np.diag([1.0, -1.0, -1.0]), np.array([0.0, 0.0, 3.0])
    ctr = np.array([0.0137, -0.0071, 0.0])
    v, f = uv_sphere(ctr, 0.5, 256, 512)                                             # 260 k faces, ~1.5 px each
    r = make_renderer(K, R, T, H, W)
    vt, ft = torch.tensor(v).float().cuda(), torch.tensor(f).cuda()
    frag = r.rasterize(vt, ft)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        frag = r.rasterize(vt, ft)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    z = frag.zbuf[0, :, :, 0].cpu().numpy()
    py, px = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
    d = np.stack([(px - 640.3) / 1400, (py - 469.8) / 1400, np.ones_like(px)], -1)
    cc = R @ ctr + T
    bq, aq = d @ cc, (d * d).sum(-1)
    disc = bq * bq - aq * (cc @ cc - 0.25)
    t = (bq - np.sqrt(np.maximum(disc, 0))) / aq
    inner = disc > 0.01 * aq * 0.25
    assert (z[inner] > 0).all() and (z[disc < 0] == -1).all()
    err = np.abs(z[inner] - t[inner]).max()
    print(f"[perf] z-buffer 940x1280, {f.shape[0]} faces: {ms:.3f} ms per mesh (host-timed, 10 launches); "
          f"max |depth - analytic| {err:.2e} on {inner.sum()} pixels")
    assert err < 2e-4                                                                # sagitta r (1 - cos(pi/512)) = 1e-5, slanted
    # holes: inside the silhouette every pixel is covered (strict w > 0 only loses pixels exactly on an edge)
    assert (z[disc > 0.001 * aq * 0.25] > 0).mean() > 0.99999


def test_depth_maps_back_propagate_into_the_vertices():
    H, W = 24, 32
    K = np.array([[40.0, 0, 15.7], [0, 40.0, 12.2], [0, 0, 1.0]])
    R, T = np.eye(3), np.array([0.0, 0.0, 2.5])
    v, f = uv_sphere([0.05, 0.0, 0.0], 0.6, 6, 8)
    r = make_renderer(K, R, T, H, W)
    vt = torch.tensor(v).float().cuda().requires_grad_(True)
    ft = torch.tensor(f).cuda()
    wgt = torch.rand(H, W, generator=torch.Generator().manual_seed(0)).cuda()
    d = r.render_multiple_depth_map([vt[None]], [ft[None]])[0]
    assert d.shape == (1, H, W, 1) and d.requires_grad
    plain = r.rasterize(vt, ft).zbuf
    assert torch.equal(d.detach(), plain)                                            # forward value = the kernel's
    hit = (d[0, :, :, 0] > 0)
    (d[0, :, :, 0] * wgt * hit).sum().backward()
    g = vt.grad.cpu().double().numpy()
    # finite differences of the float64 restatement with visibility frozen to the same faces (pytorch3d's gradient also
    # treats pix_to_face as a constant)
    zb0, p2f0, _ = RO.rasterize(v, f, R, T, 40.0, 40.0, 15.7, 12.2, H, W)
    assert (p2f0 == r.rasterize(vt, ft).pix_to_face[0, :, :, 0].cpu().numpy()).mean() > 0.99
    w = wgt.cpu().double().numpy()

    def loss(vv):
        s = RO.project(vv, R, T, 40.0, 40.0, 15.7, 12.2)
        py, px = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
        tot = 0.0
        for (i, j) in zip(*np.nonzero(p2f0 >= 0)):
            a, b, c = s[f[p2f0[i, j]]]
            e = lambda p, q: (px[i, j] - p[0]) * (q[1] - p[1]) - (py[i, j] - p[1]) * (q[0] - p[0])
            w0, w1, w2 = e(b, c), e(c, a), e(a, b)
            tot += w[i, j] / (w0 / a[2] + w1 / b[2] + w2 / c[2]) * (w0 + w1 + w2)
        return tot
    num = np.zeros_like(v)
    for i in range(0, v.shape[0], 3):
        for k in range(3):
            dv = np.zeros_like(v); dv[i, k] = 1e-6
            num[i, k] = (loss(v + dv) - loss(v - dv)) / 2e-6
    sel = np.arange(0, v.shape[0], 3)
    err = np.abs(g[sel] - num[sel]).max() / np.abs(num[sel]).max()
    print(f"[parity] d depth / d vertices vs finite differences: rel err {err:.2e}")
    assert err < 2e-3


def test_mesh_space_losses_on_the_model_match_the_restatement():
    """get_depth_order_loss / frame_instance_masks end to end on the synthetic two-person scene: canonical meshes -> posed ->
    z-buffers -> losses; the loss arithmetic against oracle/raster_oracle.py on the same depth maps, the pose gradient
    against a finite difference of the whole chain."""
    from multiply_amd import mesh_losses as ML
    from multiply_amd.mesh import canonical_mesh
    from tests.test_render_gpu import build
    H, W = 60, 80
    model, _, inp = build(H=H, W=W)
    gin = {k: (t.cuda() if torch.is_tensor(t) else t) for k, t in inp.items()}
    Kp = gin["intrinsics"][0].double().clone()
    Kp[0, 2] += 0.5; Kp[1, 2] += 0.5                      # the volume renderer shoots rays through integer (x, y)
    gin["P"] = (Kp @ torch.linalg.inv(gin["pose"][0].double()))[None].float()
    gin["img_size"] = (H, W)
    rs = np.random.RandomState(0)
    sam = torch.tensor(rs.normal(0, 4.0, (1, H, W, 2))).float().cuda()
    gin["org_sam_mask"] = sam
    meshes = [canonical_mesh(model, p, cond=gin["smpl_pose"][0, p, 3:] / np.pi, res_up=1) for p in range(2)]
    masks, depth, kps = ML.frame_instance_masks(model, gin, use_smpl_mesh=True)
    assert masks.shape == (2, H, W) and kps.shape == (2, 27, 2) and kps.dtype == torch.int32
    assert not bool((masks[0] & masks[1]).any()) and bool(masks[0].any()) and bool(masks[1].any())
    # SMPL z-buffers against the restatement (6890-vertex body, float64)
    r = ML.get_renderer(gin)
    cam = (r.cam_R[0].numpy().astype(np.float64), r.cam_T[0].numpy().astype(np.float64))
    fx, fy = r.focal_length[0].tolist(); cx, cy = r.principal_point[0].tolist()
    vs, fs, outs = ML.posed_meshes(model, gin, use_smpl_mesh=True)
    want = [RO.rasterize(v[0].cpu().numpy(), f[0].cpu().numpy(), *cam, fx, fy, cx, cy, H, W) for v, f in zip(vs, fs)]
    for p in range(2):
        same = (want[p][1] >= 0) == (depth[p].cpu().numpy() > 0)
        dz = np.abs(depth[p].cpu().numpy() - want[p][0])[same & (want[p][1] >= 0)]
        print(f"[parity] SMPL z-buffer person {p}: coverage differs on {(~same).sum()} of {H * W} pixels, "
              f"median |dz| {np.median(dz):.1e}, 99.5 % {np.quantile(dz, 0.995):.1e}")
        assert (~same).sum() <= 3 and np.quantile(dz, 0.995) < 1e-5
    _, _, wm = RO.front_depth_and_masks([d.cpu().numpy() for d in depth])
    assert (wm == masks.cpu().numpy()).all()
    # key points: the projected joints (truncated to ints as astype(np.int32) does)
    j = outs[0]["smpl_all_jnts"][0, :27].double().cpu().numpy()
    t = np.concatenate([j, np.ones((27, 1))], 1) @ gin["P"][0].double().cpu().numpy().T
    assert (kps[0].cpu().numpy() == (t[:, :2] / t[:, 2:3]).astype(np.int32)).all()

    # depth-order loss on the posed canonical meshes, with gradients into the pose
    pose = gin["smpl_pose"].clone().requires_grad_(True)
    trans = gin["smpl_trans"].clone().requires_grad_(True)
    tin = dict(gin, smpl_pose=pose, smpl_trans=trans)
    draws = [torch.arange(0, m["vertices"].shape[0], 7)[:5120] for m in meshes]
    opt = {"depth_order_weight": 0.1, "interpenetration_loss_weight": 0.005}
    order, sil, inter = ML.get_depth_order_loss(model, tin, 100, opt, meshes=meshes, draws=draws)
    (order + inter.sum()).backward()
    torch.cuda.synchronize()
    with torch.no_grad():
        vs, fs, _ = ML.posed_meshes(model, gin, meshes=meshes)
        dm = [d[0, :, :, 0].cpu().numpy() for d in r.render_multiple_depth_map(vs, fs)]
    want_order = RO.depth_order_loss(dm, sam[0].cpu().numpy(), 100, 0.1)
    print(f"[parity] depth-order loss {float(order):.6f} vs restatement {want_order:.6f}; interpenetration {float(inter):.3e}; "
          f"|d/d pose| {float(pose.grad.abs().max()):.2e}, |d/d trans| {float(trans.grad.abs().max()):.2e}")
    assert abs(float(order) - want_order) < 1e-5 * max(1.0, want_order) and want_order > 0
    assert float(sil) == 0.0 and torch.isfinite(pose.grad).all() and float(trans.grad.abs().max()) > 0
    # finite difference along the camera axis: moving person 0 in z changes its depths one for one where it is the labelled
    # but hidden surface (d loss / d z = sigmoid(gt - front)) and oppositely where it is the wrongly-front one
    eps = 1e-3

    def loss_at(dz):
        with torch.no_grad():
            t2 = gin["smpl_trans"].clone(); t2[0, 0, 2] += dz
            o, _, _ = ML.get_depth_order_loss(model, dict(gin, smpl_trans=t2), 100, {"depth_order_weight": 0.1}, meshes=meshes)
            return float(o)
    t3 = gin["smpl_trans"].clone().requires_grad_(True)
    o3, _, _ = ML.get_depth_order_loss(model, dict(gin, smpl_trans=t3), 100, {"depth_order_weight": 0.1}, meshes=meshes)
    o3.backward()
    fd = (loss_at(eps) - loss_at(-eps)) / (2 * eps)
    print(f"[parity] d depth-order / d trans_z(person 0): autograd {float(t3.grad[0, 0, 2]):.5f}, finite difference {fd:.5f}")
    assert abs(float(t3.grad[0, 0, 2]) - fd) < 0.05 * abs(fd) + 1e-3        # visibility changes at silhouettes are not differentiated

    # the rasterised canonical surfaces against the volume renderer's silhouette of the same frame
    # (with a sharp density: at the initial beta = 0.1 a ray that passes 0.2 outside the surface still accumulates acc = 0.5)
    with torch.no_grad():
        model.density.beta.fill_(0.01)
    out = model(gin)
    acc = out["acc_map"].reshape(H, W) > 0.5
    m2, _, _ = ML.frame_instance_masks(model, gin, use_smpl_mesh=False, res_up=1)
    cover = m2.any(0)
    iou = float((acc & cover).sum()) / float((acc | cover).sum())
    inside = float((acc & cover).sum()) / float(acc.sum())
    print(f"[parity] volume-rendered acc_map > 0.5 vs mesh z-buffer silhouette: {inside:.3f} of it inside, IoU {iou:.3f} "
          f"({int(cover.sum())} mesh vs {int(acc.sum())} volume pixels)")
    # the random-initialised surface is a ~0.6 sphere: the mesh keeps all of it inside the canonical box, the volume renderer
    # only the part within 0.1 of the posed body (deformer outliers are empty space) -> containment, not equality
    assert inside > 0.95 and iou > 0.6


def test_depth_refinement_stage_on_a_written_sequence(tmp_path):
    """opt_depth's per-frame body (multiply_model.py:230-486) end to end: sequence on disk -> resident frames -> BodyModelParams
    rows -> canonical meshes -> it_per_loop Adam steps on the frame's translations over render + depth-order +
    interpenetration losses.  The SAM labels name person 1 everywhere, so wherever person 0's mesh is in front of person 1's
    the order is 'wrong' and the stage has to push the two apart in depth."""
    import warnings
    warnings.filterwarnings("ignore")
    from multiply_amd import mesh_losses as ML
    from multiply_amd.body_model_params import BodyModelParams
    from multiply_amd.config import load_config, to_config
    from multiply_amd.datasets import Hi4DDataset, Hi4DTestDataset, draw_positions
    from multiply_amd.loss import Loss
    from multiply_amd.multiply import Multiply
    from multiply_amd.synthetic import make_smpl_tables, write_sequence
    root = str(tmp_path / "seq")
    H = W = 64
    w = write_sequence(root, n_frames=2, H=H, W=W)
    import os
    dopt = to_config(dict(data_root=os.path.dirname(root), data_dir=os.path.basename(root), start_frame=0, end_frame=2, num_sample=0,
                          using_SAM=False, pixel_per_batch=512))
    test = Hi4DTestDataset(dopt)
    store = test.dataset.store
    opt = load_config()
    torch.manual_seed(0)
    model = Multiply(opt, w["shape"], smpl_tables=make_smpl_tables(0))
    bml = torch.nn.ModuleList()
    for p in range(2):                                                    # multiply_model.py:44-52, 82-92
        bm = BodyModelParams(2, model_type="smpl").cuda()
        bm.init_parameters("betas", torch.tensor(w["shape"][p:p + 1]).float().cuda())
        bm.init_parameters("global_orient", torch.tensor(w["poses"][:, p, :3]).float().cuda())
        bm.init_parameters("body_pose", torch.tensor(w["poses"][:, p, 3:]).float().cuda())
        bm.init_parameters("transl", torch.tensor(w["trans"][:, p]).float().cuda())
        bml.append(bm)
    item, images, _, _, idx = test[0]
    inputs = {k: (torch.as_tensor(v)[None].cuda() if not isinstance(v, int) else torch.tensor([v]).cuda()) for k, v in item.items()
              if k in ("P", "C", "intrinsics", "pose", "smpl_params", "idx")}
    inputs["P"] = inputs["P"].float()
    inputs["img_size"] = (H, W)
    sam = torch.zeros(1, H, W, 2).cuda()
    sam[..., 0], sam[..., 1] = -8.0, 8.0
    inputs["org_sam_mask"] = sam
    rng = np.random.RandomState(0)

    def sample_fn():                                                       # the frame's weighted_sampling from the resident bytes
        pos, outside = draw_positions(store.bbox[0, 0], store.bbox[0, 1], (H, W), 96, rng)
        rgb, uv, _, sm = store.sample(0, pos, sam[0])
        return dict(uv=uv[None], index_outside=torch.from_numpy(outside)[None], sam_mask=sm[None]), dict(rgb=rgb[None])
    t0 = [bm.transl.weight.detach().clone() for bm in bml]
    hist = ML.opt_depth_frame(model, bml, Loss(opt.loss), inputs, sample_fn, epoch=100, it_per_loop=4, lr=5e-3,
                              loss_opt={"depth_order_weight": 0.1, "interpenetration_loss_weight": 0.005}, res_up=1)
    torch.cuda.synchronize()
    order = [float(h["depth_order_loss"]) for h in hist]
    print("[info] depth refinement: depth-order loss per iteration", [f"{o:.4f}" for o in order],
          "render", [f"{float(h['render_loss']):.4f}" for h in hist], "interpenetration", [f"{float(h['interpenetration_loss']):.2e}" for h in hist])
    assert len(hist) == 4 and all(np.isfinite(o) for o in order) and order[0] > 0 and order[-1] < order[0]
    for p in range(2):
        moved = (bml[p].transl.weight.detach() - t0[p]).abs()
        assert float(moved[0].max()) > 1e-3 and float(moved[1].max()) == 0.0           # the frame's row only
        assert not bml[p].transl.weight.requires_grad and bml[p].body_pose.weight.grad is None
    dz = [float(bml[p].transl.weight[0, 2] - t0[p][0, 2]) for p in range(2)]
    print(f"[info] translation change along the camera axis: person 0 {dz[0]:+.4f}, person 1 {dz[1]:+.4f}")
    assert dz[0] > 0 > dz[1]                                                            # person 0 back, person 1 forward


def test_vertex_colour_renders_of_a_scene():
    """render_multiple_meshes / render_mesh_recon (render.py:107-119, 161-208): two coloured spheres joined as one scene ->
    RGBA over white = the SoftPhong blend of the (up to 10) faces covering each pixel, against the float64 restatement; the
    nearer sphere hides the farther one; normal / shaded views stack along the height."""
    H, W = 48, 64
    K = np.array([[60.0, 0, 31.7], [0, 60.0, 24.2], [0, 0, 1.0]])
    r = make_renderer(K, np.eye(3), np.array([0.0, 0.0, 3.0]), H, W)
    va, fa = uv_sphere([-0.15, 0.0, 0.0], 0.5, 24, 48)
    vb, fb = uv_sphere([0.35, 0.05, 0.8], 0.5, 24, 48)
    t = lambda a: torch.tensor(a).float().cuda()
    red, blue = torch.tensor([1.0, 0, 0]).cuda().expand(va.shape[0], 3), torch.tensor([0, 0, 1.0]).cuda().expand(vb.shape[0], 3)
    img = r.render_multiple_meshes([t(va)[None], t(vb)[None]], [torch.tensor(fa).cuda()[None], torch.tensor(fb).cuda()[None]],
                                   [red[None], blue[None]])
    torch.cuda.synchronize()
    assert img.shape == (1, H, W, 4)
    za = r.rasterize(t(va), torch.tensor(fa).cuda()).zbuf[0, :, :, 0]
    zb = r.rasterize(t(vb), torch.tensor(fb).cuda()).zbuf[0, :, :, 0]
    is_a = (za > 0) & ((zb < 0) | (za < zb))
    is_b = (zb > 0) & ~is_a
    empty = (za < 0) & (zb < 0)
    assert bool(is_a.any()) and bool(is_b.any()) and bool(((za > 0) & (zb > 0)).any())       # they overlap in the image
    # colours: the nearest sphere's (the other one is >= 0.3 depth units behind: weight exp(-30)); alpha = 1 - prod(1 - prob) of
    # the covering faces, each prob >= 1/2, two or more of them (front and back of a sphere)
    assert torch.allclose(img[0][is_a][:, :3], torch.tensor([1.0, 0, 0]).cuda(), atol=1e-5)
    assert torch.allclose(img[0][is_b][:, :3], torch.tensor([0, 0, 1.0]).cuda(), atol=1e-5)
    assert float(img[0][is_a | is_b][:, 3].min()) >= 0.75 - 1e-6 and float(img[0][is_a | is_b][:, 3].max()) <= 1.0
    assert torch.equal(img[0][empty], torch.tensor([1.0, 1, 1, 0]).cuda().expand(int(empty.sum()), 4))
    verts, faces = np.concatenate([va, vb]), np.concatenate([fa, fb + len(va)])
    cols = np.concatenate([np.tile([[1.0, 0, 0]], (len(va), 1)), np.tile([[0, 0, 1.0]], (len(vb), 1))])
    want = RO.soft_render(verts, faces, cols, np.eye(3), np.array([0.0, 0.0, 3.0]), 60.0, 60.0, 31.7, 24.2, H, W, sigma=1e-4,
                          gamma=1e-4, K=10, blur=0.0)
    err = np.abs(img[0].cpu().numpy() - want)
    print(f"[parity] SoftPhong image of the hard rasteriser vs restatement: alpha max {err[..., 3].max():.2e}, rgb max {err[..., :3].max():.2e}")
    assert np.quantile(err[..., 3], 0.99) < 2e-4 and err[..., :3].max() < 5e-3 and (err[..., 3] > 1e-2).mean() < 0.01
    views = r.render_mesh_recon(t(va)[None], torch.tensor(fa).cuda()[None], colors=red[None], mode="npa")
    assert views.shape == (1, 3 * H, W, 4)
    nrm = views[0, H:2 * H][za > 0][:, [2, 1, 0]] * 2 - 1         # stack = [phong | normal | albedo]; normals stored as (z, y, x) * 0.5 + 0.5
    assert float((nrm.norm(dim=1) - 1).abs().max()) < 0.05         # interpolated unit vertex normals
    centre = views[0, H:2 * H][int(24.2), int(31.7 - 0.15 * 60 / 2.5)]
    assert float(centre[0]) < 0.1                                  # facing the camera: normal z = -1 -> channel 0 (z) ~ 0


def _soft_scene(rs, n_soup=250):
    """two tessellated spheres (dense small faces, front and back layers) + a triangle soup with a few big faces"""
    v1, f1 = uv_sphere([-0.25, 0.05, 3.0], 0.55, n_lat=14, n_lon=24)
    v2, f2 = uv_sphere([0.35, -0.1, 3.5], 0.6, n_lat=12, n_lon=20)
    ctr = rs.uniform(-1.2, 1.2, (n_soup, 1, 3)) * [1, 0.8, 0.5] + [0, 0, 3.2]
    size = np.where(rs.uniform(size=(n_soup, 1, 1)) < 0.05, 0.8, 0.08)
    vs = (ctr + rs.normal(size=(n_soup, 3, 3)) * size).reshape(-1, 3)
    vs[:6, 2] -= 7.0                                                               # behind the camera
    fs = np.arange(3 * n_soup).reshape(n_soup, 3)
    verts = np.concatenate([v1, v2, vs])
    faces = np.concatenate([f1, f2 + len(v1), fs + len(v1) + len(v2)])
    cols = np.concatenate([np.tile([[1.0, 0, 0]], (len(v1), 1)), np.tile([[0, 1.0, 0]], (len(v2), 1)), rs.uniform(0, 1, (len(vs), 3))])
    return verts, faces, cols


@pytest.mark.parametrize("H,W,K", [(48, 64, 100), (60, 44, 100), (48, 64, 5)])
def test_soft_render_matches_the_restatement(H, W, K):
    """mp_raster_soft_bins + mp_raster_soft against oracle/raster_oracle.soft_render (float64): the silhouette channel, the
    blended colours and the per-pixel face selection, wide / tall images, and a small K that forces evictions."""
    import ctypes as C
    from multiply_amd import hip, render
    rs = np.random.RandomState(5)
    verts, faces, cols = _soft_scene(rs)
    Kc = np.array([[75.0, 0, W / 2 - 0.7], [0, 70.0, H / 2 + 0.4], [0, 0, 1.0]])
    ang = 0.15
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    T = np.array([0.03, -0.02, 0.2])
    r = make_renderer(Kc, R, T, H, W)
    render_K = render.SOFT_K
    render.SOFT_K = K
    try:
        img, sel = r.soft_rasterize(torch.tensor(verts).float(), torch.tensor(faces), torch.tensor(cols).float(), want_sel=True)
    finally:
        render.SOFT_K = render_K
    torch.cuda.synchronize()
    img, sel = img.cpu().numpy().astype(np.float64), np.sort(sel.cpu().numpy(), axis=-1)
    blur = render.SOFT_BLUR
    idx, zbuf, dists, bary = RO.soft_fragments(verts, faces, R, T, Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2], H, W, blur, K)
    want = RO.soft_render(verts, faces, cols, R, T, Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2], H, W, K=K)
    same_set = (np.sort(idx, axis=-1) == sel).all(-1)
    n_full = int((idx[..., K - 1] >= 0).sum())
    ea, ec = np.abs(img[..., 3] - want[..., 3]), np.abs(img[..., :3] - want[..., :3]).max(-1)
    print(f"[parity] soft render {H}x{W} K={K}: {int((idx[..., 0] >= 0).sum())} blended pixels ({n_full} with full lists), selection "
          f"differs on {(~same_set).sum()}; alpha max {ea.max():.2e} mean {ea.mean():.2e}; rgb max {ec.max():.2e} mean {ec.mean():.2e} "
          f"(pixels with the same selection: alpha {ea[same_set].max():.2e}, rgb {ec[same_set].max():.2e})")
    assert (idx[..., 0] >= 0).mean() > 0.3 and (K == 100 or n_full > 100)
    # float32 vs float64 may flip the selection when the list is full: a pixel outside two faces that share an edge gets the SAME
    # clipped depth from both (an exact tie in float64, broken by face id), and which of them is evicted changes the product
    assert (~same_set).mean() < 0.01 and (K < 100 or same_set.all())
    assert ea[same_set].max() < 2e-4 and ea[same_set].mean() < 2e-5
    # weights are exp(zinv / 1e-4) of a float32 zinv ~ 0.97: 8e-4 relative per weight
    assert ec[same_set].max() < 5e-3 and ec[same_set].mean() < 2e-4
    assert np.allclose(img[(idx[..., 0] < 0) & same_set], [1, 1, 1, 0], atol=0)                 # untouched pixels: white, transparent


def test_soft_render_back_propagates_into_the_vertices_and_feeds_the_silhouette_loss():
    """Renderer.softrender_multiple_meshes: forward = the kernel's image whether or not a gradient is asked for; the gradient
    = autograd through the torch blend of the kernel's selection, checked against the float64 CPU evaluation of the same
    selection; and get_depth_order_loss's silhouette term against the restatement's image."""
    from multiply_amd import render
    rs = np.random.RandomState(6)
    H, W = 40, 52
    v1, f1 = uv_sphere([-0.2, 0.0, 3.0], 0.5, n_lat=10, n_lon=16)
    v2, f2 = uv_sphere([0.3, 0.05, 3.4], 0.55, n_lat=9, n_lon=14)
    Kc = np.array([[60.0, 0, 25.3], [0, 58.0, 19.6], [0, 0, 1.0]])
    R, T = np.eye(3), np.array([0.0, 0.0, 0.1])
    r = make_renderer(Kc, R, T, H, W)
    tv = [torch.tensor(v1).float().cuda()[None].requires_grad_(True), torch.tensor(v2).float().cuda()[None].requires_grad_(True)]
    tf = [torch.tensor(f1).cuda()[None], torch.tensor(f2).cuda()[None]]
    tc = [torch.tensor([[1.0, 0, 0]]).repeat(len(v1), 1).cuda()[None], torch.tensor([[0, 1.0, 0]]).repeat(len(v2), 1).cuda()[None]]
    with torch.no_grad():
        plain = r.softrender_multiple_meshes(tv, tf, tc)
    img = r.softrender_multiple_meshes(tv, tf, tc)
    assert img.shape == (1, H, W, 4) and torch.equal(img.detach(), plain) and img.requires_grad
    wgt = torch.tensor(rs.normal(size=(H, W, 4))).float().cuda()
    (img[0] * wgt).sum().backward()
    torch.cuda.synchronize()
    verts = np.concatenate([v1, v2]); faces = np.concatenate([f1, f2 + len(v1)])
    cols = np.concatenate([np.tile([[1.0, 0, 0]], (len(v1), 1)), np.tile([[0, 1.0, 0]], (len(v2), 1))])
    want = RO.soft_render(verts, faces, cols, R, T, Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2], H, W)
    e = np.abs(plain[0].cpu().numpy() - want)
    assert e[..., 3].max() < 2e-4 and e[..., :3].max() < 5e-3
    _, sel = r.soft_rasterize(torch.tensor(verts).float(), torch.tensor(faces), torch.tensor(cols).float(), want_sel=True)
    cv = torch.tensor(verts, dtype=torch.float64, requires_grad=True)
    act, soft = render.soft_blend_selected(cv, torch.tensor(faces), torch.tensor(cols), sel.cpu(), torch.tensor(R), torch.tensor(T),
                                           torch.tensor([Kc[0, 0], Kc[1, 1]]), torch.tensor([Kc[0, 2], Kc[1, 2]]), H, W)
    (soft * wgt.cpu().double()[act[:, 0], act[:, 1]]).sum().backward()
    got = torch.cat([tv[0].grad[0], tv[1].grad[0]]).cpu().double()
    rel = float((got - cv.grad).abs().max() / cv.grad.abs().max())
    print(f"[grad parity] d soft image / d vertices: float32 device autograd vs float64 of the same selection, rel-to-max {rel:.2e} "
          f"(|grad| max {float(cv.grad.abs().max()):.3e})")
    assert float(cv.grad.abs().max()) > 1.0 and rel < 2e-2          # exp(zinv / 1e-4) of a float32 zinv: ~1e-3 per weight


def test_silhouette_term_of_the_depth_order_loss():
    from multiply_amd import mesh_losses as ML
    from multiply_amd.mesh import canonical_mesh
    from tests.test_render_gpu import build
    H, W = 48, 64
    model, _, inp = build(H=H, W=W)
    gin = {k: (t.cuda() if torch.is_tensor(t) else t) for k, t in inp.items()}
    Kp = gin["intrinsics"][0].double().clone()
    Kp[0, 2] += 0.5; Kp[1, 2] += 0.5
    gin["P"] = (Kp @ torch.linalg.inv(gin["pose"][0].double()))[None].float()
    gin["img_size"] = (H, W)
    rs = np.random.RandomState(1)
    gin["org_sam_mask"] = torch.tensor(rs.normal(0, 4.0, (1, H, W, 2))).float().cuda()
    meshes = [canonical_mesh(model, p, cond=gin["smpl_pose"][0, p, 3:] / np.pi, res_up=1) for p in range(2)]
    trans = gin["smpl_trans"].clone().requires_grad_(True)
    opt = {"depth_order_weight": 0.1, "silhouette_weight": 0.3}
    order, sil, inter = ML.get_depth_order_loss(model, dict(gin, smpl_trans=trans), 100, opt, meshes=meshes)
    sil.backward()
    r = ML.get_renderer(gin)
    with torch.no_grad():
        vs, fs, _ = ML.posed_meshes(model, gin, meshes=meshes)
    verts = np.concatenate([v[0].cpu().numpy() for v in vs]).astype(np.float64)
    nv = np.cumsum([0] + [v.shape[1] for v in vs])
    faces = np.concatenate([f[0].cpu().numpy() + o for f, o in zip(fs, nv[:-1])])
    cols = np.concatenate([np.tile(np.array([ML.COLOR_DICT[p]]) / 255.0, (v.shape[1], 1)) for p, v in enumerate(vs)])
    fx, fy = r.focal_length[0].tolist(); cx, cy = r.principal_point[0].tolist()
    img = RO.soft_render(verts, faces, cols, r.cam_R[0].numpy().astype(np.float64), r.cam_T[0].numpy().astype(np.float64),
                         fx, fy, cx, cy, H, W)
    gt = ML.gt_instance_map(gin["org_sam_mask"], 2).cpu().numpy().astype(np.float64)
    want = 0.3 * np.mean((gt - 255 * img[..., :3] * img[..., 3:]) ** 2) * (1 - 100 / 1000)
    print(f"[parity] silhouette loss {float(sil):.4f} vs restatement {want:.4f}; |d/d trans| {float(trans.grad.abs().max()):.3e}")
    assert abs(float(sil) - want) < 2e-3 * want and want > 10.0
    assert torch.isfinite(trans.grad).all() and float(trans.grad.abs().max()) > 0
