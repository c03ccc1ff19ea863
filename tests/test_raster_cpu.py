"""The z-buffer restatement (oracle/raster_oracle.py) against analytic cases, and the loss arithmetic on top of the depth
maps against hand-computed values.  pytorch3d (the reference's rasteriser) is absent: these anchors are what pins the
oracle that tests/test_raster_gpu.py compares the HIP kernel with."""
import numpy as np

from oracle import raster_oracle as RO

FX, FY, CX, CY, H, W = 90.0, 80.0, 31.5, 23.0, 48, 64
EYE, ZERO = np.eye(3), np.zeros(3)


def uv_sphere(c, r, n_lat=24, n_lon=48):
    th = np.linspace(0, np.pi, n_lat + 1)[1:-1]
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    ring = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.sin(th), np.sin(ph)), np.outer(np.cos(th), np.ones_like(ph))], -1)
    v = np.concatenate([[[0, 0, 1.0]], ring.reshape(-1, 3), [[0, 0, -1.0]]]) * r + np.asarray(c)
    f = []
    idx = lambda i, j: 1 + i * n_lon + (j % n_lon)
    for j in range(n_lon):
        f.append([0, idx(0, j), idx(0, j + 1)])
        f.append([len(v) - 1, idx(n_lat - 2, j + 1), idx(n_lat - 2, j)])
        for i in range(n_lat - 2):
            f.append([idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)])
            f.append([idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)])
    return v, np.asarray(f)


def pixel_dirs():
    py, px = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
    return (px - CX) / FX, (py - CY) / FY


def test_fronto_parallel_quad_has_constant_depth_and_exact_coverage():
    # quad spanning x in [-0.5, 0.7], y in [-0.4, 0.35] at z = 3: pixel centres inside its projection, nothing else
    x0, x1, y0, y1, z = -0.5, 0.7, -0.4, 0.35, 3.0
    v = np.array([[x0, y0, z], [x1, y0, z], [x1, y1, z], [x0, y1, z]])
    for faces in ([[0, 1, 2], [0, 2, 3]], [[2, 1, 0], [3, 2, 0]]):          # either winding
        zb, p2f, bary = RO.rasterize(v, faces, EYE, ZERO, FX, FY, CX, CY, H, W)
        dx, dy = pixel_dirs()
        want = (dx * z > x0) & (dx * z < x1) & (dy * z > y0) & (dy * z < y1)
        on_diag = np.abs((dx * z - x0) * (y1 - y0) - (dy * z - y0) * (x1 - x0)) < 1e-12
        assert ((p2f >= 0) == want)[~on_diag].all()
        assert np.allclose(zb[p2f >= 0], z, atol=1e-12) and (zb[p2f < 0] == -1).all()
        assert np.allclose(bary[p2f >= 0].sum(-1), 1.0, atol=1e-12)


def test_slanted_plane_depth_is_perspective_correct():
    # plane z = 3 + 0.8 x through one big triangle pair: depth along the pixel ray (dx, dy, 1) t is t = 3 / (1 - 0.8 dx)
    a, b = 3.0, 0.8
    xs, ys = np.array([-2.0, 2.0]), np.array([-2.0, 2.0])
    v = np.array([[x, y, a + b * x] for y in ys for x in xs])
    zb, p2f, _ = RO.rasterize(v, [[0, 1, 3], [0, 3, 2]], EYE, ZERO, FX, FY, CX, CY, H, W)
    dx, _ = pixel_dirs()
    assert (p2f >= 0).all()
    assert np.abs(zb - a / (1 - b * dx)).max() < 1e-10
    # screen-space-linear interpolation would be off by far more than that
    assert np.abs(zb - a / (1 - b * dx)).max() < 1e-6 * np.abs(zb - (a + b * dx * a)).max()


def test_sphere_depth_matches_the_ray_sphere_intersection_under_a_posed_camera():
    ang = 0.3
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]) @ np.diag([1.0, -1.0, -1.0])
    c_world, r = np.array([0.1, -0.05, 0.2]), 0.45
    T = np.array([0.0, 0.02, 3.0])
    v, f = uv_sphere(c_world, r, 48, 96)
    zb, p2f, _ = RO.rasterize(v, f, R, T, FX, FY, CX, CY, H, W)
    cc = R @ c_world + T
    dx, dy = pixel_dirs()
    d = np.stack([dx, dy, np.ones_like(dx)], -1)
    bq, aq = (d * cc).sum(-1), (d * d).sum(-1)
    disc = bq * bq - aq * (cc @ cc - r * r)
    t = (bq - np.sqrt(np.maximum(disc, 0))) / aq
    inner = disc > 0.02 * aq * r * r                    # away from the silhouette, where the facetted sphere is inside
    assert (p2f[inner] >= 0).all() and inner.sum() > 300
    assert np.abs(zb[inner] - t[inner]).max() < 2.5e-3  # sagitta of the 48 x 96 tessellation: r (1 - cos(pi/96)) ~ 2.4e-4 / cos
    assert (p2f[disc < 0] == -1).all()                  # inscribed mesh: never outside the analytic silhouette


def test_nearest_face_wins_and_hidden_or_degenerate_faces_are_ignored():
    near = np.array([[-1, -1, 2.0], [1, -1, 2.0], [0, 1.5, 2.0]])
    far = near * [2.5, 2.5, 1.5]
    behind = near * [1, 1, -1]
    degenerate = np.array([[0, 0, 1.0], [0.1, 0.1, 1.0], [0.2, 0.2, 1.0]])
    v = np.concatenate([far, near, behind, degenerate])
    f = [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]]
    zb, p2f, _ = RO.rasterize(v, f, EYE, ZERO, FX, FY, CX, CY, H, W)
    assert set(np.unique(p2f)) <= {-1, 0, 1} and np.allclose(zb[p2f == 1], 2.0, atol=1e-12) and (p2f == 1).sum() > 0
    assert np.allclose(zb[p2f == 0], 3.0, atol=1e-12) and (p2f == 0).sum() > 0          # the far one shows around the near one's outline


def test_front_depth_instance_masks_and_depth_order_loss():
    d0 = np.array([[2.0, -1.0, 3.0, 2.5]])
    d1 = np.array([[2.5, 4.0, -1.0, 2.0]])
    mx, front, masks = RO.front_depth_and_masks([d0, d1])
    assert front.tolist() == [[2.0, 4.0, 3.0, 2.0]] and masks.tolist() == [[[True, False, True, False]], [[False, True, False, True]]]
    big = 8.0
    # pixel 0: SAM says person 1, meshes say person 0 in front -> penalised; pixel 3: SAM person 0, meshes person 1 -> penalised;
    # pixel 1: SAM agrees; pixel 2: SAM names person 1 who is not rendered there -> invalid
    sam = np.array([[[-big, big], [-big, big], [-big, big], [big, -big]]])
    got = RO.depth_order_loss([d0, d1], sam, epoch=200, depth_order_weight=0.1)
    want = 0.1 * (1 - 200 / 1000) * (np.log(1 + np.exp(0.5)) * 2)
    assert abs(got - want) < 1e-12
    assert RO.depth_order_loss([d0, d1], sam, epoch=1500) == 0.0
    amb = np.zeros_like(sam)                                                   # sigmoid = 0.5 + 0.5: sum 1.0 is allowed ...
    assert RO.depth_order_loss([d0, d1], amb + 1.0, epoch=0) == 0.0            # ... 0.73 * 2 > 1.01 is not


def test_projection_decomposition_matches_its_construction():
    rs = np.random.RandomState(0)
    for s in (1.0, 1.7):
        K = np.array([[900.0, 0.3, 470.0], [0, 880.0, 640.0], [0, 0, 1.0]])
        q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        R = q * np.sign(np.linalg.det(q))
        c = rs.normal(size=3)
        P = K @ np.concatenate([R, (-R @ c)[:, None]], 1) @ np.diag([s, s, s, 1.0])
        K2, R2, c2 = RO.decompose_projection(P)
        assert np.allclose(K2, K * s) and np.allclose(R2, R) and np.allclose(c2, c / s)      # K keeps the scale (cv2 does not normalise)
        from multiply_amd.render import decompose_projection
        K3, R3, c3 = decompose_projection(P)
        assert np.allclose(K3, K2) and np.allclose(R3, R2) and np.allclose(c3, c2)


# ---- the soft silhouette render (oracle/raster_oracle.soft_render; multiply_amd.render.soft_blend_selected in torch) ----------
SIGMA, GAMMA = 5e-5, 1e-4
BLUR = np.log(1.0 / 1e-4 - 1.0) * SIGMA


def test_soft_silhouette_of_a_fronto_parallel_triangle_is_the_sigmoid_of_the_squared_ndc_distance():
    # triangle at z = 3 whose vertical right edge projects to x = 40.2 px: a pixel centre at distance d px outside has
    # alpha = sigmoid(-(d * 2 / min(H, W))^2 / sigma), pixels well inside are opaque red, far pixels the white background
    z, x_right_px = 3.0, 40.2
    x1 = (x_right_px - CX) / FX * z
    v = np.array([[-0.5, -0.6, z], [x1, -0.6, z], [x1, 0.55, z]])
    red = np.tile([[1.0, 0.0, 0.0]], (3, 1))
    r = 24                                                    # a row inside the edge's vertical extent
    for faces in ([[0, 1, 2]], [[2, 1, 0]]):                  # either winding
        img = RO.soft_render(v, faces, red, EYE, ZERO, FX, FY, CX, CY, H, W)
        for c in (40, 41):                                    # centres at 40.5 and 41.5: 0.3 and 1.3 px outside
            d = (c + 0.5 - x_right_px) * 2.0 / min(H, W)
            want = 1.0 / (1.0 + np.exp(d * d / SIGMA))
            assert abs(img[r, c, 3] - want) < 1e-9, (c, img[r, c, 3], want)
            # the COLOUR is the face's however small its probability (the background's weight delta is 1e-10 once any face is
            # blended): only the alpha channel carries the silhouette, which is why the caller multiplies the two (:668)
            assert np.allclose(img[r, c, :3], [1, 1e-10 / (want + 1e-10), 1e-10 / (want + 1e-10)], rtol=1e-6, atol=0)
        assert BLUR < ((42.5 - x_right_px) * 2.0 / min(H, W)) ** 2                             # beyond the blur radius: untouched
        assert np.allclose(img[r, 42], [1, 1, 1, 0], atol=0) and np.allclose(img[0, 0], [1, 1, 1, 0], atol=0)
        assert np.allclose(img[r, 35], [1, 0, 0, 1], atol=1e-9)                               # 4 px inside: the face colour, opaque
        d_in = (x_right_px - 39.5) * 2.0 / min(H, W)                                          # 0.7 px inside: prob above one half
        assert abs(img[r, 39, 3] - 1.0 / (1.0 + np.exp(-d_in * d_in / SIGMA))) < 1e-9


def test_soft_blend_weights_follow_depth_and_keep_the_nearest_faces_only():
    # three coincident big triangles at depths 3.00, 3.02, 3.5 in red, green, blue: inside all of them every prob is 1, so
    # rgb = sum_k exp((zinv_k - zinv_max) / gamma) colour_k / (sum + delta); K = 2 drops the blue one entirely
    tri = np.array([[-3.0, -3.0, 1.0], [3.0, -3.0, 1.0], [0.0, 3.0, 1.0]])
    depths = [3.0, 3.02, 3.5]
    v = np.concatenate([tri * [d, d, d] for d in depths])
    faces = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
    cols = np.repeat(np.eye(3), 3, axis=0)
    idx, zbuf, dists, bary = RO.soft_fragments(v, faces, EYE, ZERO, FX, FY, CX, CY, H, W, BLUR, 3)
    assert (idx[24, 32] == [0, 1, 2]).all() and np.allclose(zbuf[24, 32], depths)
    img = RO.soft_render(v, faces, cols, EYE, ZERO, FX, FY, CX, CY, H, W, K=3)
    zinv = (100.0 - np.array(depths)) / 99.0
    w = np.exp((zinv - zinv.max()) / GAMMA)
    delta = max(np.exp((1e-10 - zinv.max()) / GAMMA), 1e-10)
    assert np.allclose(img[24, 32, :3], (w + delta) / (w.sum() + delta), rtol=1e-9) and img[24, 32, 3] == 1.0
    assert 0.1 < w[1] < 0.2 and w[2] < 1e-20                  # 0.02 behind: a visible share; 0.5 behind: none
    idx2 = RO.soft_fragments(v, faces, EYE, ZERO, FX, FY, CX, CY, H, W, BLUR, 2)[0]
    assert (idx2[24, 32] == [0, 1]).all()
    # K smaller than the candidates also changes the silhouette product: one face outside by 0.3 px behind two covering ones
    assert np.allclose(bary[24, 32].sum(-1), 1.0)


def test_torch_soft_blend_matches_the_restatement_and_its_gradient_a_finite_difference():
    import torch
    from multiply_amd.render import soft_blend_selected
    rs = np.random.RandomState(3)
    v, f = uv_sphere([0.05, -0.02, 3.0], 0.45, n_lat=8, n_lon=12)
    v2, f2 = uv_sphere([0.35, 0.1, 3.6], 0.5, n_lat=7, n_lon=10)
    verts, faces = np.concatenate([v, v2]), np.concatenate([f, f2 + len(v)])
    cols = np.concatenate([np.tile([[1.0, 0, 0]], (len(v), 1)), np.tile([[0, 1.0, 0]], (len(v2), 1))]) * rs.uniform(0.5, 1, (len(verts), 1))
    Rm = np.array([[np.cos(0.1), 0, np.sin(0.1)], [0, 1, 0], [-np.sin(0.1), 0, np.cos(0.1)]])
    Tv = np.array([0.02, -0.01, 0.1])
    K = 6                                                        # fewer slots than candidates at many pixels
    blur = BLUR
    idx, zbuf, dists, bary = RO.soft_fragments(verts, faces, Rm, Tv, FX, FY, CX, CY, H, W, blur, K)
    fc = cols[faces]
    tex = (bary[..., None] * fc[np.maximum(idx, 0)]).sum(-2) * (idx >= 0)[..., None]
    want = RO.softmax_rgb_blend(idx, zbuf, dists, tex, SIGMA, GAMMA)
    assert (idx[..., K - 1] >= 0).sum() > 50 and ((idx[..., 0] >= 0) & (idx[..., K - 1] < 0)).sum() > 50   # full and partly filled lists
    tv = torch.tensor(verts, dtype=torch.float64, requires_grad=True)
    args = (torch.tensor(faces), torch.tensor(cols), torch.tensor(idx), torch.tensor(Rm), torch.tensor(Tv),
            torch.tensor([FX, FY]), torch.tensor([CX, CY]), H, W)
    act, got = soft_blend_selected(tv, *args, chunk=300)
    assert act.shape[0] == int((idx[..., 0] >= 0).sum())
    assert np.abs(got.detach().numpy() - want[act[:, 0].numpy(), act[:, 1].numpy()]).max() < 1e-9
    assert np.allclose(want[idx[..., 0] < 0], [1, 1, 1, 0])
    # d (weighted image sum) / d vertex coordinates against a central difference of the torch blend at a FIXED selection
    wgt = torch.tensor(rs.normal(size=(act.shape[0], 4)))
    (got * wgt).sum().backward()
    g = tv.grad.numpy()
    assert np.abs(g).max() > 1e-3
    order = np.argsort(-np.abs(g).reshape(-1))[:6]
    for flat in order:
        i, j = divmod(int(flat), 3)
        h = 1e-7
        vals = []
        for s in (+1, -1):
            vv = verts.copy()
            vv[i, j] += s * h
            vals.append(float((soft_blend_selected(torch.tensor(vv), *args)[1] * wgt).sum()))
        fd = (vals[0] - vals[1]) / (2 * h)
        assert abs(fd - g[i, j]) < 2e-4 * max(1.0, abs(fd)), (i, j, fd, g[i, j])
