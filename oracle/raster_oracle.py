"""CPU restatement of the mesh z-buffer and of the mesh-space losses built on it -- TEST INFRASTRUCTURE ONLY (tests/; the
product path is multiply_amd/render.py + multiply_amd/mesh_losses.py + csrc/raster.hip).

PARITY UNPINNED for the rasterisation rule: the reference delegates it to pytorch3d (code/lib/model/render.py:1-18, 64-66;
README.md:15, no version pin), which is absent here and is not under /root/reference, and the reference holds no test or
golden image for it.  `rasterize` restates pytorch3d's published naive rasteriser (csrc/rasterize_meshes: pixel centre
sampling, barycentric coverage with area + 1e-8, perspective-correct depth, nearest face) in float64, in the camera
convention render.py:69-78 sets up (an OpenCV world->camera R, T whose x / y rows pytorch3d mirrors, i.e. the image is the
OpenCV projection u = fx X/Z + cx, v = fy Y/Z + cy).  It is anchored on analytic cases in tests/test_raster_cpu.py (planes
at known depth, a tessellated sphere against the ray / sphere intersection).

`soft_render` restates, equally unpinned, what the reference's `softrender_multiple_meshes` asks of pytorch3d
(render.py:79-105, 121-133: BlendParams(sigma=5e-5, gamma=1e-4), blur_radius = log(1/1e-4 - 1) sigma, 100 faces per
pixel, SoftPhongShader under white ambient light): the blurred rasteriser (squared NDC distance to the face outline, signed;
clipped perspective-correct barycentrics; the faces_per_pixel nearest in z) followed by softmax_rgb_blend.

The loss arithmetic ON TOP of the depth maps is the reference's own code and is restated line by line:
  front_depth_and_masks   multiply_model.py:640-652 (torch) / :881-898 (numpy)
  depth_order_loss        multiply_model.py:653-736
  decompose_projection    multiply_model.py:553-576 (cv2.decomposeProjectionMatrix: RQ of the left 3x3 block, K is NOT
                          normalised by K[2,2]; the camera centre is the null vector of P)
"""
import numpy as np

K_EPS = 1e-8


def project(verts, R, T, fx, fy, cx, cy):
    cam = verts @ np.asarray(R, np.float64).T + np.asarray(T, np.float64)
    z = cam[:, 2]
    return np.stack([fx * cam[:, 0] / z + cx, fy * cam[:, 1] / z + cy, z], 1)


def _edge(px, py, a, b):
    return (px - a[0]) * (b[1] - a[1]) - (py - a[1]) * (b[0] - a[0])


def rasterize(verts, faces, R, T, fx, fy, cx, cy, H, W, z_clip=1e-6):
    """-> zbuf (H, W) float64 (-1 = empty), pix_to_face (H, W) int64 (-1), bary (H, W, 3).  Brute force: every face against
    every pixel centre (vectorised over the image per face).  pytorch3d skips faces entirely behind the camera (zmax < 0)
    and projects faces that straddle the camera plane to meaningless screen triangles; here, as in the kernel, a face with
    any vertex nearer than z_clip is dropped -- identical for every scene that is in front of the camera."""
    verts, faces = np.asarray(verts, np.float64), np.asarray(faces, np.int64)
    s = project(verts, R, T, fx, fy, cx, cy)
    py, px = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
    zbuf = np.full((H, W), np.inf)
    p2f = -np.ones((H, W), np.int64)
    bary = -np.ones((H, W, 3))
    for f, (i0, i1, i2) in enumerate(faces):
        v0, v1, v2 = s[i0], s[i1], s[i2]
        if not min(v0[2], v1[2], v2[2]) >= z_clip:
            continue
        area = _edge(v2[0], v2[1], v0, v1)
        if abs(area) <= K_EPS:
            continue
        a = area + K_EPS
        w0, w1, w2 = _edge(px, py, v1, v2) / a, _edge(px, py, v2, v0) / a, _edge(px, py, v0, v1) / a
        inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
        if not inside.any():
            continue
        t0, t1, t2 = w0 * v1[2] * v2[2], v0[2] * w1 * v2[2], v0[2] * v1[2] * w2
        d = np.maximum(t0 + t1 + t2, K_EPS)
        b0, b1, b2 = t0 / d, t1 / d, t2 / d
        pz = b0 * v0[2] + b1 * v1[2] + b2 * v2[2]
        win = inside & (pz >= 0) & (pz < zbuf)            # strict: the lower face id keeps a tie
        zbuf[win] = pz[win]
        p2f[win] = f
        bary[win] = np.stack([b0, b1, b2], -1)[win]
    zbuf[p2f < 0] = -1.0
    return zbuf, p2f, bary


def _seg_dist2(px, py, a, b):
    """squared distance of the pixel centres to segment ab (pytorch3d PointLineDistanceForward)"""
    bax, bay = b[0] - a[0], b[1] - a[1]
    l2 = bax * bax + bay * bay
    if l2 <= K_EPS:
        return (px - b[0]) ** 2 + (py - b[1]) ** 2
    t = np.clip(((px - a[0]) * bax + (py - a[1]) * bay) / l2, 0.0, 1.0)
    return (px - (a[0] + t * bax)) ** 2 + (py - (a[1] + t * bay)) ** 2


def soft_fragments(verts, faces, R, T, fx, fy, cx, cy, H, W, blur_radius, K, z_clip=1e-6):
    """The blurred rasteriser: per pixel the K candidate faces nearest in z (ties: lower face id), each with its signed
    squared distance (negative inside) and clipped perspective-correct barycentrics.  Distances are in pytorch3d's NDC
    units, where the SHORTER image side spans [-1, 1]: one pixel = 2 / min(H, W); NDC mirrors both axes with respect to the
    pixel grid, which changes neither distances nor orientations, so the arithmetic runs on pixel coordinates scaled by that
    factor.  -> idx (H, W, K) int64 (-1 = empty), zbuf, dists (H, W, K), bary (H, W, K, 3)."""
    verts, faces = np.asarray(verts, np.float64), np.asarray(faces, np.int64)
    s = project(verts, R, T, fx, fy, cx, cy)
    sc = 2.0 / min(H, W)
    s[:, :2] *= sc
    py, px = np.meshgrid((np.arange(H) + 0.5) * sc, (np.arange(W) + 0.5) * sc, indexing="ij")
    cand = [[[] for _ in range(W)] for _ in range(H)]
    rad = np.sqrt(blur_radius)
    for f, (i0, i1, i2) in enumerate(faces):
        v0, v1, v2 = s[i0], s[i1], s[i2]
        if not min(v0[2], v1[2], v2[2]) >= z_clip:
            continue
        area = _edge(v2[0], v2[1], v0, v1)
        if abs(area) <= K_EPS:
            continue
        xs, ys = (v0[0], v1[0], v2[0]), (v0[1], v1[1], v2[1])
        box = (px >= min(xs) - rad) & (px <= max(xs) + rad) & (py >= min(ys) - rad) & (py <= max(ys) + rad)
        if not box.any():
            continue
        a = area + K_EPS
        w0, w1, w2 = _edge(px, py, v1, v2) / a, _edge(px, py, v2, v0) / a, _edge(px, py, v0, v1) / a
        inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
        dist = np.minimum(np.minimum(_seg_dist2(px, py, v0, v1), _seg_dist2(px, py, v0, v2)), _seg_dist2(px, py, v1, v2))
        t0, t1, t2 = w0 * v1[2] * v2[2], v0[2] * w1 * v2[2], v0[2] * v1[2] * w2
        d = np.maximum(t0 + t1 + t2, K_EPS)
        b = np.maximum(np.stack([t0 / d, t1 / d, t2 / d], -1), 0.0)          # BarycentricClipForward
        b = b / np.maximum(b.sum(-1, keepdims=True), 1e-5)
        pz = b[..., 0] * v0[2] + b[..., 1] * v1[2] + b[..., 2] * v2[2]
        keep = box & (pz >= 0) & (inside | (dist < blur_radius))
        for r, c in zip(*np.nonzero(keep)):
            cand[r][c].append((pz[r, c], f, -dist[r, c] if inside[r, c] else dist[r, c], b[r, c]))
    idx = -np.ones((H, W, K), np.int64)
    zbuf, dists, bary = -np.ones((H, W, K)), -np.ones((H, W, K)), -np.ones((H, W, K, 3))
    for r in range(H):
        for c in range(W):
            for k, (z, f, dd, b) in enumerate(sorted(cand[r][c], key=lambda e: (e[0], e[1]))[:K]):
                idx[r, c, k], zbuf[r, c, k], dists[r, c, k], bary[r, c, k] = f, z, dd, b
    return idx, zbuf, dists, bary


def softmax_rgb_blend(idx, zbuf, dists, texels, sigma, gamma, background=(1.0, 1.0, 1.0), znear=1.0, zfar=100.0):
    """pytorch3d.renderer.blending.softmax_rgb_blend: texels (H, W, K, 3) -> (H, W, 4) RGBA"""
    eps = 1e-10
    mask = idx >= 0
    with np.errstate(over="ignore"):
        prob = mask / (1.0 + np.exp(dists / sigma))
    alpha = np.prod(1.0 - prob, axis=-1)
    z_inv = (zfar - zbuf) / (zfar - znear) * mask
    z_inv_max = np.maximum(z_inv.max(-1, keepdims=True), eps)
    w = prob * np.exp((z_inv - z_inv_max) / gamma)
    delta = np.maximum(np.exp((eps - z_inv_max) / gamma), eps)
    denom = w.sum(-1, keepdims=True) + delta
    rgb = ((w[..., None] * texels).sum(-2) + delta * np.asarray(background, np.float64)) / denom
    return np.concatenate([rgb, 1.0 - alpha[..., None]], -1)


def soft_render(verts, faces, colors, R, T, fx, fy, cx, cy, H, W, sigma=5e-5, gamma=1e-4, K=100, blur=None):
    """render.py:121-133 for the joined scene: vertex colours `colors` (V, 3) -> (H, W, 4).  blur = None: the soft
    renderer's log(1/1e-4 - 1) sigma; render.py:107-119 (render_multiple_meshes) is the same blend with sigma = 1e-4 (the
    BlendParams default), K = 10 and blur = 0: only faces that cover the pixel centre, for which the barycentric clipping
    pytorch3d then switches off is the identity."""
    blur = np.log(1.0 / 1e-4 - 1.0) * sigma if blur is None else blur
    idx, zbuf, dists, bary = soft_fragments(verts, faces, R, T, fx, fy, cx, cy, H, W, blur, K)
    fc = np.asarray(colors, np.float64)[np.asarray(faces, np.int64)]            # (F, 3 corners, 3)
    tex = (bary[..., None] * fc[np.maximum(idx, 0)]).sum(-2)                     # interpolate_face_attributes
    return softmax_rgb_blend(idx, zbuf, dists, tex * (idx >= 0)[..., None], sigma, gamma)


def front_depth_and_masks(depth_maps, max_depth=999.0):
    """depth_maps: list of (H, W) with -1 = no hit.  -> stacked 'max' maps (H, W, P), front map (H, W), instance masks
    (P, H, W) = `depth_i == front` (multiply_model.py:881-898)."""
    mx = []
    for d in depth_maps:
        d = np.array(d, dtype=np.float64, copy=True)
        d[d < 0] = max_depth
        mx.append(d)
    mx = np.stack(mx, -1)
    front = mx.min(-1)
    masks = np.stack([np.asarray(d) == front for d in depth_maps], 0)
    return mx, front, masks


def depth_order_loss(depth_maps, sam_logits, epoch, depth_order_weight=0.005, milestone=1000, max_depth=999.0):
    """multiply_model.py:640-736 without the image dumps: sam_logits (H, W, P) raw mask logits (org_sam_mask)."""
    mx, front, _ = front_depth_and_masks(depth_maps, max_depth)
    valid = front < max_depth
    sam = 1.0 / (1.0 + np.exp(-np.asarray(sam_logits, np.float64)))
    ssum = sam.sum(-1)
    valid &= ssum <= 1 + 1e-2
    valid &= ssum >= 0.7
    idx = sam.argmax(-1)
    gt = np.take_along_axis(mx, idx[..., None], -1)[..., 0]
    valid &= gt < max_depth
    gt, fr = gt[valid], front[valid]
    ex = ~(gt == fr)
    if ex.sum() == 0:
        return 0.0
    loss = np.log(1 + np.exp(gt[ex] - fr[ex])).sum()
    return depth_order_weight * (1 - min(milestone, epoch) / milestone) * loss


def decompose_projection(P):
    """cv2.decomposeProjectionMatrix restated: P (3, 4) -> K (3, 3) upper triangular with K[0,0], K[1,1] > 0 (not
    normalised), R (3, 3) a proper rotation, c (3,) camera centre; P[:, :3] = K R and P [c; 1] = 0."""
    P = np.asarray(P, np.float64)
    M = P[:, :3]
    # RQ by Gram-Schmidt on the rows, bottom up
    r3 = M[2] / np.linalg.norm(M[2])
    r2 = M[1] - (M[1] @ r3) * r3
    r2 /= np.linalg.norm(r2)
    r1 = np.cross(r2, r3)
    R = np.stack([r1, r2, r3])
    K = M @ R.T
    for i in (0, 1):                       # positive focal lengths; keep det(R) = +1 by mirroring the last axis with it
        if K[i, i] < 0:
            K[:, i] *= -1; R[i] *= -1
    if np.linalg.det(R) < 0:
        K[:, 2] *= -1; R[2] *= -1
    K[np.tril_indices(3, -1)] = 0.0
    _, _, vt = np.linalg.svd(P)
    c = vt[-1]
    return K, R, c[:3] / c[3]
