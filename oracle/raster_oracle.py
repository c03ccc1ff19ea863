"""CPU restatement of the mesh z-buffer and of the mesh-space losses built on it -- TEST INFRASTRUCTURE ONLY (tests/; the
product path is multiply_amd/render.py + multiply_amd/mesh_losses.py + csrc/raster.hip).

PARITY UNPINNED for the rasterisation rule: the reference delegates it to pytorch3d (code/lib/model/render.py:1-18, 64-66;
README.md:15, no version pin), which is absent here and is not under /root/reference, and the reference holds no test or
golden image for it.  `rasterize` restates pytorch3d's published naive rasteriser (csrc/rasterize_meshes: pixel centre
sampling, barycentric coverage with area + 1e-8, perspective-correct depth, nearest face) in float64, in the camera
convention render.py:69-78 sets up (an OpenCV world->camera R, T whose x / y rows pytorch3d mirrors, i.e. the image is the
OpenCV projection u = fx X/Z + cx, v = fy Y/Z + cy).  It is anchored on analytic cases in tests/test_raster_cpu.py (planes
at known depth, a tessellated sphere against the ray / sphere intersection).

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
