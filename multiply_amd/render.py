"""Mesh rasteriser wrapper with the reference's `Renderer` surface (code/lib/model/render.py:26-157) over the HIP z-buffer
(csrc/raster.hip, mp_raster_zbuf).

What the callers of the hot path use (multiply_model.py:553-576 get_renderer, :396 / :634 / :875
render_multiple_depth_map) is built in full: the constructor from an intrinsic matrix and an image size, `set_camera(R, T)`
with an OpenCV world->camera pose, and per-mesh depth maps shaped like pytorch3d's `fragments.zbuf` -- (1, H, W, K) with
the nearest face in slot 0 and -1 where nothing is hit -- that back-propagate into the vertices.  The reference asks
pytorch3d for the K = 10 nearest faces and reads slot 0 only (multiply_model.py:641, :882); this class returns K = 1.

`render_multiple_meshes` (hard rasteriser, 10 faces per pixel, SoftPhongShader with the default blend parameters) and
`softrender_multiple_meshes` (sigma = 5e-5, gamma = 1e-4, 100 faces per pixel; feeds the silhouette term of
multiply_model.py:636-637, :721, whose weight is 0 in every shipped config) run the (blurred) rasteriser + softmax blend of
csrc/raster.hip (mp_raster_soft_bins, mp_raster_soft); the soft one back-propagates into the vertices through a torch
re-evaluation of the kernel's face selection.
"""
import ctypes as C

import numpy as np
import torch

from . import hip

Z_CLIP = 1e-6
# render.py:79-86: BlendParams(sigma=5e-5, gamma=1e-4), blur_radius = log(1 / 1e-4 - 1) sigma (a squared NDC distance), 100 faces
SOFT_SIGMA, SOFT_GAMMA, SOFT_K = 5e-5, 1e-4, 100
SOFT_BLUR = float(np.log(1.0 / 1e-4 - 1.0) * SOFT_SIGMA)
HARD_K = 10                                  # render.py:58-59: faces_per_pixel of the hard rasteriser


class Fragments:
    """the three pytorch3d `Fragments` fields the callers read"""

    def __init__(self, zbuf, pix_to_face, bary_coords):
        self.zbuf, self.pix_to_face, self.bary_coords = zbuf, pix_to_face, bary_coords


def decompose_projection(P):
    """cv2.decomposeProjectionMatrix as get_renderer uses it (multiply_model.py:565-570): P (3, 4) -> K (upper triangular,
    K[0,0], K[1,1] > 0, NOT divided by K[2,2]), R (proper rotation), camera centre c with P [c; 1] = 0."""
    import scipy.linalg
    P = np.asarray(P, np.float64)
    K, R = scipy.linalg.rq(P[:, :3])
    for i in (0, 1):
        if K[i, i] < 0:
            K[:, i] *= -1
            R[i] *= -1
    if np.linalg.det(R) < 0:
        K[:, 2] *= -1
        R[2] *= -1
    c = np.linalg.svd(P)[2][-1]
    return K, R, c[:3] / c[3]


def get_renderer(inputs):
    """multiply_model.py:553-576: the camera of the frame, with the SMPL scale folded into the projection (the meshes it
    renders are divided by that scale)."""
    img_size = inputs["img_size"]
    P = inputs["P"][0].detach().cpu().numpy().astype(np.float64)
    sp = inputs["smpl_params"]
    if not bool((sp[:, 0, 0] == sp[:, min(1, sp.shape[1] - 1), 0]).all()):
        raise AssertionError("the persons of a frame share one scale (multiply_model.py:558)")
    scale = float(sp[0, 0, 0])
    P_norm = np.eye(4)
    P_norm[:, :] = P
    P_norm = P_norm @ np.diag([scale, scale, scale, 1.0])
    K, R, c = decompose_projection(P_norm[:3])
    T = -R @ c
    r = Renderer(img_size=[int(img_size[0]), int(img_size[1])], cam_intrinsic=K)
    r.set_camera(torch.tensor(R)[None].float(), torch.tensor(T)[None].float())
    return r


def soft_blend_selected(verts, faces, colors, sel, R, T, focal, principal, H, W, chunk=8192):
    """The soft blend in torch for the pixels that blend at least one face, GIVEN the face selection sel (H, W, K; -1 padded,
    any order): verts (V, 3) world, faces (F, 3), colors (V, 3), R (3, 3) / T (3,) OpenCV world -> camera, focal / principal
    (2,) in pixels.  -> act (n, 2) [row, col], rgba (n, 4); differentiable in verts and colors (pytorch3d back-propagates
    through the distances, the clipped barycentrics and the depths of the selected faces)."""
    dev = verts.device
    act = (sel[..., 0] >= 0).nonzero(as_tuple=False)
    if act.shape[0] == 0:
        return act, torch.zeros(0, 4, device=dev) + 0.0 * verts.sum()
    R, T = R.to(dev), T.to(dev)
    sc = 2.0 / min(H, W)
    dt = torch.float64 if verts.dtype == torch.float64 else torch.float32       # float64: the CPU tests' finite differences
    R, T, focal, principal = R.to(dt), T.to(dt), focal.to(dt), principal.to(dt)
    cam = verts.reshape(-1, 3).to(dt) @ R.t() + T
    z = cam[:, 2]
    sxy = torch.stack([focal[0] * cam[:, 0] / z + principal[0],
                       focal[1] * cam[:, 1] / z + principal[1]], 1) * sc
    faces = faces.reshape(-1, 3).long().to(dev)
    cols = colors.reshape(-1, 3).to(device=dev, dtype=dt)
    eps = 1e-10

    def seg(p, a, b):
        ba = b - a
        l2 = (ba * ba).sum(-1)
        t = (((p - a) * ba).sum(-1) / l2.clamp(min=1e-30)).clamp(0.0, 1.0)
        q = a + t[..., None] * ba
        return torch.where(l2 <= 1e-8, ((p - b) ** 2).sum(-1), ((p - q) ** 2).sum(-1))

    def edge(p, a, b):
        return (p[..., 0] - a[..., 0]) * (b[..., 1] - a[..., 1]) - (p[..., 1] - a[..., 1]) * (b[..., 0] - a[..., 0])

    def blend(sxy, z, cols, rc):
        fsel = sel[rc[:, 0], rc[:, 1]].long()                                     # (n, K)
        mask = fsel >= 0
        tri = faces[fsel.clamp(min=0)]                                            # (n, K, 3)
        p = (torch.stack([rc[:, 1], rc[:, 0]], 1).to(dt) + 0.5)[:, None, :] * sc   # (n, 1, 2)
        v0, v1, v2 = sxy[tri[..., 0]], sxy[tri[..., 1]], sxy[tri[..., 2]]
        z0, z1, z2 = z[tri[..., 0]], z[tri[..., 1]], z[tri[..., 2]]
        a = edge(v2, v0, v1) + 1e-8
        w0, w1, w2 = edge(p, v1, v2) / a, edge(p, v2, v0) / a, edge(p, v0, v1) / a
        inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
        dist = torch.minimum(torch.minimum(seg(p, v0, v1), seg(p, v0, v2)), seg(p, v1, v2))
        sd = torch.where(inside, -dist, dist)
        t0, t1, t2 = w0 * z1 * z2, z0 * w1 * z2, z0 * z1 * w2
        d = (t0 + t1 + t2).clamp(min=1e-8)
        b = torch.stack([t0 / d, t1 / d, t2 / d], -1).clamp(min=0.0)
        b = b / b.sum(-1, keepdim=True).clamp(min=1e-5)
        pz = b[..., 0] * z0 + b[..., 1] * z1 + b[..., 2] * z2
        tex = (b[..., None] * cols[tri]).sum(-2)                                   # (n, K, 3)
        prob = torch.sigmoid(-sd / SOFT_SIGMA) * mask
        alpha = torch.prod(1.0 - prob, dim=-1)
        zinv = (100.0 - pz) / 99.0 * mask
        zmax = zinv.max(dim=-1, keepdim=True).values.clamp(min=eps)
        w = prob * torch.exp((zinv - zmax) / SOFT_GAMMA)
        delta = torch.exp((eps - zmax) / SOFT_GAMMA).clamp(min=eps)
        den = w.sum(-1, keepdim=True) + delta
        rgb = ((w[..., None] * tex).sum(-2) + delta) / den                         # white background
        return torch.cat([rgb, 1.0 - alpha[:, None]], 1)

    # a chunk's (n, K = 100, ...) intermediates are recomputed in the backward pass instead of kept: without the checkpoint
    # autograd holds every chunk's until backward (several GB for a 512 x 512 frame with the blur halo)
    grad = torch.is_grad_enabled() and (sxy.requires_grad or cols.requires_grad)
    parts = []
    for i in range(0, act.shape[0], chunk):
        rc = act[i:i + chunk]
        if grad:
            from torch.utils.checkpoint import checkpoint
            parts.append(checkpoint(blend, sxy, z, cols, rc, use_reentrant=False))
        else:
            parts.append(blend(sxy, z, cols, rc))
    return act, torch.cat(parts)


class Renderer:
    def __init__(self, focal_length=None, principal_point=None, img_size=None, cam_intrinsic=None, device="cuda"):
        hip.require_device()
        self.device = torch.device(device)
        self.cam_intrinsic = np.asarray(cam_intrinsic, np.float64)
        self.image_size = [int(img_size[0]), int(img_size[1])]                  # (H, W)
        self.render_img_size = int(np.max(self.image_size))
        k = self.cam_intrinsic.astype(np.float32)
        self.focal_length = torch.tensor([[k[0, 0], k[1, 1]]], device=self.device)       # render.py:38-42 (no skew, raw K)
        self.principal_point = torch.tensor([[k[0, 2], k[1, 2]]], device=self.device)
        self.cam_R = torch.eye(3)[None]          # OpenCV convention here; render.py:44-48 holds the same pose mirrored
        self.cam_T = torch.zeros(1, 3)
        self._keys = self._big = None

    def set_camera(self, R, T):
        """R (1, 3, 3), T (1, 3): OpenCV world -> camera (render.py:69-78 mirrors x / y for pytorch3d, which mirrors them
        back when it maps to the screen: the image is the plain OpenCV projection)."""
        self.cam_R = R.detach().float().cpu().reshape(1, 3, 3).clone()
        self.cam_T = T.detach().float().cpu().reshape(1, 3).clone()

    # ------------------------------------------------------------------------------------------------------------------
    def _cam16(self):
        fl, pp = self.focal_length[0].cpu(), self.principal_point[0].cpu()
        v = torch.cat([self.cam_R.reshape(9), self.cam_T.reshape(3), fl, pp]).numpy().astype(np.float32)
        return (C.c_float * 16)(*v.tolist())

    def rasterize(self, verts, faces):
        """verts (V, 3) / (1, V, 3) world space, faces (F, 3) / (1, F, 3) -> Fragments with zbuf (1, H, W, 1) [detached],
        pix_to_face (1, H, W, 1) int64, bary_coords (1, H, W, 1, 3)."""
        H, W = self.image_size
        v = verts.detach().reshape(-1, 3).float().contiguous().to(self.device)
        f = faces.reshape(-1, 3).to(device=self.device, dtype=torch.int32).contiguous()
        if self._keys is None or self._keys.numel() < H * W:
            self._keys = torch.empty(H * W, dtype=torch.int64, device=self.device)
        if self._big is None or self._big.numel() < f.shape[0] + 1:
            self._big = torch.empty(f.shape[0] + 1, dtype=torch.int32, device=self.device)
        zbuf = torch.empty(H, W, dtype=torch.float32, device=self.device)
        p2f = torch.empty(H, W, dtype=torch.int32, device=self.device)
        bary = torch.empty(H, W, 3, dtype=torch.float32, device=self.device)
        cam = self._cam16()
        hip.check(hip.lib().mp_raster_zbuf(hip.ptr(v), v.shape[0], hip.ptr(f), f.shape[0], C.cast(cam, C.c_void_p),
                                           Z_CLIP, H, W, hip.ptr(self._keys), hip.ptr(self._big), hip.ptr(zbuf),
                                           hip.ptr(p2f), hip.ptr(bary), hip.stream()), "mp_raster_zbuf")
        return Fragments(zbuf[None, :, :, None], p2f.long()[None, :, :, None], bary[None, :, :, None, :])

    def _depth_with_grad(self, verts, faces, frag):
        """the z-buffer again for the covered pixels only, in torch, so that d depth / d vertices exists (pytorch3d's
        rasteriser back-propagates through the barycentrics and the vertex depths); visibility is the kernel's."""
        H, W = self.image_size
        p2f = frag.pix_to_face[0, :, :, 0]
        hit = (p2f >= 0).nonzero(as_tuple=False)
        if hit.shape[0] == 0:
            return frag.zbuf + 0.0 * verts.sum()
        tri = verts.reshape(-1, 3)[faces.reshape(-1, 3).long().to(verts.device)[p2f[hit[:, 0], hit[:, 1]]]]   # (N, 3, 3)
        R, T = self.cam_R[0].to(verts.device), self.cam_T[0].to(verts.device)
        cam = tri @ R.t() + T
        z = cam[..., 2]
        x = self.focal_length[0, 0] * cam[..., 0] / z + self.principal_point[0, 0]
        y = self.focal_length[0, 1] * cam[..., 1] / z + self.principal_point[0, 1]
        px, py = hit[:, 1].float() + 0.5, hit[:, 0].float() + 0.5
        edge = lambda ax, ay, bx, by: (px - ax) * (by - ay) - (py - ay) * (bx - ax)
        area = (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0]) - (y[:, 2] - y[:, 0]) * (x[:, 1] - x[:, 0]) + 1e-8
        w0, w1, w2 = edge(x[:, 1], y[:, 1], x[:, 2], y[:, 2]) / area, edge(x[:, 2], y[:, 2], x[:, 0], y[:, 0]) / area, \
            edge(x[:, 0], y[:, 0], x[:, 1], y[:, 1]) / area
        t0, t1, t2 = w0 * z[:, 1] * z[:, 2], z[:, 0] * w1 * z[:, 2], z[:, 0] * z[:, 1] * w2
        d = (t0 + t1 + t2).clamp(min=1e-8)
        pz = (t0 * z[:, 0] + t1 * z[:, 1] + t2 * z[:, 2]) / d
        flat = frag.zbuf.reshape(-1)
        lin = hit[:, 0] * W + hit[:, 1]
        return flat.index_add(0, lin, pz - pz.detach()).reshape(1, H, W, 1)      # forward value = the kernel's, bit for bit

    def render_multiple_depth_map(self, verts_list, faces_list, verts_colors_list=None):
        """render.py:134-157: one z-buffer per mesh, each (1, H, W, 1); depth = camera-space z of the nearest face, -1 = none"""
        out = []
        for v, f in zip(verts_list, faces_list):
            frag = self.rasterize(v, f)
            out.append(self._depth_with_grad(v, f, frag) if (torch.is_grad_enabled() and v.requires_grad) else frag.zbuf)
        return out

    def _join(self, verts_list, faces_list, verts_colors_list):
        """join_meshes_as_scene: one vertex / face / colour table"""
        nv = np.cumsum([0] + [v.reshape(-1, 3).shape[0] for v in verts_list])
        verts = torch.cat([v.reshape(-1, 3).float() for v in verts_list])
        faces = torch.cat([f.reshape(-1, 3).long() + int(o) for f, o in zip(faces_list, nv[:-1])])
        cols = torch.cat([c.reshape(-1, 3).float() for c in verts_colors_list])
        return verts, faces, cols

    def render_multiple_meshes(self, verts_list, faces_list, verts_colors_list):
        """render.py:107-119: the meshes joined as one scene through the hard rasteriser (blur radius 0, the 10 nearest
        covering faces per pixel) and SoftPhongShader under a white ambient light with BlendParams' defaults (sigma = gamma
        = 1e-4) -> (1, H, W, 4) RGBA over white: the depth-softmax of the covering faces' colours -- the nearest face's
        unless another lies within ~0.05 depth units -- and A = 1 - prod(1 - sigmoid(d^2 / sigma)) over them (d = the pixel
        centre's NDC distance to the face outline: below 1 where the faces are small).  No gradient (visualisation)."""
        verts, faces, cols = self._join(verts_list, faces_list, verts_colors_list)
        with torch.no_grad():
            return self.soft_rasterize(verts, faces, cols, sigma=1e-4, gamma=1e-4, blur=0.0, K=HARD_K)[0][None]

    def soft_rasterize(self, verts, faces, colors, want_sel=False, sigma=None, gamma=None, blur=None, K=None):
        """csrc/raster.hip mp_raster_soft_bins + mp_raster_soft on one joined mesh: -> image (H, W, 4), and with want_sel the
        (H, W, K) faces each pixel blended (-1 padded).  One host read (the length of the tile lists).  Defaults: the soft
        renderer's sigma / gamma / blur radius / faces per pixel."""
        sigma, gamma = SOFT_SIGMA if sigma is None else sigma, SOFT_GAMMA if gamma is None else gamma
        blur, K = SOFT_BLUR if blur is None else blur, SOFT_K if K is None else K
        H, W = self.image_size
        v = verts.detach().reshape(-1, 3).float().contiguous().to(self.device)
        f = faces.reshape(-1, 3).to(device=self.device, dtype=torch.int32).contiguous()
        c = colors.detach().reshape(-1, 3).float().contiguous().to(self.device)
        if c.shape[0] != v.shape[0]:
            raise ValueError("one colour per vertex")
        T = ((H + 7) // 8) * ((W + 7) // 8)
        tile_n = torch.empty(T, dtype=torch.int32, device=self.device)
        offsets = torch.empty(T + 1, dtype=torch.int32, device=self.device)
        cam = self._cam16()
        L = hip.lib()
        hip.check(L.mp_raster_soft_bins(hip.ptr(v), v.shape[0], hip.ptr(f), f.shape[0], C.cast(cam, C.c_void_p), Z_CLIP, H, W,
                                        blur, hip.ptr(tile_n), hip.ptr(offsets), hip.stream()), "mp_raster_soft_bins")
        total = int(offsets[T])
        if total < 0:
            raise RuntimeError("soft render: the tile lists exceed 2^31 entries")
        lst = torch.empty(max(total, 1), dtype=torch.int32, device=self.device)
        image = torch.empty(H, W, 4, dtype=torch.float32, device=self.device)
        sel = torch.empty(H, W, K, dtype=torch.int32, device=self.device) if want_sel else None
        bg = (C.c_float * 3)(1.0, 1.0, 1.0)                                       # BlendParams' default background
        hip.check(L.mp_raster_soft(hip.ptr(v), v.shape[0], hip.ptr(f), f.shape[0], hip.ptr(c), C.cast(cam, C.c_void_p), Z_CLIP,
                                   H, W, sigma, gamma, blur, K, 1.0, 100.0, C.cast(bg, C.c_void_p),
                                   hip.ptr(tile_n), hip.ptr(offsets), hip.ptr(lst), hip.ptr(image),
                                   hip.ptr(sel) if want_sel else None, hip.stream()), "mp_raster_soft")
        return image, sel

    def _soft_with_grad(self, verts, faces, colors, image, sel):
        """the blend once more in torch with the kernel's face selection, so that d image / d vertices (and colours) exists;
        the forward value stays the kernel's"""
        H, W = self.image_size
        act, soft = soft_blend_selected(verts, faces, colors, sel, self.cam_R[0], self.cam_T[0], self.focal_length[0],
                                        self.principal_point[0], H, W)
        lin = act[:, 0] * W + act[:, 1]
        return image.reshape(-1, 4).index_add(0, lin, soft - soft.detach()).reshape(H, W, 4)

    def softrender_multiple_meshes(self, verts_list, faces_list, verts_colors_list):
        """render.py:121-133: the meshes joined as one scene through the blurred rasteriser (sigma = 5e-5, gamma = 1e-4, 100
        faces per pixel) and SoftPhongShader under white ambient light -> (1, H, W, 4): RGB = depth-softmax blend of the
        interpolated vertex colours over white, A = 1 - prod(1 - sigmoid(-d / sigma)), the soft silhouette.  Differentiable
        in the vertices and colours."""
        verts, faces, cols = self._join(verts_list, faces_list, verts_colors_list)
        need_grad = torch.is_grad_enabled() and (verts.requires_grad or cols.requires_grad)
        image, sel = self.soft_rasterize(verts, faces, cols, want_sel=need_grad)
        if need_grad:
            image = self._soft_with_grad(verts.to(self.device), faces, cols.to(self.device), image, sel)
        return image[None]

    def render_mesh_recon(self, verts, faces, R=None, T=None, colors=None, mode="npat"):
        """render.py:161-208: shaded / normal / albedo / textured views side by side (along H), vertex normals as the
        area-weighted mean of the adjacent face normals (pytorch3d Meshes.verts_normals_list)."""
        with torch.no_grad():
            v, f = verts.reshape(-1, 3).float().to(self.device), faces.reshape(-1, 3).long().to(self.device)
            fn = torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=1)
            vn = torch.zeros_like(v).index_add_(0, f.reshape(-1), fn.repeat_interleave(3, 0))
            vn = torch.nn.functional.normalize(vn, eps=1e-6, dim=1)
            shades = vn[:, 2:3].clamp(min=0).expand(-1, 3)                            # front light (0, 0, 1)
            res = []
            if "p" in mode:
                res.append(self.render_multiple_meshes([v], [f], [shades]))
            if "n" in mode:
                res.append(self.render_multiple_meshes([v], [f], [(vn * 0.5 + 0.5)[:, [2, 1, 0]]]))
            if "a" in mode:
                assert colors is not None
                res.append(self.render_multiple_meshes([v], [f], [colors.reshape(-1, 3).to(self.device)]))
            if "t" in mode:
                assert colors is not None
                res.append(self.render_multiple_meshes([v], [f], [colors.reshape(-1, 3).to(self.device) * shades]))
            return torch.cat(res, dim=1)
