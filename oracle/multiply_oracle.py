"""CPU oracle: an fp32 restatement (torch, CPU) of the MultiPly volume-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under multiply_amd/ may import this file; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker / baseline.

Pinning: the building blocks below are pinned against golden vectors produced by importing the
reference's own modules in the build container (tests/golden/make_golden.py -> tests/golden/*.npz;
tests/test_oracle_golden.py).  Two steps rest on third-party code that is absent from /root/reference
and therefore PARITY-UNPINNED: (1) the packed compositing of nerfacc (unpinned version,
requirement.txt:13; call sites multiply.py:455-478) is restated from nerfacc's published
render_transmittance_from_density (alpha = 1-exp(-sigma*dt), T = exp(-exclusive_cumsum(sigma*dt)))
and cross-checked against the reference's own dense path Multiply.volume_rendering
(multiply.py:663-680); (2) trimesh's oriented bounding box (multiply.py:208-214) is not restated at
all: the set of rays that hit each person's box is an explicit input (`hit_index`).

All file:line citations are into /root/reference/code/.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- encodings / MLPs
def fourier_embed(x, multires):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]  (lib/model/embedders.py:8-34,36-49)."""
    out = [x]
    for k in range(multires):
        f = float(2.0 ** k)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def weight_norm_effective(g, v):
    """w = g * v / ||v|| with the norm per output row (torch.nn.utils.weight_norm, dim=0; networks.py:82-83)."""
    return v * (g / v.flatten(1).norm(dim=1, keepdim=True).view_as(g))


def linear_params(sd, prefix, l):
    """Effective (W, b) of layer `lin{l}` from a state dict that may hold weight-norm (g, v) pairs."""
    k = f"{prefix}lin{l}."
    if k + "weight_g" in sd:
        w = weight_norm_effective(sd[k + "weight_g"], sd[k + "weight_v"])
    else:
        w = sd[k + "weight"]
    return w, sd[k + "bias"]


def softplus100(x):
    """nn.Softplus(beta=100) (networks.py:85): x if 100x > 20 else log1p(exp(100x))/100."""
    return F.softplus(x, beta=100.0, threshold=20.0)


def implicit_forward(sd, prefix, x, cond_vec, multires, skip_in=(4,), n_lin=9):
    """ImplicitNet.forward (networks.py:126-208) for cond in {'smpl','frame'}: returns (N, 1+feat).

    x (N, d_in); cond_vec (C,) is concatenated to the embedded input of layer 0 (networks.py:164-165),
    the embedded input is re-injected at the skip layer and divided by sqrt(2) (networks.py:166-167).
    """
    emb = fourier_embed(x, multires) if multires > 0 else x
    h = emb
    for l in range(n_lin):
        w, b = linear_params(sd, prefix, l)
        if l == 0 and cond_vec is not None:
            h = torch.cat([h, cond_vec.view(1, -1).expand(h.shape[0], -1)], -1)
        if l in skip_in:
            h = torch.cat([h, emb], 1) / np.sqrt(2)
        h = F.linear(h, w, b)
        if l < n_lin - 1:
            h = softplus100(h)
    return h


def rendering_forward_pose_no_view(sd, prefix, points, normals, body_pose, feats, n_lin=5):
    """RenderingNet.forward, mode 'pose_no_view' (networks.py:277-281,305-311): sigmoid(MLP([x_c, n, lin_pose(pose), feat]))."""
    pose8 = F.linear(body_pose.view(1, -1), sd[prefix + "lin_pose.weight"], sd[prefix + "lin_pose.bias"])
    h = torch.cat([points, normals, pose8.expand(points.shape[0], -1), feats], -1)
    for l in range(n_lin):
        w, b = linear_params(sd, prefix, l)
        h = F.linear(h, w, b)
        if l < n_lin - 1:
            h = torch.relu(h)
    return torch.sigmoid(h)


def rendering_forward_nerf_frame(sd, prefix, view_dirs, feats, frame_code, multires_view=4, n_lin=2):
    """RenderingNet.forward, mode 'nerf_frame_encoding' (networks.py:265-266,274-276,305-311)."""
    v = fourier_embed(view_dirs, multires_view)
    h = torch.cat([v, frame_code.view(1, -1).expand(v.shape[0], -1), feats], -1)
    for l in range(n_lin):
        w, b = linear_params(sd, prefix, l)
        h = F.linear(h, w, b)
        if l < n_lin - 1:
            h = torch.relu(h)
    return torch.sigmoid(h)


def laplace_density(sdf, beta):
    """LaplaceDensity.density_func (density.py:20-29): (1/beta)(0.5 + 0.5 sign(s) expm1(-|s|/beta))."""
    alpha = 1.0 / beta
    return alpha * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


# ----------------------------------------------------------------------------- SMPL
class SMPLTables:
    """fp32 tensors of the SMPL model file (lib/smpl/body_models.py:186-225)."""

    def __init__(self, tables):
        f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
        self.v_template = f32(tables["v_template"])
        self.shapedirs = f32(np.asarray(tables["shapedirs"])[:, :, :10])
        pd = np.asarray(tables["posedirs"])
        self.posedirs = f32(pd.reshape(-1, pd.shape[-1]).T)          # (207, 20670)  body_models.py:212-214
        self.J_regressor = f32(tables["J_regressor"])
        self.lbs_weights = f32(tables["weights"])
        parents = np.asarray(tables["kintree_table"])[0].astype(np.int64).copy()
        parents[0] = -1
        self.parents = parents


def rodrigues(rot_vecs):
    """batch_rodrigues (lib/smpl/lbs.py:276-307): note the angle is ||r + 1e-8||, the axis r/angle."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    axis = rot_vecs / angle
    c, s = torch.cos(angle)[:, :, None], torch.sin(angle)[:, :, None]
    rx, ry, rz = axis[:, 0], axis[:, 1], axis[:, 2]
    z = torch.zeros_like(rx)
    K = torch.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).view(-1, 3, 3)
    I = torch.eye(3, dtype=rot_vecs.dtype)[None]
    return I + s * K + (1 - c) * torch.bmm(K, K)


def smpl_lbs(T, betas, full_pose):
    """lbs() (lib/smpl/lbs.py:136-229) for batch 1.  betas (10,), full_pose (72,).

    Returns verts (V,3), joints (24,3) [posed], A (24,4,4) [relative bone transforms], W (V,24)."""
    v_shaped = T.v_template + torch.einsum("l,mkl->mk", betas, T.shapedirs)             # lbs.py:184, 252-273
    J = torch.einsum("ik,ji->jk", v_shaped, T.J_regressor)                               # lbs.py:188, 232-249
    R = rodrigues(full_pose.view(-1, 3))                                                 # (24,3,3)
    pose_feature = (R[1:] - torch.eye(3)).reshape(1, -1)                                 # lbs.py:199
    v_posed = v_shaped + torch.matmul(pose_feature, T.posedirs).view(-1, 3)              # lbs.py:201-212
    # kinematic chain (lbs.py:323-377)
    rel = J.clone()
    rel[1:] = rel[1:] - J[T.parents[1:]]
    tm = torch.zeros(24, 4, 4)
    tm[:, :3, :3] = R
    tm[:, :3, 3] = rel
    tm[:, 3, 3] = 1.0
    chain = [tm[0]]
    for i in range(1, 24):
        chain.append(torch.matmul(chain[int(T.parents[i])], tm[i]))
    G = torch.stack(chain, 0)
    posed_joints = G[:, :3, 3].clone()
    Jh = torch.cat([J, torch.zeros(24, 1)], 1)[:, :, None]                               # (24,4,1)
    A = G - F.pad(torch.matmul(G, Jh), [3, 0])                                           # lbs.py:374-375
    Tv = torch.matmul(T.lbs_weights, A.view(24, 16)).view(-1, 4, 4)                      # lbs.py:217-221
    vh = torch.cat([v_posed, torch.ones(v_posed.shape[0], 1)], 1)[:, :, None]
    verts = torch.matmul(Tv, vh)[:, :3, 0]
    return verts, posed_joints, A, T.lbs_weights


def canonical_thetas():
    """The "A-pose" of SMPLServer.__init__ (lib/model/smpl.py:34-39): theta[5]=pi/6, theta[8]=-pi/6."""
    th = torch.zeros(72)
    th[5] = np.pi / 6
    th[8] = -np.pi / 6
    return th


def smpl_server_forward(T, scale, transl, thetas, betas, tfs_c_inv=None):
    """SMPLServer.forward (lib/model/smpl.py:50-94), batch 1.  scale (), transl (3,), thetas (72,), betas (10,).

    With tfs_c_inv (24,4,4) the bone transforms are made relative to the canonical pose (smpl.py:90-91)."""
    verts, joints, A, W = smpl_lbs(T, betas, thetas)
    out_verts = verts * scale + transl * scale                                           # smpl.py:77-78
    out_joints = joints * scale + transl * scale
    top = A[:, :3, :] * scale                                                            # smpl.py:87
    top = torch.cat([top[:, :, :3], top[:, :, 3:] + (transl * scale).reshape(1, 3, 1)], 2)   # smpl.py:88
    tf = torch.cat([top, A[:, 3:, :]], 1)                                                # (out-of-place: autograd-safe)
    if tfs_c_inv is not None:
        tf = torch.einsum("nij,njk->nik", tf, tfs_c_inv)
    return dict(smpl_verts=out_verts, smpl_jnts=out_joints, smpl_tfs=tf, smpl_weights=W)


class SMPLServerOracle:
    """Canonical-pose precomputation of SMPLServer.__init__ (smpl.py:34-47) + forward."""

    def __init__(self, tables, betas):
        self.T = tables if isinstance(tables, SMPLTables) else SMPLTables(tables)
        self.betas = torch.as_tensor(np.asarray(betas), dtype=torch.float32)
        out = smpl_server_forward(self.T, torch.tensor(1.0), torch.zeros(3), canonical_thetas(), self.betas)
        self.verts_c = out["smpl_verts"]
        self.joints_c = out["smpl_jnts"]
        self.tfs_c_inv = out["smpl_tfs"].inverse()
        self.weights = out["smpl_weights"]

    def forward(self, scale, transl, thetas, betas):
        return smpl_server_forward(self.T, scale, transl, thetas, betas, self.tfs_c_inv)


# ----------------------------------------------------------------------------- deformer
def knn1(pts, verts, chunk=8192):
    """Exact nearest vertex (pytorch3d.ops.knn_points with K=1; deformer.py:39): (squared dist, index)."""
    d_all, i_all = [], []
    for s in range(0, pts.shape[0], chunk):
        p = pts[s:s + chunk]
        d2 = ((p[:, None, :] - verts[None, :, :]) ** 2).sum(-1)
        d, i = d2.min(1)
        d_all.append(d)
        i_all.append(i)
    return torch.cat(d_all), torch.cat(i_all)


def query_weights(pts, verts, skin_w):
    """SMPLDeformer.query_skinning_weights_smpl_multi with K=1 (deformer.py:37-50): weights row of the
    nearest vertex (the exp(-d)/sum normalisation is identically 1 for K=1), outlier = dist > 0.1."""
    d2, idx = knn1(pts, verts)
    d2 = torch.clamp(d2, max=4)
    return skin_w[idx].detach(), torch.sqrt(d2) > 0.1, idx


def query_weights_k(pts, verts, skin_w, K):
    """SMPLDeformer.query_skinning_weights_smpl_multi for general K (deformer.py:37-50): K nearest vertices (pytorch3d
    knn_points = exact K-NN on squared distances), conf = exp(-min(d2,4)) normalised over the K, weights =
    sum_k conf_k skin_w[idx_k]; outlier from the nearest one."""
    d2 = ((pts[:, None, :] - verts[None, :, :]) ** 2).sum(-1)
    dk, idx = torch.topk(d2, K, dim=1, largest=False)
    dk = torch.clamp(dk, max=4)
    conf = torch.exp(-dk)
    conf = conf / conf.sum(-1, keepdim=True)
    w = (skin_w[idx] * conf[..., None]).sum(1)
    return w.detach(), torch.sqrt(dk[:, 0]) > 0.1


def skinning(x, w, tfs, inverse):
    """skinning() (deformer.py:72-88).  x (N,3), w (N,24), tfs (24,4,4)."""
    xh = F.pad(x, (0, 1), value=1.0)
    w_tf = torch.einsum("pn,nij->pij", w, tfs)
    if inverse:
        xh = torch.einsum("pij,pj->pi", w_tf.inverse(), xh)
    else:
        xh = torch.einsum("pij,pj->pi", w_tf, xh)
    return xh[:, :3]


def deform_inverse(x, tfs, posed_verts, skin_w):
    """SMPLDeformer.forward(inverse=True) (deformer.py:19-30): x_c and the outlier mask."""
    w, outlier, _ = query_weights(x, posed_verts, skin_w)
    return skinning(x, w, tfs, inverse=True), outlier


# ----------------------------------------------------------------------------- canonical-mesh flags
def mesh_signed_distance(pts, face_verts, chunk=256):
    """kaolin.metrics.trianglemesh.point_to_mesh_distance (sqrt taken, multiply.py:155-157) and kaolin.ops.mesh.check_sign
    (multiply.py:158-160) restated in float64: |d| = distance to the closest triangle; negative when the ray
    p + t(1,0,0), t > 0 crosses the surface an odd number of times.  kaolin 0.13 is third party and absent from
    /root/reference: parity unpinned (closed-form geometry; differences can only arise in degenerate ray/edge hits).
    pts (N,3), face_verts (F,3,3) -> (N,) float64."""
    P = pts.double()
    A, B, C = [face_verts[:, i].double() for i in range(3)]
    nrm = torch.cross(B - A, C - A, dim=-1)
    n2 = (nrm * nrm).sum(-1)
    out = torch.empty(P.shape[0], dtype=torch.float64)

    def seg(p, u, v):
        uv = (v - u)[None]
        t = (((p[:, None] - u[None]) * uv).sum(-1) / (uv * uv).sum(-1).clamp_min(1e-300)).clamp(0.0, 1.0)
        q = u[None] + t[..., None] * uv
        return ((p[:, None] - q) ** 2).sum(-1)

    for s0 in range(0, P.shape[0], chunk):
        p = P[s0:s0 + chunk]
        ap = p[:, None] - A[None]
        dpl = (ap * nrm[None]).sum(-1)
        inside = torch.ones(p.shape[0], A.shape[0], dtype=torch.bool)
        for u, v in ((A, B), (B, C), (C, A)):
            e = torch.cross((v - u)[None].expand(p.shape[0], -1, -1), p[:, None] - u[None], dim=-1)
            inside &= (e * nrm[None]).sum(-1) >= 0
        d2 = torch.where(inside & (n2[None] > 0), dpl ** 2 / n2[None].clamp_min(1e-300),
                         torch.minimum(torch.minimum(seg(p, A, B), seg(p, B, C)), seg(p, C, A)))
        dist = d2.min(1)[0].sqrt()
        # crossings of the +x ray: (y,z) crossing-number rule, then the x of the plane point
        cn = torch.zeros(p.shape[0], A.shape[0], dtype=torch.long)
        for u, v in ((A, B), (B, C), (C, A)):
            uy, vy = u[None, :, 1] > p[:, None, 1], v[None, :, 1] > p[:, None, 1]
            t = (p[:, None, 1] - u[None, :, 1]) / (v[None, :, 1] - u[None, :, 1])
            zc = u[None, :, 2] + t * (v[None, :, 2] - u[None, :, 2])
            cn += ((uy != vy) & (zc > p[:, None, 2])).long()
        x = A[None, :, 0] - (nrm[None, :, 1] * (p[:, None, 1] - A[None, :, 1]) +
                             nrm[None, :, 2] * (p[:, None, 2] - A[None, :, 2])) / nrm[None, :, 0]
        hit = (cn % 2 == 1) & (nrm[None, :, 0] != 0) & (x > p[:, None, 0])
        inside_mesh = hit.sum(1) % 2 == 1
        out[s0:s0 + chunk] = torch.where(inside_mesh, -dist, dist)
    return out


def off_in_surface_flags(x_cano, n_samples, face_verts, threshold=0.05):
    """Multiply.check_off_in_surface_points_cano_mesh (multiply.py:153-167)."""
    sd = mesh_signed_distance(x_cano, face_verts).reshape(-1, n_samples)
    m = sd.min(1)[0]
    return m > threshold, m <= 0.0, sd


# ----------------------------------------------------------------------------- scene model
class PersonOracle:
    def __init__(self, sd, p, server):
        self.sd = sd
        self.imp = f"foreground_implicit_network_list.{p}."
        self.ren = f"foreground_rendering_network_list.{p}."
        self.server = server          # SMPLServerOracle (model.smpl_server_list[p]; deformer uses the same betas)

    def implicit(self, x_c, cond):
        return implicit_forward(self.sd, self.imp, x_c, cond, multires=6)

    def sdf_func(self, x, cond, tfs, posed_verts, eval_mode):
        """Multiply.sdf_func_with_smpl_deformer (multiply.py:137-151)."""
        x_c, outlier = deform_inverse(x, tfs, posed_verts, self.server.weights)
        out = self.implicit(x_c, cond)
        sdf = out[:, 0:1].clone()
        if eval_mode:
            sdf[outlier] = 4.0
        return sdf, x_c, out[:, 1:]


def get_camera_rays(uv, pose, intrinsics):
    """rend_util.get_camera_params + lift, pose-matrix branch (lib/utils/rend_util.py:45-87).  uv (R,2)."""
    fx, fy = intrinsics[0, 0], intrinsics[1, 1]
    cx, cy, sk = intrinsics[0, 2], intrinsics[1, 2], intrinsics[0, 1]
    x, y = uv[:, 0], uv[:, 1]
    z = torch.ones_like(x)
    x_lift = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_lift = (y - cy) / fy * z
    pc = torch.stack([x_lift, y_lift, z, torch.ones_like(z)], -1)
    world = (pose @ pc.T).T[:, :3]
    cam = pose[:3, 3]
    return F.normalize(world - cam[None], dim=1), cam


def sphere_far(cam_loc, dirs, r):
    """Far root of rend_util.get_sphere_intersections (rend_util.py:131-147), clamped at 0."""
    b = (dirs * cam_loc).sum(-1, keepdim=True)
    under = b ** 2 - ((cam_loc ** 2).sum(-1, keepdim=True) - r ** 2)
    assert (under > 0).all(), "BOUNDING SPHERE PROBLEM"
    return (torch.sqrt(under) - b).clamp_min(0.0)


def error_bound(beta, sdf, dists, d_star):
    """ErrorBoundSampler.get_error_bound (ray_sampler.py:222-230). sdf (R,n), dists/d_star (R,n-1), beta (R,1) or ()."""
    dens = laplace_density(sdf, beta)
    shifted = torch.cat([torch.zeros(dists.shape[0], 1), dists * dens[:, :-1]], -1)
    integral = torch.cumsum(shifted, -1)
    err_sec = torch.exp(-d_star / beta) * (dists ** 2.0) / (4 * beta ** 2)
    err_int = torch.cumsum(err_sec, -1)
    bound = (torch.clamp(torch.exp(err_int), max=1.0e6) - 1.0) * torch.exp(-integral[:, :-1])
    return bound.max(-1)[0]


def inverse_cdf(bins, cdf, u):
    """ray_sampler.py:174-186."""
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


class SamplerCfg:
    def __init__(self, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10,
                 max_total_iters=5, add_tiny=1.0e-6, near=0.0, radius=3.0, N_bg=32):
        self.N_samples, self.N_samples_eval, self.N_samples_extra = N_samples, N_samples_eval, N_samples_extra
        self.eps, self.beta_iters, self.max_total_iters, self.add_tiny = eps, beta_iters, max_total_iters, add_tiny
        self.near, self.radius, self.N_bg = near, radius, N_bg


def error_bound_sample(cfg, dirs, cam, sdf_fn, beta0, draws=None):
    """ErrorBoundSampler.get_z_vals (ray_sampler.py:66-220), foreground part.

    dirs, cam (R,3); sdf_fn(points (N,3)) -> (N,1) is the no-grad SDF query of ray_sampler.py:85-88.
    draws=None reproduces `not model.training` (deterministic linspace draws); a dict with
    't_rand' (R,N_eval), 'u_final' (R,N_samples), 'extra_idx' (N_extra,) supplies the training-mode randomness
    (ray_sampler.py:38,171,202) as explicit inputs.
    Returns z_vals (R, N_samples+N_samples_extra+2) sorted, and the number of loop iterations run."""
    R = dirs.shape[0]
    far = sphere_far(cam, dirs, cfg.radius)
    near = cfg.near * torch.ones(R, 1)
    t = torch.linspace(0.0, 1.0, steps=cfg.N_samples_eval)
    z_vals = near * (1.0 - t) + far * t                                         # ray_sampler.py:29-30
    if draws is not None:                                                        # stratified jitter, :32-40
        mids = 0.5 * (z_vals[:, 1:] + z_vals[:, :-1])
        upper = torch.cat([mids, z_vals[:, -1:]], -1)
        lower = torch.cat([z_vals[:, :1], mids], -1)
        z_vals = lower + (upper - lower) * draws["t_rand"]
    samples, samples_idx = z_vals, None
    dists = z_vals[:, 1:] - z_vals[:, :-1]
    bound = (1.0 / (4.0 * torch.log(torch.tensor(cfg.eps + 1.0)))) * (dists ** 2.0).sum(-1)
    beta = torch.sqrt(bound)
    total_iters, not_converge = 0, True
    sdf = None
    while not_converge and total_iters < cfg.max_total_iters:
        pts = (cam[:, None, :] + samples[:, :, None] * dirs[:, None, :]).reshape(-1, 3)
        with torch.no_grad():
            samples_sdf = sdf_fn(pts)
        if samples_idx is not None:
            merged = torch.cat([sdf.reshape(-1, z_vals.shape[1] - samples.shape[1]),
                                samples_sdf.reshape(-1, samples.shape[1])], -1)
            sdf = torch.gather(merged, 1, samples_idx).reshape(-1, 1)
        else:
            sdf = samples_sdf
        d = sdf.reshape(z_vals.shape)
        dists = z_vals[:, 1:] - z_vals[:, :-1]
        a, b, c = dists, d[:, :-1].abs(), d[:, 1:].abs()
        first = a.pow(2) + b.pow(2) <= c.pow(2)
        second = a.pow(2) + c.pow(2) <= b.pow(2)
        d_star = torch.zeros(R, z_vals.shape[1] - 1)
        d_star[first] = b[first]
        d_star[second] = c[second]
        s = (a + b + c) / 2.0
        area = s * (s - a) * (s - b) * (s - c)
        mask = ~first & ~second & (b + c - a > 0)
        d_star[mask] = (2.0 * torch.sqrt(area[mask])) / (a[mask])
        d_star = (d[:, 1:].sign() * d[:, :-1].sign() == 1) * d_star              # :110

        curr = error_bound(beta0, d, dists, d_star)
        beta[curr <= cfg.eps] = beta0
        beta_min, beta_max = beta0 * torch.ones(R), beta
        for _ in range(cfg.beta_iters):
            mid = (beta_min + beta_max) / 2.0
            curr = error_bound(mid[:, None], d, dists, d_star)
            ok = curr <= cfg.eps
            beta_max = torch.where(ok, mid, beta_max)
            beta_min = torch.where(~ok, mid, beta_min)
        beta = beta_max

        dens = laplace_density(d, beta[:, None])
        dists_inf = torch.cat([dists, 1e10 * torch.ones(R, 1)], -1)
        free = dists_inf * dens
        shifted = torch.cat([torch.zeros(R, 1), free[:, :-1]], -1)
        alpha = 1 - torch.exp(-free)
        trans = torch.exp(-torch.cumsum(shifted, -1))
        weights = alpha * trans
        total_iters += 1
        not_converge = bool(beta.max() > beta0)
        more = not_converge and total_iters < cfg.max_total_iters
        if more:
            N = cfg.N_samples_eval
            err_sec = torch.exp(-d_star / beta[:, None]) * (dists_inf[:, :-1] ** 2.0) / (4 * beta[:, None] ** 2)
            err_int = torch.cumsum(err_sec, -1)
            bound_op = (torch.clamp(torch.exp(err_int), max=1.0e6) - 1.0) * trans[:, :-1]
            pdf = bound_op + cfg.add_tiny
        else:
            N = cfg.N_samples
            pdf = weights[:, :-1] + 1e-5
        pdf = pdf / pdf.sum(-1, keepdim=True)
        cdf = torch.cat([torch.zeros(R, 1), torch.cumsum(pdf, -1)], -1)
        if more or draws is None:
            u = torch.linspace(0.0, 1.0, steps=N)[None].repeat(R, 1)
        else:
            u = draws["u_final"]
        samples = inverse_cdf(z_vals, cdf, u.contiguous())
        if more:
            z_vals, samples_idx = torch.sort(torch.cat([z_vals, samples], -1), -1)

    if cfg.N_samples_extra > 0:
        if draws is not None:
            sidx = draws["extra_idx"]
            if sidx.dim() == 2:          # one row per possible list length n = N_eval * k
                sidx = sidx[z_vals.shape[1] // cfg.N_samples_eval - 1]
        else:
            sidx = torch.linspace(0, z_vals.shape[1] - 1, cfg.N_samples_extra).long()
        extra = torch.cat([near, far, z_vals[:, sidx]], -1)
    else:
        extra = torch.cat([near, far], -1)
    z_out, _ = torch.sort(torch.cat([samples, extra], -1), -1)
    return z_out, total_iters


def bg_depths(cfg, R, t_rand=None):
    """inverse_sphere_sampler (ray_sampler.py:64, 21-42) * 1/radius (multiply.py:482-483): (R, N_bg) in [0, 1/r]."""
    t = torch.linspace(0.0, 1.0, steps=cfg.N_bg)
    z = (0.0 * (1.0 - t) + 1.0 * t)[None].repeat(R, 1)
    if t_rand is not None:
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat([mids, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z * (1.0 / cfg.radius)


def depth2pts_outside(ray_o, ray_d, depth, radius=3.0):
    """Multiply.depth2pts_outside (multiply.py:698-726): NeRF++ inverted-sphere points (x,y,z,1/r)."""
    o_dot_d = (ray_d * ray_o).sum(-1)
    under = o_dot_d ** 2 - ((ray_o ** 2).sum(-1) - radius ** 2)
    d_sphere = torch.sqrt(under) - o_dot_d
    p_sphere = ray_o + d_sphere[..., None] * ray_d
    p_mid = ray_o - o_dot_d[..., None] * ray_d
    p_mid_norm = p_mid.norm(dim=-1)
    axis = torch.cross(ray_o, p_sphere, dim=-1)
    axis = axis / axis.norm(dim=-1, keepdim=True)
    phi = torch.asin(p_mid_norm / radius)
    theta = torch.asin(p_mid_norm * depth)
    ang = (phi - theta)[..., None]
    p_new = p_sphere * torch.cos(ang) + torch.cross(axis, p_sphere, dim=-1) * torch.sin(ang) + \
        axis * (axis * p_sphere).sum(-1, keepdim=True) * (1.0 - torch.cos(ang))
    p_new = p_new / p_new.norm(dim=-1, keepdim=True)
    return torch.cat([p_new, depth[..., None]], -1)


def bg_volume_weights(z_bg, bg_sdf):
    """Multiply.bg_volume_rendering with AbsDensity (multiply.py:682-696, density.py:32-34)."""
    dens = bg_sdf.abs().reshape(-1, z_bg.shape[1])
    dists = z_bg[:, :-1] - z_bg[:, 1:]
    dists = torch.cat([dists, 1e10 * torch.ones(dists.shape[0], 1)], -1)
    free = dists * dens
    shifted = torch.cat([torch.zeros(dists.shape[0], 1), free[:, :-1]], -1)
    alpha = 1 - torch.exp(-free)
    trans = torch.exp(-torch.cumsum(shifted, -1))
    return alpha * trans


def dense_volume_rendering(z_vals, z_max, sdf, beta):
    """Multiply.volume_rendering (multiply.py:663-680): the reference's own dense compositing (cross-check only)."""
    dens = laplace_density(sdf, beta).reshape(-1, z_vals.shape[1])
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], z_max[:, None] - z_vals[:, -1:]], -1)
    free = dists * dens
    shifted = torch.cat([torch.zeros(dists.shape[0], 1), free], -1)
    alpha = 1 - torch.exp(-free)
    trans = torch.exp(-torch.cumsum(shifted, -1))
    return alpha * trans[:, :-1], trans[:, -1]


def packed_composite(n_rays, hit_index, z_list, zmax_list, sdf_list, rgb_list, nrm_list, beta, person_ids):
    """multiply.py:425-480: pack every person's samples, sort by t_end then (stably) by ray, nerfacc weights.

    nerfacc (unpinned third party) restated: alpha_i = 1-exp(-sigma_i (t_end-t_start)),
    T_i = exp(-sum_{k<i in the same ray} sigma_k dt_k), w = alpha*T.  bg transmittance = the EXCLUSIVE T of each
    ray's last packed sample (multiply.py:457-463), 1 for rays without samples."""
    rows = []
    for k, pid in enumerate(person_ids):
        z, zmax = z_list[k], zmax_list[k]
        S = z.shape[1]
        zz = torch.cat([z, zmax[:, None]], 1)
        ray = hit_index[k].to(torch.float32)[:, None].expand(-1, S)
        rows.append(torch.cat([ray.reshape(-1, 1), zz[:, :-1].reshape(-1, 1), zz[:, 1:].reshape(-1, 1),
                               sdf_list[k].reshape(-1, 1), rgb_list[k].reshape(-1, 3), nrm_list[k].reshape(-1, 3),
                               torch.full((z.numel(), 1), float(pid))], 1))
    pk = torch.cat(rows, 0)
    pk = pk[torch.sort(pk[:, 2], stable=True)[1]]
    pk = pk[torch.sort(pk[:, 0], stable=True)[1]]
    ray_idx = pk[:, 0].long()
    sig_dt = laplace_density(pk[:, 3], beta) * (pk[:, 2] - pk[:, 1])
    alpha = 1.0 - torch.exp(-sig_dt)
    # exclusive per-ray cumsum (nerfacc scans each ray separately; a global scan minus the ray's base reproduces that
    # only if it is carried in float64 -- the float32 result is what is used)
    sd64 = sig_dt.double()
    csum = torch.cumsum(sd64, 0)
    first = torch.ones_like(ray_idx, dtype=torch.bool)
    first[1:] = ray_idx[1:] != ray_idx[:-1]
    start_pos = torch.nonzero(first).flatten()
    seg_id = torch.cumsum(first.long(), 0) - 1
    base = (csum - sd64)[start_pos][seg_id]
    excl = (csum - sd64 - base).float()
    trans = torch.exp(-excl)
    w = alpha * trans
    acc_rgb = torch.zeros(n_rays, 3).index_add_(0, ray_idx, w[:, None] * pk[:, 4:7])
    acc_nrm = torch.zeros(n_rays, 3).index_add_(0, ray_idx, w[:, None] * pk[:, 7:10])
    acc_w = torch.zeros(n_rays).index_add_(0, ray_idx, w)
    acc_person = []
    for pid in person_ids:
        m = pk[:, 10] == float(pid)
        acc_person.append(torch.zeros(n_rays).index_add_(0, ray_idx[m], w[m]))
    acc_person = torch.stack(acc_person, 1)
    last = torch.ones_like(first)
    last[:-1] = first[1:]
    bg_T = torch.ones(n_rays)
    bg_T[ray_idx[last]] = trans[last]
    return acc_rgb, acc_nrm, acc_w, acc_person, bg_T


class MultiplyOracle:
    """Eval-mode Multiply.forward (multiply.py:174-598) assembled from the pieces above."""

    def __init__(self, state_dict, smpl_tables, betas, cfg=None):
        self.sd = {k: v.detach().clone().float() for k, v in state_dict.items()}
        self.cfg = cfg or SamplerCfg()
        self.T = SMPLTables(smpl_tables)
        betas = np.asarray(betas, dtype=np.float32).reshape(-1, 10)
        self.P = betas.shape[0]
        self.servers = [SMPLServerOracle(self.T, betas[p]) for p in range(self.P)]
        self.persons = [PersonOracle(self.sd, p, self.servers[p]) for p in range(self.P)]

    def beta(self):
        return self.sd["density.beta"].abs() + 1e-4                                # density.py:31-33

    def shade(self, person, x_c, cond, tfs, create_graph=False):
        """forward_gradient + get_rbg_value (multiply.py:600-661): normals and colours at canonical points.

        J = d(forward_skinning)/d x_c with the canonical-space nearest-vertex weights (deformer.py:31-35); because
        the weights are detached, J is the upper-left 3x3 of the blended transform."""
        sv = person.server
        if not x_c.requires_grad:       # (training with pose optimisation: x_c carries the graph back to the bone transforms,
            x_c = x_c.detach().requires_grad_(True)     # multiply.py:623 `pnts_c.requires_grad_(True)` on a non-leaf)
        w_c, _, _ = query_weights(x_c.detach(), sv.verts_c, sv.weights)
        Jm = torch.einsum("pn,nij->pij", w_c, tfs)[:, :3, :3]
        out = person.implicit(x_c, cond)
        sdf = out[:, :1]
        grads = torch.autograd.grad(sdf, x_c, torch.ones_like(sdf), create_graph=create_graph)[0]
        g = torch.einsum("bi,bij->bj", grads, Jm.inverse())
        normals = F.normalize(F.normalize(g, dim=1), dim=-1, eps=1e-6)
        rgb = rendering_forward_pose_no_view(self.sd, person.ren, x_c, normals, cond, out[:, 1:])
        return rgb[:, :3], normals, sdf

    @torch.no_grad()
    def background(self, dirs, cam, frame_code, t_rand=None):
        R = dirs.shape[0]
        z_bg = torch.flip(bg_depths(self.cfg, R, t_rand), dims=[-1])                 # multiply.py:516
        N = z_bg.shape[1]
        pts = depth2pts_outside(cam[:, None, :].expand(-1, N, -1), dirs[:, None, :].expand(-1, N, -1), z_bg,
                                self.cfg.radius).reshape(-1, 4)
        out = implicit_forward(self.sd, "bg_implicit_network.", pts, frame_code, multires=10)
        rgb = rendering_forward_nerf_frame(self.sd, "bg_rendering_network.",
                                           dirs[:, None, :].expand(-1, N, -1).reshape(-1, 3), out[:, 1:], frame_code)
        w = bg_volume_weights(z_bg, out[:, :1])
        return (w[:, :, None] * rgb.reshape(-1, N, 3)).sum(1)

    def forward_eval(self, inp, hit_index, person_list=None):
        """inp: dict of torch tensors like Multiply.forward's input (batch dim 1).  hit_index[p]: sorted long tensor
        of the rays of person p's box (explicit input, see module docstring).  Returns the eval output dict plus
        intermediates for parity tests."""
        dirs, cam1 = get_camera_rays(inp["uv"][0], inp["pose"][0], inp["intrinsics"][0])
        R = dirs.shape[0]
        cam = cam1[None].expand(R, -1)
        scale = inp["smpl_params"][0, :, 0]
        beta = self.beta()
        persons = list(range(self.P)) if person_list is None else person_list
        z_l, zmax_l, sdf_l, rgb_l, nrm_l, hit_l, iters_l, xc_l = [], [], [], [], [], [], [], []
        for p in persons:
            so = self.servers[p].forward(scale[p], inp["smpl_trans"][0, p], inp["smpl_pose"][0, p],
                                         inp["smpl_shape"][0, p])
            tfs, pv = so["smpl_tfs"], so["smpl_verts"]
            cond = inp["smpl_pose"][0, p, 3:] / np.pi                              # multiply.py:270
            idx = hit_index[p]
            d, c = dirs[idx], cam[idx]
            person = self.persons[p]
            fn = lambda pts: person.sdf_func(pts, cond, tfs, pv, eval_mode=True)[0]
            z, iters = error_bound_sample(self.cfg, d, c, fn, beta)
            zmax, z = z[:, -1], z[:, :-1]
            pts = (c[:, None, :] + z[:, :, None] * d[:, None, :]).reshape(-1, 3)
            with torch.no_grad():
                sdf, x_c, _ = person.sdf_func(pts, cond, tfs, pv, eval_mode=True)
            rgb, nrm, _ = self.shade(person, x_c, cond, tfs)
            S = z.shape[1]
            z_l.append(z); zmax_l.append(zmax); sdf_l.append(sdf.reshape(-1, S)); hit_l.append(idx)
            rgb_l.append(rgb.detach().reshape(-1, S, 3)); nrm_l.append(nrm.detach().reshape(-1, S, 3))
            iters_l.append(iters); xc_l.append(x_c.reshape(-1, S, 3))
        with torch.no_grad():
            fg_rgb, nrm, acc, acc_person, bg_T = packed_composite(R, hit_l, z_l, zmax_l, sdf_l, rgb_l, nrm_l, beta,
                                                                  persons)
            if inp.get("idx", None) is not None:
                code = self.sd["frame_latent_encoder.weight"][int(inp["idx"][0])]
                bg_rgb = self.background(dirs, cam, code)
            else:
                bg_rgb = torch.ones_like(fg_rgb)
            rgb_values = fg_rgb + bg_T[:, None] * bg_rgb                             # multiply.py:544-545
            fg_out = fg_rgb + bg_T[:, None] * torch.ones_like(fg_rgb)                # multiply.py:590
        return dict(acc_map=acc, acc_person_list=acc_person, rgb_values=rgb_values, fg_rgb_values=fg_out,
                    normal_values=nrm, bg_transmittance=bg_T, bg_rgb=bg_rgb, z_vals=z_l, z_max=zmax_l, sdf=sdf_l,
                    rgb_samples=rgb_l, normal_samples=nrm_l, iters=iters_l, x_c=xc_l, ray_dirs=dirs, cam_loc=cam1)

    # ------------------------------------------------------------------------------------------- training forward
    def background_train(self, dirs, cam, frame_code, t_rand):
        """background branch with autograd enabled (multiply.py:514-539; depths jittered: ray_sampler.py:32-40)"""
        R = dirs.shape[0]
        z_bg = torch.flip(bg_depths(self.cfg, R, t_rand), dims=[-1])
        N = z_bg.shape[1]
        pts = depth2pts_outside(cam[:, None, :].expand(-1, N, -1), dirs[:, None, :].expand(-1, N, -1), z_bg,
                                self.cfg.radius).reshape(-1, 4)
        out = implicit_forward(self.sd, "bg_implicit_network.", pts, frame_code, multires=10)
        rgb = rendering_forward_nerf_frame(self.sd, "bg_rendering_network.",
                                           dirs[:, None, :].expand(-1, N, -1).reshape(-1, 3), out[:, 1:], frame_code)
        w = bg_volume_weights(z_bg, out[:, :1])
        return (w[:, :, None] * rgb.reshape(-1, N, 3)).sum(1)

    def forward_train(self, inp, hit_index, z_given, draws, cond_zero=False, person_list=None):
        """Training-mode Multiply.forward (multiply.py:254-588) from the sampler's output on: `z_given[k]` (R_p, N+N_extra+2)
        are the depths the sampler returned (it runs under no_grad, ray_sampler.py:86-87, so it is an input here);
        draws = {'person': {p: {eik_idx, eik_noise}}, 'bg_rand'}.  Every tensor of self.sd that requires grad receives
        gradients through the returned outputs (torch autograd, incl. the double backward through the normals and
        the eikonal term)."""
        dirs, cam1 = get_camera_rays(inp["uv"][0], inp["pose"][0], inp["intrinsics"][0])
        R = dirs.shape[0]
        cam = cam1[None].expand(R, -1)
        scale = inp["smpl_params"][0, :, 0]
        beta = self.beta()
        persons = list(range(self.P)) if person_list is None else person_list
        z_l, zmax_l, sdf_l, rgb_l, nrm_l, hit_l, gth_l = [], [], [], [], [], [], []
        body_grad = any(inp[k].requires_grad for k in ("smpl_pose", "smpl_trans", "smpl_shape"))
        smpl_surface_loss, zero_pose_loss = torch.zeros(1), torch.zeros(1)
        for k, p in enumerate(persons):
            with torch.set_grad_enabled(body_grad):
                so = self.servers[p].forward(scale[p], inp["smpl_trans"][0, p], inp["smpl_pose"][0, p],
                                             inp["smpl_shape"][0, p])
            tfs, pv = so["smpl_tfs"], so["smpl_verts"].detach()
            cond = inp["smpl_pose"][0, p, 3:] / np.pi
            if cond_zero:
                cond = cond * 0.0                                                     # multiply.py:271-273
            idx = hit_index[k]
            d, c = dirs[idx], cam[idx]
            person = self.persons[p]
            zz = z_given[k]
            zmax, z = zz[:, -1], zz[:, :-1]
            pts = (c[:, None, :] + z[:, :, None] * d[:, None, :]).reshape(-1, 3)
            with torch.set_grad_enabled(body_grad):   # weights are detached (deformer.py:47), the transforms are not
                x_c, _ = deform_inverse(pts, tfs, pv, person.server.weights)          # training: outliers keep their sdf
            rgb, nrm, sdf = self.shade(person, x_c, cond, tfs, create_graph=True)
            # eikonal samples (multiply.py:322-331)
            dr = draws["person"][p]
            xe = (person.server.verts_c[dr["eik_idx"]] + dr["eik_noise"] * 0.01).detach().requires_grad_(True)
            se = person.implicit(xe, cond)[:, :1]
            gth = torch.autograd.grad(se, xe, torch.ones_like(se), create_graph=True)[0]
            # the two optional regularisers (multiply.py:336-394; weight 0 in the shipped configs), when their draws are handed in
            if dr.get("surf_idx") is not None:
                # SDF at posed SMPL vertices (head / hands / feet excluded by the draw), warped to canonical space: above 0.02 is implausible
                sample = pv.reshape(-1, 3)[dr["surf_idx"].long()]
                xs, _ = deform_inverse(sample, tfs, pv, person.server.weights)
                ss = person.implicit(xs, cond)[:, 0]
                bad = ss > 0.02
                if bool(bad.any()):
                    smpl_surface_loss = smpl_surface_loss + F.l1_loss(ss[bad], torch.full_like(ss[bad], 0.02), reduction="mean")
            if draws.get("zp_idx") is not None:
                # network q under THIS person's conditioning vs under a zero conditioning, at vertices of q's canonical mesh
                for q in range(self.P):
                    vq = self.persons[q].server.verts_c[draws["zp_idx"][(p, q)].long()]
                    o_pred = self.persons[q].implicit(vq, cond)
                    o_zero = self.persons[q].implicit(vq, cond * 0.0)
                    zero_pose_loss = zero_pose_loss + F.l1_loss(o_pred[:, :1], o_zero[:, :1], reduction="mean") + \
                        F.l1_loss(o_pred[:, 1:], o_zero[:, 1:], reduction="mean")
            S = z.shape[1]
            z_l.append(z); zmax_l.append(zmax); sdf_l.append(sdf.reshape(-1, S)); hit_l.append(idx)
            rgb_l.append(rgb.reshape(-1, S, 3)); nrm_l.append(nrm.reshape(-1, S, 3)); gth_l.append(gth)
        fg_rgb, nrm, acc, acc_person, bg_T = packed_composite(R, hit_l, z_l, zmax_l, sdf_l, rgb_l, nrm_l, beta, persons)
        if inp.get("idx", None) is not None:
            code = self.sd["frame_latent_encoder.weight"][int(inp["idx"][0])]
            bg_rgb = self.background_train(dirs, cam, code, draws["bg_rand"])
        else:
            bg_rgb = torch.ones_like(fg_rgb)
        rgb_values = fg_rgb + bg_T[:, None] * bg_rgb
        return dict(rgb_values=rgb_values, normal_values=nrm, acc_map=acc, acc_person_list=acc_person,
                    grad_theta=torch.cat(gth_l, 0)[None], bg_transmittance=bg_T, bg_rgb=bg_rgb, sdf=sdf_l,
                    rgb_samples=rgb_l, smpl_surface_loss=smpl_surface_loss, zero_pose_loss=zero_pose_loss)
