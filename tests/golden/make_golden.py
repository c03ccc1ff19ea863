"""Generates tests/golden/*.npz by running the REFERENCE's own modules (build container only).

    python tests/golden/make_golden.py            # needs /root/reference, writes next to this file

The reference (eth-ait/MultiPly, /root/reference/code) hard-codes .cuda() and imports third-party
packages that are absent here, so the harness
  * registers empty stub modules for imageio, skimage, cv2, trimesh, kaolin, nerfacc, hydra, pytorch3d,
  * provides pytorch3d.ops.knn_points as an exact brute-force nearest-neighbour search,
  * turns Tensor.cuda / Module.cuda into the identity,
  * writes a SYNTHETIC SMPL pickle (multiply_amd.synthetic.make_smpl_tables, seed 0) where SMPLServer looks
    for the licensed model, and points hydra.utils.to_absolute_path at that temporary directory.
Only inputs and outputs (data) are stored; network weights are NOT stored: they are re-created from
torch.manual_seed(SEED) by constructing the networks in the reference's construction order
(multiply.py:53-66), and their checksums are stored so the consumer can prove it holds the same weights.
Nothing from /root/reference is copied into the repository.
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference/code"
SEED = 0


class AttrDict(dict):
    __getattr__ = dict.__getitem__

    def get(self, k, d=None):
        return self[k] if k in self else d


def install_stubs(tmp):
    def knn_points(p1, p2, K=1, return_nn=False, **kw):
        d2 = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1) if p1.shape[1] * p2.shape[1] < 4e7 else None
        if d2 is None:
            outs_d, outs_i = [], []
            for s in range(0, p1.shape[1], 4096):
                dd = ((p1[:, s:s + 4096, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
                d, i = torch.topk(dd, K, dim=-1, largest=False)
                outs_d.append(d); outs_i.append(i)
            d, i = torch.cat(outs_d, 1), torch.cat(outs_i, 1)
        else:
            d, i = torch.topk(d2, K, dim=-1, largest=False)
        nn_pts = torch.gather(p2[:, None].expand(-1, p1.shape[1], -1, -1), 2, i[..., None].expand(-1, -1, -1, 3))
        return d, i, nn_pts

    names = ["imageio", "skimage", "skimage.measure", "cv2", "trimesh", "kaolin", "kaolin.ops", "kaolin.ops.mesh",
             "kaolin.metrics", "kaolin.metrics.trianglemesh", "nerfacc", "hydra", "hydra.utils", "pytorch3d",
             "pytorch3d.ops"]
    for n in names:
        sys.modules[n] = types.ModuleType(n)
    sys.modules["hydra"].utils = sys.modules["hydra.utils"]
    sys.modules["hydra.utils"].to_absolute_path = lambda p: os.path.join(tmp, p)
    sys.modules["pytorch3d"].ops = sys.modules["pytorch3d.ops"]
    sys.modules["pytorch3d.ops"].knn_points = knn_points
    sys.modules["kaolin"].ops = sys.modules["kaolin.ops"]
    sys.modules["kaolin.ops"].mesh = sys.modules["kaolin.ops.mesh"]
    sys.modules["kaolin.ops.mesh"].index_vertices_by_faces = lambda v, f: None
    for n in ["render_weight_from_density", "pack_info", "accumulate_along_rays"]:
        setattr(sys.modules["nerfacc"], n, None)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def model_opts():
    """The shipped model config, code/confs/model/taichi01_model.yaml:17-76 (values only)."""
    imp = AttrDict(feature_vector_size=256, d_in=3, d_out=1, dims=[256] * 8, init="geometry", bias=0.6, skip_in=[4],
                   weight_norm=True, embedder_mode="fourier", multires=6, cond="smpl", number_person=2,
                   scene_bounding_sphere=3.0)
    ren = AttrDict(feature_vector_size=256, mode="pose_no_view", d_in=14, d_out=3, dims=[256] * 4, weight_norm=True,
                   multires_view=-1)
    bgi = AttrDict(feature_vector_size=256, d_in=4, d_out=1, dims=[256] * 8, init="none", bias=0.0, skip_in=[4],
                   weight_norm=False, embedder_mode="fourier", multires=10, cond="frame")
    bgr = AttrDict(feature_vector_size=256, mode="nerf_frame_encoding", d_in=3, d_out=3, dims=[128],
                   weight_norm=False, multires_view=4)
    samp = dict(near=0.0, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10,
                max_total_iters=5, N_samples_inverse_sphere=32, add_tiny=1.0e-6)
    return imp, ren, bgi, bgr, samp


def checksum(sd):
    tot = 0.0
    for k in sorted(sd):
        tot += float(sd[k].double().abs().sum()) + 3.0 * float(sd[k].double().sum())
    return tot


def main():
    tmp = tempfile.mkdtemp(prefix="mp_golden_")
    install_stubs(tmp)
    from multiply_amd.synthetic import make_smpl_tables, make_scene
    tables = make_smpl_tables(0)
    os.makedirs(os.path.join(tmp, "lib/smpl/smpl_model"), exist_ok=True)
    for g in ["MALE", "FEMALE", "NEUTRAL"]:
        with open(os.path.join(tmp, f"lib/smpl/smpl_model/SMPL_{g}.pkl"), "wb") as f:
            pickle.dump(tables, f)
    sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    from lib.model.networks import ImplicitNet, RenderingNet
    from lib.model.density import LaplaceDensity, AbsDensity
    from lib.model.deformer import SMPLDeformer
    from lib.model.smpl import SMPLServer
    from lib.model.ray_sampler import ErrorBoundSampler
    from lib.model.multiply import Multiply
    from lib.model.loss import Loss
    from lib.utils import rend_util

    imp_o, ren_o, bgi_o, bgr_o, samp_o = model_opts()
    P = 2
    scene = make_scene(P, seed=0, H=64, W=64)
    betas = scene["smpl_params"][0, :, 76:]

    # ---- hand-assembled Multiply (its __init__ needs trimesh/kaolin assets), same construction order
    torch.manual_seed(SEED)
    m = Multiply.__new__(Multiply)
    torch.nn.Module.__init__(m)
    m.use_person_encoder = False
    m.foreground_implicit_network_list = torch.nn.ModuleList()
    m.foreground_rendering_network_list = torch.nn.ModuleList()
    for p in range(P):
        m.foreground_implicit_network_list.append(ImplicitNet(imp_o))
        m.foreground_rendering_network_list.append(RenderingNet(ren_o))
    m.with_bkgd = True
    m.bg_implicit_network = ImplicitNet(bgi_o)
    m.bg_rendering_network = RenderingNet(bgr_o)
    m.frame_latent_encoder = torch.nn.Embedding(75, 32)
    m.deformer_list = torch.nn.ModuleList([SMPLDeformer(betas=betas[p], gender="male") for p in range(P)])
    m.sdf_bounding_sphere = 3.0
    m.density = LaplaceDensity(params_init={"beta": 0.1}, beta_min=0.0001)
    m.bg_density = AbsDensity()
    m.ray_sampler = ErrorBoundSampler(3.0, inverse_sphere_bg=True, **samp_o)
    m.smpl_server_list = torch.nn.ModuleList([SMPLServer(gender="male", betas=betas[p]) for p in range(P)])
    m.eval()
    sd = {k: v.detach() for k, v in m.state_dict().items() if "smpl" not in k.split(".")[0] and "deformer" not in k}
    G = {"weights_checksum": np.array(checksum(sd)), "state_keys": np.array(sorted(sd.keys()))}
    print("weights checksum", G["weights_checksum"], len(sd), "tensors")
    rng = np.random.RandomState(1)
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)

    # ---- G1 camera rays (rend_util.get_camera_params)
    uv, pose, K = f32(scene["uv"]), f32(scene["pose"]), f32(scene["intrinsics"])
    Ksk = K.clone(); Ksk[0, 0, 1] = 0.7
    dirs, cam = rend_util.get_camera_params(uv, pose, Ksk)
    G.update(g1_uv=uv.numpy(), g1_pose=pose.numpy(), g1_K=Ksk.numpy(), g1_dirs=dirs.numpy(), g1_cam=cam.numpy())

    # ---- G2 SMPLServer.forward for three poses
    sp = f32(scene["smpl_params"])
    poses = [torch.zeros(72), None, sp[0, 0, 4:76]]
    th_a = torch.zeros(72); th_a[5] = np.pi / 6; th_a[8] = -np.pi / 6
    poses[1] = th_a
    for i, th in enumerate(poses):
        with torch.no_grad():
            o = m.smpl_server_list[0](sp[:, 0, 0] * (1.0 + 0.1 * i), sp[:, 0, 1:4], th[None], sp[:, 0, 76:])
        G[f"g2_thetas{i}"] = th.numpy()
        for k in ["smpl_verts", "smpl_tfs", "smpl_jnts", "smpl_weights"]:
            if k == "smpl_weights":
                continue
            G[f"g2_{k}{i}"] = o[k].numpy()[0].astype(np.float32)
    G["g2_scale_trans_betas"] = sp[0, 0].numpy()
    G["g2_verts_c"] = m.smpl_server_list[0].verts_c[0].numpy()
    G["g2_tfs_c_inv"] = m.smpl_server_list[0].tfs_c_inv.numpy()

    # ---- G3 deformer
    with torch.no_grad():
        so = [m.smpl_server_list[p](sp[:, p, 0], sp[:, p, 1:4], sp[:, p, 4:76], sp[:, p, 76:]) for p in range(P)]
    pv = so[0]["smpl_verts"]
    pick = rng.randint(0, 6890, 4096)
    x = pv[0, pick] + f32(rng.normal(0, 0.06, (4096, 3)))
    with torch.no_grad():
        x_c, outl = m.deformer_list[0].forward(x, so[0]["smpl_tfs"], return_weights=False, inverse=True, smpl_verts=pv)
        x_d = m.deformer_list[0].forward_skinning(x_c[None], None, so[0]["smpl_tfs"])[0]
    G.update(g3_x=x.numpy(), g3_xc=x_c.numpy(), g3_outlier=outl.numpy(), g3_xd=x_d.numpy())

    # ---- G4 networks
    cond = {"smpl": sp[:, 0, 7:76] / np.pi}
    xin = f32(rng.uniform(-0.8, 0.8, (512, 3)))
    net = m.foreground_implicit_network_list[0]
    with torch.no_grad():
        out = net(xin, cond)[0]
    # d sdf / d x by autograd on the reference network (ImplicitNet.gradient, networks.py:210-220, slices the
    # POINT axis by mistake and is not on the hot path; multiply.py:653-659 differentiates the sdf column)
    xg = xin.clone().requires_grad_(True)
    grad = torch.autograd.grad(net(xg, cond)[0][:, 0].sum(), xg)[0].detach()
    G.update(g4_x=xin.numpy(), g4_cond=cond["smpl"].numpy()[0], g4_imp=out.numpy(), g4_grad=grad.numpy())
    nrm = torch.nn.functional.normalize(f32(rng.normal(0, 1, (512, 3))), dim=1)
    with torch.no_grad():
        rgb = m.foreground_rendering_network_list[0](xin, nrm, None, cond["smpl"], out[:, 1:])
    G.update(g4_nrm=nrm.numpy(), g4_rgb=rgb.numpy())
    x4 = f32(rng.uniform(-1, 1, (256, 4)))
    code = m.frame_latent_encoder(torch.tensor([3]))
    vd = torch.nn.functional.normalize(f32(rng.normal(0, 1, (256, 3))), dim=1)
    with torch.no_grad():
        bo = m.bg_implicit_network(x4, {"frame": code})[0]
        brgb = m.bg_rendering_network(None, None, vd, None, bo[:, 1:], code)
    G.update(g4_bg_x=x4.numpy(), g4_bg_view=vd.numpy(), g4_bg_imp=bo.numpy(), g4_bg_rgb=brgb.numpy())

    # ---- G5 density
    s5 = f32(np.linspace(-0.5, 0.5, 1001))
    for i, b in enumerate([0.1, 0.01, 1e-3]):
        G[f"g5_sigma{i}"] = m.density(s5, beta=torch.tensor(b)).detach().numpy()
    G["g5_sdf"] = s5.numpy()

    # ---- G6 ErrorBoundSampler.get_z_vals (eval) on rays through the two bodies
    dirs, cam = rend_util.get_camera_params(uv, pose, K)
    dirs = dirs[0]; camr = cam.repeat(dirs.shape[0], 1)
    # rays around the image centre columns where the bodies are: take a strided subset of 96 rays
    sel = torch.arange(0, dirs.shape[0], dirs.shape[0] // 96)[:96]
    G["g6_sel"] = sel.numpy()
    for p in range(P):
        condp = {"smpl": sp[:, p, 7:76] / np.pi}
        with torch.no_grad():
            (z, z_bg), _ = m.ray_sampler.get_z_vals(dirs[sel], camr[sel], m, condp, so[p]["smpl_tfs"], eval_mode=True,
                                                     smpl_verts=so[p]["smpl_verts"], person_id=p)
        G[f"g6_z{p}"] = z.numpy(); G[f"g6_zbg{p}"] = z_bg.numpy()
        m.eval()

    # ---- G7 forward_gradient / get_rbg_value ; G8 compositing pieces; G10 assembled 1-person eval render
    torch.set_grad_enabled(True)
    p = 0
    condp = {"smpl": sp[:, p, 7:76] / np.pi}
    z = f32(G["g6_z0"]); z_max = z[:, -1]; zz = z[:, :-1]
    d_s, c_s = dirs[sel], camr[sel]
    pts = (c_s[:, None] + zz[:, :, None] * d_s[:, None]).reshape(-1, 3)
    sdf, xc, feat = m.sdf_func_with_smpl_deformer(pts, condp, so[p]["smpl_tfs"], smpl_verts=so[p]["smpl_verts"], person_id=p)
    sdf = sdf.detach()
    view = -d_s[:, None].repeat(1, zz.shape[1], 1).reshape(-1, 3)
    rgbf, others = m.get_rbg_value(pts, xc.detach().clone(), view, condp, so[p]["smpl_tfs"], feature_vectors=feat,
                                   person_id=p, is_training=False)
    rgbf, nrmf = rgbf.detach(), others["normals"].detach()
    G.update(g7_pts=pts.numpy(), g7_sdf=sdf.numpy(), g7_xc=xc.detach().numpy(), g7_rgb=rgbf.numpy(), g7_nrm=nrmf.numpy())
    with torch.no_grad():
        w, bgT = m.volume_rendering(zz, z_max, sdf)
        fg = (w[:, :, None] * rgbf.reshape(-1, zz.shape[1], 3)).sum(1)
        nv = (w[:, :, None] * nrmf.reshape(-1, zz.shape[1], 3)).sum(1)
        G.update(g8_w=w.numpy(), g8_bgT=bgT.numpy(), g8_fg=fg.numpy(), g8_nrm=nv.numpy(), g8_acc=w.sum(-1).numpy())
        zbg = torch.flip(f32(G["g6_zbg0"]), dims=[-1])
        N = zbg.shape[1]
        bp = m.depth2pts_outside(c_s[:, None].repeat(1, N, 1), d_s[:, None].repeat(1, N, 1), zbg)
        code = m.frame_latent_encoder(torch.tensor([5]))
        bo = m.bg_implicit_network(bp.reshape(-1, 4), {"frame": code})[0]
        brgb = m.bg_rendering_network(None, None, d_s[:, None].repeat(1, N, 1).reshape(-1, 3), None, bo[:, 1:], code)
        bw = m.bg_volume_rendering(zbg, bo[:, :1])
        bgv = (bw[:, :, None] * brgb.reshape(-1, N, 3)).sum(1)
        G.update(g8_bg_pts=bp.numpy(), g8_bg_w=bw.numpy(), g8_bg_rgb=bgv.numpy())
        G["g10_rgb_dense"] = (fg + bgT[:, None] * bgv).numpy()

    # ---- G9 Loss.forward on a synthetic training output dict
    lopt = AttrDict(eikonal_weight=0.1, bce_weight=5.0e-3, opacity_sparse_weight=3.0e-3, in_shape_weight=1.0e-2,
                    sam_mask_weight=3.0e-2, smpl_surface_milestone=800, sam_start_epoch=50, depth_order_weight=0.1,
                    silhouette_weight=0.0, interpenetration_loss_weight=0.005, zero_pose_weight=0.0)
    loss = Loss(lopt)
    Rr = 256
    mo = dict(fg_rgb_values_each_person_list=[], rgb_values=f32(rng.uniform(0, 1, (Rr, 3))),
              grad_theta=f32(rng.normal(0, 1, (1, 1024, 3))), acc_map=f32(rng.uniform(0.01, 0.99, Rr)),
              index_in_surface=torch.tensor(rng.uniform(0, 1, Rr) > 0.5), index_off_surface=None, epoch=120,
              temporal_loss=f32([0.0123]), smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1),
              sam_mask=f32(rng.normal(0, 4, (Rr, 2))), acc_person_list=f32(rng.uniform(0, 1, (Rr, 2))))
    gt = dict(rgb=f32(rng.uniform(0, 1, (1, Rr, 3))))
    lo = loss(mo, gt)
    for k in ["rgb_values", "grad_theta", "acc_map", "index_in_surface", "temporal_loss", "sam_mask", "acc_person_list"]:
        G["g9_in_" + k] = mo[k].numpy()
    G["g9_gt_rgb"] = gt["rgb"].numpy()
    for k, v in lo.items():
        G["g9_out_" + k] = np.asarray(v.detach().numpy(), dtype=np.float32).reshape(-1)

    G["scene_smpl_params"] = scene["smpl_params"]
    out = os.path.join(HERE, "reference_eval.npz")
    np.savez_compressed(out, **G)
    print("wrote", out, os.path.getsize(out) / 1e6, "MB")


if __name__ == "__main__":
    main()
