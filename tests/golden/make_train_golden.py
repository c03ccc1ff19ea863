"""Generates tests/golden/reference_train.npz and tests/golden/reference_state_keys.npz by running the REFERENCE's own
modules in TRAINING mode (build container only; same stub harness as make_golden.py):

    python tests/golden/make_train_golden.py

What is pinned (SURVEY.md §8c "training-mode fixtures additionally record the random draws"):
  T1  ErrorBoundSampler.get_z_vals with model.training = True (ray_sampler.py:21-42, 66-220): every torch.rand /
      randperm / randint it consumes is RECORDED in call order and stored next to its outputs (z_vals, z_vals_bg), so
      the oracle can be fed the identical randomness;
  T2  the per-sample training arithmetic for those depths: sdf_func_with_smpl_deformer (no outlier override in training),
      the eikonal points (randperm + PointInSpace.get_points, multiply.py:322-331, sampler.py:84-108) and their
      gradient(create_graph), get_rbg_value / forward_gradient with create_graph = True (multiply.py:600-661);
  T3  a training loss assembled from the reference's OWN pieces -- dense volume_rendering (multiply.py:663-680), the
      background branch with jittered inverse depths, Loss.forward (loss.py:108-177) -- and torch.autograd.grad of that
      loss w.r.t. EVERY parameter: stored as per-tensor L2 norm + projection on a seeded random direction (full tensors for
      a few small ones).  (The packed nerfacc compositing of multiply.py:425-480 is third party and stays unpinned; the
      dense path is the reference's own restatement of it.)
  keys the full state-dict key list (+ shapes, dtypes) of the module tree incl. the SMPL buffers.
Only data is stored; no reference source text.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG   # noqa: E402


class DrawRecorder:
    """Wraps torch.rand / randn_like / randperm / randint for the duration of a `with` block and logs what they return."""

    def __init__(self):
        self.log = []

    def __enter__(self):
        self.saved = (torch.rand, torch.randn_like, torch.randperm, torch.randint)
        rec = self

        def wrap(name, fn):
            def f(*a, **k):
                out = fn(*a, **k)
                rec.log.append((name, out.detach().clone()))
                return out
            return f
        torch.rand, torch.randn_like = wrap("rand", self.saved[0]), wrap("randn_like", self.saved[1])
        torch.randperm, torch.randint = wrap("randperm", self.saved[2]), wrap("randint", self.saved[3])
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn_like, torch.randperm, torch.randint = self.saved


def main():
    import tempfile
    import pickle
    tmp = tempfile.mkdtemp(prefix="mp_golden_")
    MG.install_stubs(tmp)
    from multiply_amd.synthetic import make_smpl_tables, make_scene
    tables = make_smpl_tables(0)
    os.makedirs(os.path.join(tmp, "lib/smpl/smpl_model"), exist_ok=True)
    for g in ["MALE", "FEMALE", "NEUTRAL"]:
        with open(os.path.join(tmp, f"lib/smpl/smpl_model/SMPL_{g}.pkl"), "wb") as f:
            pickle.dump(tables, f)
    sys.path.insert(0, MG.REF)
    import warnings
    warnings.filterwarnings("ignore")
    from lib.model.networks import ImplicitNet, RenderingNet
    from lib.model.density import LaplaceDensity, AbsDensity
    from lib.model.deformer import SMPLDeformer
    from lib.model.smpl import SMPLServer
    from lib.model.ray_sampler import ErrorBoundSampler
    from lib.model.sampler import PointInSpace
    from lib.model import multiply as ref_multiply
    from lib.model.multiply import Multiply
    from lib.model.loss import Loss
    from lib.utils import rend_util

    imp_o, ren_o, bgi_o, bgr_o, samp_o = MG.model_opts()
    P = 2
    scene = make_scene(P, seed=0, H=64, W=64)
    betas = scene["smpl_params"][0, :, 76:]
    torch.manual_seed(MG.SEED)
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
    m.sampler = PointInSpace()

    # ---- full state-dict contract
    sd = m.state_dict()
    keys = sorted(sd.keys())
    np.savez_compressed(os.path.join(HERE, "reference_state_keys.npz"), keys=np.array(keys),
                        shapes=np.array([",".join(map(str, sd[k].shape)) for k in keys]),
                        dtypes=np.array([str(sd[k].dtype) for k in keys]),
                        param_keys=np.array(sorted(k for k, _ in m.named_parameters())))
    print(len(keys), "state-dict keys,", sum(1 for _ in m.named_parameters()), "parameters")

    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    G = {"weights_checksum": np.array(MG.checksum({k: v.detach() for k, v in sd.items() if "smpl" not in k.split(".")[0]
                                                    and "deformer" not in k}))}
    uv, pose, K = f32(scene["uv"]), f32(scene["pose"]), f32(scene["intrinsics"])
    sp = f32(scene["smpl_params"])
    with torch.no_grad():
        so = [m.smpl_server_list[p](sp[:, p, 0], sp[:, p, 1:4], sp[:, p, 4:76], sp[:, p, 76:]) for p in range(P)]
    dirs, cam = rend_util.get_camera_params(uv, pose, K)
    dirs = dirs[0]
    camr = cam.repeat(dirs.shape[0], 1)
    sel = torch.arange(0, dirs.shape[0], dirs.shape[0] // 40)[:40]          # 40 rays across the image (both bodies, sky)
    G["sel"] = sel.numpy()
    G["scene_smpl_params"] = scene["smpl_params"]
    p = 0
    condp = {"smpl": sp[:, p, 7:76] / np.pi}                                   # epoch 301: conditioning on (multiply.py:268-272)
    d_s, c_s = dirs[sel], camr[sel]

    # ---- T1: the sampler in training mode, draws recorded in call order
    m.train()
    torch.manual_seed(1234)
    with DrawRecorder() as rec, torch.no_grad():
        (z, z_bg), z_eik = m.ray_sampler.get_z_vals(d_s, c_s, m, condp, so[p]["smpl_tfs"], eval_mode=True,
                                                     smpl_verts=so[p]["smpl_verts"], person_id=p)
    m.train()
    names = [n for n, _ in rec.log]
    print("sampler draws:", [(n, tuple(t.shape)) for n, t in rec.log])
    # call order (ray_sampler.py): rand t_rand (R,128) [UniformSampler :38]; rand u (R,64) [final inverse CDF :171]; randperm
    # [extra samples :202]; randint [eikonal depth :211, unused by the caller]; rand (R,32) [inverse-sphere sampler :38]
    assert names == ["rand", "rand", "randperm", "randint", "rand"], names
    G.update(t1_t_rand=rec.log[0][1].numpy(), t1_u_final=rec.log[1][1].numpy(), t1_perm=rec.log[2][1].numpy(),
             t1_bg_rand=rec.log[4][1].numpy(), t1_z=z.numpy(), t1_zbg=z_bg.numpy())

    # ---- T2: per-sample training arithmetic
    torch.set_grad_enabled(True)
    z_max, zz = z[:, -1], z[:, :-1]
    S = zz.shape[1]
    pts = (c_s[:, None] + zz[:, :, None] * d_s[:, None]).reshape(-1, 3)
    sdf, xc, feat = m.sdf_func_with_smpl_deformer(pts, condp, so[p]["smpl_tfs"], smpl_verts=so[p]["smpl_verts"], person_id=p)
    with DrawRecorder() as rec2:
        verts_c = m.smpl_server_list[p].verts_c.repeat(1, 1, 1)
        indices = torch.randperm(verts_c.shape[1])[:512]
        vsel = torch.index_select(verts_c, 1, indices)
        sample = m.sampler.get_points(vsel, global_ratio=0.)
    print("eikonal draws:", [(n, tuple(t.shape)) for n, t in rec2.log])
    assert [n for n, _ in rec2.log] == ["randperm", "randn_like", "rand"]       # rand: the empty global part (0 points)
    G.update(t2_eik_perm=rec2.log[0][1].numpy(), t2_eik_noise=rec2.log[1][1].numpy()[0], t2_eik_points=sample.detach().numpy()[0])
    sample.requires_grad_()
    local_pred = m.foreground_implicit_network_list[p](sample, condp, person_id=p)[..., 0:1]
    grad_theta = ref_multiply.gradient(sample, local_pred)
    view = -d_s[:, None].repeat(1, S, 1).reshape(-1, 3)
    rgbf, others = m.get_rbg_value(pts, xc, view, condp, so[p]["smpl_tfs"], feature_vectors=feat, person_id=p, is_training=True)
    nrmf = others["normals"]
    G.update(t2_sdf=sdf.detach().numpy(), t2_xc=xc.detach().numpy(), t2_rgb=rgbf.detach().numpy(), t2_nrm=nrmf.detach().numpy(),
             t2_grad_theta=grad_theta.detach().numpy()[0])

    # ---- T3: loss from the reference's own dense path + autograd w.r.t. every parameter
    w, bgT = m.volume_rendering(zz, z_max, sdf)
    fg = (w[:, :, None] * rgbf.reshape(-1, S, 3)).sum(1)
    nv = (w[:, :, None] * nrmf.reshape(-1, S, 3)).sum(1)
    acc = w.sum(-1)
    zbg = torch.flip(z_bg, dims=[-1])                                           # multiply.py:516
    N = zbg.shape[1]
    bp = m.depth2pts_outside(c_s[:, None].repeat(1, N, 1), d_s[:, None].repeat(1, N, 1), zbg)
    code = m.frame_latent_encoder(torch.tensor([5]))
    bo = m.bg_implicit_network(bp.reshape(-1, 4), {"frame": code})[0]
    brgb = m.bg_rendering_network(None, None, d_s[:, None].repeat(1, N, 1).reshape(-1, 3), None, bo[:, 1:], code)
    bw = m.bg_volume_rendering(zbg, bo[:, :1])
    bgv = (bw[:, :, None] * brgb.reshape(-1, N, 3)).sum(1)
    rgb_values = fg + bgT[:, None] * bgv
    lopt = MG.AttrDict(eikonal_weight=0.1, bce_weight=5.0e-3, opacity_sparse_weight=3.0e-3, in_shape_weight=1.0e-2,
                       sam_mask_weight=3.0e-2, smpl_surface_milestone=800, sam_start_epoch=50, depth_order_weight=0.1,
                       silhouette_weight=0.0, interpenetration_loss_weight=0.005, zero_pose_weight=0.0)
    loss_fn = Loss(lopt)
    rng = np.random.RandomState(7)
    R = len(sel)
    gt = dict(rgb=f32(rng.uniform(0, 1, (1, R, 3))))
    sam = f32(rng.normal(0, 4, (R, 1)))
    mo = dict(fg_rgb_values_each_person_list=[], rgb_values=rgb_values, grad_theta=grad_theta, acc_map=acc,
              index_in_surface=None, index_off_surface=None, epoch=301, temporal_loss=torch.zeros(1),
              smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1), sam_mask=sam, acc_person_list=acc[:, None])
    lo = loss_fn(mo, gt)
    G.update(t3_fg=fg.detach().numpy(), t3_nrm=nv.detach().numpy(), t3_acc=acc.detach().numpy(), t3_bgT=bgT.detach().numpy(),
             t3_bg_rgb=bgv.detach().numpy(), t3_rgb_values=rgb_values.detach().numpy(), t3_gt_rgb=gt["rgb"].numpy(),
             t3_sam=sam.numpy())
    for k, v in lo.items():
        G["t3_loss_" + k] = np.asarray(v.detach().numpy(), dtype=np.float32).reshape(-1)
    named = [(k, v) for k, v in m.named_parameters() if v.requires_grad]
    grads = torch.autograd.grad(lo["loss"], [v for _, v in named], allow_unused=True)
    gk, gn, gp = [], [], []
    prj = torch.Generator().manual_seed(99)
    for (k, v), g in zip(named, grads):
        if g is None:
            continue
        r = torch.randn(v.shape, generator=prj)
        gk.append(k); gn.append(float(g.double().norm())); gp.append(float((g.double() * r.double()).sum()))
        if g.numel() <= 300:
            G["t3_grad_full_" + k] = g.numpy()
    G.update(t3_grad_keys=np.array(gk), t3_grad_norm=np.array(gn), t3_grad_proj=np.array(gp))
    print(len(gk), "parameter tensors receive a gradient; loss", float(lo["loss"]))
    out = os.path.join(HERE, "reference_train.npz")
    np.savez_compressed(out, **G)
    print("wrote", out, os.path.getsize(out) / 1e6, "MB")


if __name__ == "__main__":
    main()
