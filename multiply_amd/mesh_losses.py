"""Mesh-space losses and the instance-mask refresh on top of the per-person z-buffers: the arithmetic of
multiply_model.py:578-736 (get_depth_order_loss), :521-551 (get_interpenetration_loss), :741-939 (get_instance_mask, the
part between the frame's camera and the mask / key-point arrays) and :969-974 (the deformed canonical mesh), with the file
dumps, tqdm loops and trainer state left to the caller.

Device work: canonical mesh extraction (mesh.py / csrc/mise.hip), skinning weights of the mesh vertices (mp_query_weights),
the z-buffers (csrc/raster.hip), the inside test of the interpenetration term (mp_mesh_signed_distance: the sign of the
signed distance stands in for kaolin.ops.mesh.check_sign).  Everything after those is small torch arithmetic on (H, W)
maps that autograd differentiates: depth -> vertices -> blended bone transforms -> SMPLServer (_PoseTfs) -> pose.
"""
import numpy as np
import torch

from . import hip
from .render import get_renderer

MAX_DEPTH = 999.0
DEPTH_LOSS_MILESTONE = 1000
# multiply_model.py:596: instance colours of the persons (RGB 0..255), background appended as black (:627-630)
COLOR_DICT = [[255, 0.0, 0.0], [0.0, 255, 0.0], [0.0, 0.0, 255], [125, 125, 0.0], [0.0, 125, 125], [125, 0.0, 125],
              [64, 0.0, 0.0], [0.0, 64, 0.0], [0.0, 0.0, 64], [32, 32, 0.0], [0.0, 32, 32], [32, 0.0, 32]]


def skinning(x, w, tfs, inverse=False):
    """deformer.py:72-89 in torch (differentiable in x, w and tfs): x (B,N,3), w (B,N,J), tfs (B,J,4,4) -> (B,N,3)"""
    x_h = torch.nn.functional.pad(x, (0, 1), value=1.0)
    if inverse:
        w_tf = torch.einsum("bpn,bnij->bpij", w, tfs)
        x_h = torch.einsum("bpij,bpj->bpi", w_tf.inverse(), x_h)
    else:
        x_h = torch.einsum("bpn,bnij,bpj->bpi", w, tfs, x_h)
    return x_h[:, :, :3]


def deformed_mesh_vertices(model, verts_c, smpl_tfs, person):
    """multiply_model.py:969-974: canonical mesh vertices (1,N,3) -> posed space with the skinning weights of the nearest
    canonical SMPL vertices (constants) and the bone transforms (differentiable)."""
    w = model.deformer_list[person].query_weights(verts_c[0].detach())
    return skinning(verts_c, w, smpl_tfs.reshape(1, 24, 4, 4))


def front_depth(depth_maps):
    """multiply_model.py:640-652: depth_maps list of (H,W), -1 = empty -> (H,W,P) with 999 for empty, its min over persons"""
    mx = torch.stack([torch.where(d < 0, torch.full_like(d, MAX_DEPTH), d) for d in depth_maps], dim=-1)
    return mx, mx.min(dim=-1).values


def instance_masks(depth_maps):
    """multiply_model.py:893-898: (P,H,W) bool, person p is the front-most surface of the pixel"""
    _, front = front_depth(depth_maps)
    return torch.stack([d == front for d in depth_maps], 0)


def depth_order_loss(depth_maps, org_sam_mask, epoch, depth_order_weight=0.005):
    """multiply_model.py:653-736 (the loss itself): pixels whose SAM label names a person that the meshes put BEHIND
    another one are pushed with log(1 + exp(z_labelled - z_front)).  org_sam_mask (1,H,W,P) or (H,W,P) raw logits."""
    mx, front = front_depth(depth_maps)
    valid = front < MAX_DEPTH
    sam = torch.sigmoid(org_sam_mask).reshape(*front.shape, -1)
    ssum = sam.sum(dim=-1)
    valid = valid & (ssum <= 1 + 1e-2) & (ssum >= 0.7)
    gt = torch.gather(mx, -1, sam.argmax(dim=-1, keepdim=True)).squeeze(-1)
    valid = valid & (gt < MAX_DEPTH)
    use = valid & ~(gt == front)
    # the reference returns a fresh 0.0 when nothing is out of order (:726-727); the masked sum is that same zero without a
    # host round trip
    loss = torch.where(use, torch.log(1 + torch.exp(torch.where(use, gt - front, torch.zeros_like(gt)))),
                       torch.zeros_like(gt)).sum()
    return depth_order_weight * (1 - min(DEPTH_LOSS_MILESTONE, epoch) / DEPTH_LOSS_MILESTONE) * loss


def gt_instance_map(org_sam_mask, n_person):
    """multiply_model.py:656-666: (H,W,3) colour-coded arg-max of [sigmoid(sam), 1 - sum] (the silhouette target)"""
    sam = torch.sigmoid(org_sam_mask).reshape(*org_sam_mask.shape[-3:])
    fb = torch.cat([sam, 1 - sam.sum(dim=-1, keepdim=True)], dim=-1)
    colors = torch.tensor(COLOR_DICT[:n_person] + [[0.0, 0.0, 0.0]], device=sam.device)
    return colors[fb.argmax(dim=-1)]


def _nearest(points, verts, chunk=2048):
    out = []
    for i in range(0, points.shape[0], chunk):
        out.append(torch.cdist(points[i:i + chunk], verts).argmin(dim=1))
    return torch.cat(out) if out else torch.zeros(0, dtype=torch.long, device=points.device)


def interpenetration_loss(vertex_list, face_list, num_points=5120, draws=None):
    """multiply_model.py:521-551: `num_points` random vertices of every mesh that lie INSIDE another person's mesh are
    pulled to that mesh's nearest vertex (summed squared distance), outliers further than 0.1 ignored.
    vertex_list[p] (1,N,3), face_list[p] (1,F,3); draws[p] = the vertex ids to use instead of torch.randperm."""
    dev = vertex_list[0].device
    total = torch.zeros(1, device=dev)
    L = hip.lib()
    for pid, vertex in enumerate(vertex_list):
        idx = (torch.randperm(vertex.shape[1])[:num_points] if draws is None else draws[pid]).to(dev)
        sample = torch.index_select(vertex, 1, idx)                                # (1,n,3)
        pts = sample[0].detach().float().contiguous()
        for qid, partner in enumerate(vertex_list):
            if qid == pid:
                continue
            fv = partner[0].detach()[face_list[qid].reshape(-1, 3).long()].float().contiguous()     # (F,3,3)
            sd = torch.empty(pts.shape[0], dtype=torch.float32, device=dev)
            hip.check(L.mp_mesh_signed_distance(hip.ptr(pts), pts.shape[0], hip.ptr(fv), fv.shape[0], hip.ptr(sd),
                                                hip.stream()), "mp_mesh_signed_distance")
            inside = sd < 0
            pen = sample[0][inside]
            nn_pts = partner[0][_nearest(pen.detach(), partner[0].detach())]
            stable = (pen - nn_pts).norm(dim=-1) < 0.1
            d = torch.where(stable[:, None], pen - nn_pts, torch.zeros_like(pen))
            total = total + (d * d).sum()
    return total


def posed_meshes(model, inputs, use_smpl_mesh=False, res_up=2, meshes=None):
    """The per-person meshes of one frame in the renderer's units (divided by the SMPL scale): the SMPL surface
    (multiply_model.py:823-829, epochs <= 190) or the person's canonical zero level set, posed (:586-620, :831-848).
    `meshes` = already extracted canonical meshes (mesh.canonical_mesh) to reuse.  -> verts (1,N,3) list, faces (1,F,3)
    list, SMPL outputs list."""
    from .mesh import canonical_mesh
    sp = inputs["smpl_params"]
    pose, shape, trans = inputs["smpl_pose"], inputs["smpl_shape"], inputs["smpl_trans"]
    vs, fs, outs = [], [], []
    for p, server in enumerate(model.smpl_server_list):
        scale = sp[:, p, 0]
        out = server(scale, trans[:, p], pose[:, p], shape[:, p])
        outs.append(out)
        if use_smpl_mesh:
            v = out["smpl_verts"]
            f = torch.from_numpy(np.ascontiguousarray(server.smpl.faces.astype(np.int64))).to(v.device)[None]
        else:
            m = meshes[p] if meshes is not None else canonical_mesh(model, p, cond=pose[0, p, 3:] / np.pi, res_up=res_up)
            v = deformed_mesh_vertices(model, m["vertices"][None], out["smpl_tfs"], p)
            f = m["faces"][None]
        vs.append((1 / scale.squeeze()) * v)
        fs.append(f)
    return vs, fs, outs


def get_depth_order_loss(model, inputs, epoch, loss_opt=None, meshes=None, draws=None):
    """multiply_model.py:578-736 -> (depth-order loss, silhouette loss, interpenetration loss).  inputs: P, smpl_params,
    smpl_pose / smpl_shape / smpl_trans (these may require grad), img_size, org_sam_mask."""
    loss_opt = loss_opt or {}
    renderer = get_renderer(inputs)
    vs, fs, _ = posed_meshes(model, inputs, meshes=meshes)
    depth = [d[0, :, :, 0] for d in renderer.render_multiple_depth_map(vs, fs)]
    fade = 1 - min(DEPTH_LOSS_MILESTONE, epoch) / DEPTH_LOSS_MILESTONE
    inter = loss_opt.get("interpenetration_loss_weight", 0.0) * fade * interpenetration_loss(vs, fs, draws=draws)
    sil_w = loss_opt.get("silhouette_weight", 0.0)
    sil = torch.zeros((), device=depth[0].device)
    if sil_w != 0.0:                       # multiply_model.py:618-637, :656-668, :721 (the reference renders it even at weight 0)
        cols = [torch.tensor(COLOR_DICT[p], device=v.device).repeat(v.shape[1], 1)[None] / 255.0 for p, v in enumerate(vs)]
        rmap = 255 * renderer.softrender_multiple_meshes(vs, fs, cols)[0]
        rgb = rmap[..., :3] * (rmap[..., [3]] / 255.0)
        sil = sil_w * torch.nn.functional.mse_loss(gt_instance_map(inputs["org_sam_mask"], len(vs)), rgb) * fade
    order = depth_order_loss(depth, inputs["org_sam_mask"], epoch, loss_opt.get("depth_order_weight", 0.005))
    return order, sil, inter


def frame_instance_masks(model, inputs, use_smpl_mesh, res_up=2):
    """One frame of get_instance_mask (multiply_model.py:790-898, :860-870): the per-person z-buffers of the SMPL meshes
    (epochs <= 190) or of the posed canonical meshes -> instance masks (P,H,W) bool, depth maps, and the 27 projected key
    points per person (P,27,2) int32 that the SAM prompts are built from."""
    with torch.no_grad():
        renderer = get_renderer(inputs)
        vs, fs, outs = posed_meshes(model, inputs, use_smpl_mesh=use_smpl_mesh, res_up=res_up)
        depth = [d[0, :, :, 0] for d in renderer.render_multiple_depth_map(vs, fs)]
        Pm = inputs["P"][0].double()
        kps = []
        for out in outs:
            j = out["smpl_all_jnts"][0, :27].double()
            t = torch.cat([j, torch.ones_like(j[:, :1])], 1) @ Pm.t()
            kps.append((t[:, :2] / t[:, 2:3]).to(torch.int32))                      # astype(np.int32): truncation
        return instance_masks(depth), depth, torch.stack(kps, 0)


def body_model_inputs(body_model_list, frame_idx):
    """multiply_model.py:162-168 / 264-274: the per-frame rows of every person's BodyModelParams stacked into the model's
    smpl_trans (1,P,3), smpl_shape (1,P,10), smpl_pose (1,P,72) inputs (they carry the embeddings' gradients)."""
    rows = [bm(frame_idx) for bm in body_model_list]
    trans = torch.stack([r["transl"] for r in rows], dim=1)
    shape = torch.stack([r["betas"] for r in rows], dim=1)
    pose = torch.cat((torch.stack([r["global_orient"] for r in rows], dim=1),
                      torch.stack([r["body_pose"] for r in rows], dim=1)), dim=2)
    return trans, shape, pose


def opt_depth_frame(model, body_model_list, loss_fn, inputs, sample_fn, epoch, it_per_loop, lr, loss_opt=None, depth_pose=False,
                    depth_cond_zero=False, res_up=2):
    """One frame of the trainer's depth-refinement stage (multiply_model.py:230-486): the persons' canonical meshes are
    extracted once, then `it_per_loop` Adam steps on the frame's body-model rows -- the translations only, or every body
    parameter with depth_pose -- minimise  render loss (a 512-ray training forward of the scene model) + depth-order loss
    + interpenetration loss  of the meshes re-posed with the current rows.

    inputs: the frame's test item on the device (P, C, intrinsics, pose, smpl_params, idx, img_size, org_sam_mask), batch
    dimension 1.  sample_fn() -> (dict with uv (1,n,2), index_outside, sam_mask (1,n,P); dict with rgb (1,n,3)): the frame's
    pixel samples of one iteration (the reference draws them with weighted_sampling; datasets.SceneStore.sample does the
    same from the resident frame).  Returns the per-iteration {'render_loss','depth_order_loss','interpenetration_loss'}."""
    from .mesh import canonical_mesh
    loss_opt = loss_opt or {}
    params = list(p for bm in body_model_list for p in bm.parameters()) if depth_pose else [bm.transl.weight for bm in body_model_list]
    saved = [p.requires_grad for p in params]
    for p in params:
        p.requires_grad_(True)
    opt = torch.optim.Adam([{"params": params, "lr": lr}], lr=lr, eps=1e-8)
    frame = inputs["idx"].reshape(-1).long()
    scale = inputs["smpl_params"][:, :, 0]
    renderer = get_renderer(inputs)
    with torch.no_grad():
        _, _, pose0 = body_model_inputs(body_model_list, frame)
        meshes = []
        for p in range(len(model.smpl_server_list)):
            cond = pose0[0, p, 3:] * 0.0 if depth_cond_zero else pose0[0, p, 3:] / np.pi
            meshes.append(canonical_mesh(model, p, cond=cond, res_up=res_up))
    faces = [m["faces"][None] for m in meshes]
    fade = 1 - min(DEPTH_LOSS_MILESTONE, epoch) / DEPTH_LOSS_MILESTONE
    was_training = model.training
    model.train()
    history = []
    for _ in range(it_per_loop):
        smp_in, smp_tg = sample_fn()
        opt.zero_grad()
        trans, shape, pose = body_model_inputs(body_model_list, frame)
        minp = dict(smpl_trans=trans, smpl_shape=shape, smpl_pose=pose, smpl_pose_last=pose,       # temporal loss disabled (:361)
                    idx=inputs["idx"], P=inputs["P"], C=inputs["C"], intrinsics=inputs["intrinsics"], pose=inputs["pose"],
                    smpl_params=inputs["smpl_params"], current_epoch=epoch, **smp_in)
        out = model(minp, cond_zero_shit=True) if depth_cond_zero else model(minp)
        render = loss_fn(out, smp_tg)["loss"]
        verts = []
        for p, server in enumerate(model.smpl_server_list):
            so = server(scale[:, p], trans[:, p], pose[:, p], shape[:, p])
            verts.append((1 / scale[:, p].squeeze()) * deformed_mesh_vertices(model, meshes[p]["vertices"][None], so["smpl_tfs"], p))
        depth = [d[0, :, :, 0] for d in renderer.render_multiple_depth_map(verts, faces)]
        inter = loss_opt.get("interpenetration_loss_weight", 0.0) * fade * interpenetration_loss(verts, faces)
        order = depth_order_loss(depth, inputs["org_sam_mask"], epoch, loss_opt.get("depth_order_weight", 0.005))
        (inter.sum() + order + render.sum()).backward()
        opt.step()
        history.append({"render_loss": render.detach().reshape(()), "depth_order_loss": order.detach().reshape(()),
                        "interpenetration_loss": inter.detach().reshape(())})
    for p, r in zip(params, saved):
        p.requires_grad_(r)
    model.train(was_training)
    return history
