"""Synthetic SMPL tables and synthetic scenes (no licensed assets needed).

The licensed SMPL model (`lib/smpl/smpl_model/SMPL_{MALE,FEMALE,NEUTRAL}.pkl`, reference
README.md:17-23) is not redistributable, so benchmarks, fixtures and tests use a seeded
synthetic body with exactly the same table names and shapes the reference's SMPL class
reads (code/lib/smpl/body_models.py:186-225): v_template (6890,3), shapedirs (6890,3,10),
posedirs (6890,3,207), J_regressor (24,6890), kintree_table (2,24), weights (6890,24),
f (13776,3).  Everything is a deterministic function of the seed (numpy RandomState), so the
tables are regenerated on the GPU box instead of being shipped.
"""
import numpy as np

SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
                        dtype=np.int64)
NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_FACES = 13776

# rough T-pose rest joints (metres, y up), left = +x
_REST_JOINTS = np.array([
    [0.00, -0.24, 0.02], [0.07, -0.33, 0.01], [-0.07, -0.33, 0.01], [0.00, -0.12, -0.01],
    [0.10, -0.71, 0.01], [-0.10, -0.71, 0.01], [0.00, 0.02, 0.00], [0.09, -1.11, -0.03],
    [-0.09, -1.11, -0.03], [0.00, 0.08, 0.02], [0.11, -1.17, 0.09], [-0.11, -1.17, 0.09],
    [0.00, 0.29, -0.01], [0.08, 0.19, -0.01], [-0.08, 0.19, -0.01], [0.00, 0.38, 0.03],
    [0.18, 0.23, -0.02], [-0.18, 0.23, -0.02], [0.44, 0.22, -0.04], [-0.44, 0.22, -0.04],
    [0.69, 0.23, -0.04], [-0.69, 0.23, -0.04], [0.78, 0.22, -0.05], [-0.78, 0.22, -0.05]], dtype=np.float64)

# capsule radius around the bone that ends in joint j (bone = parent(j) -> j); joint 0 gets a pelvis blob
_BONE_RADIUS = np.array([0.13, 0.10, 0.10, 0.13, 0.075, 0.075, 0.135, 0.05, 0.05, 0.14, 0.04, 0.04, 0.065, 0.08, 0.08,
                         0.09, 0.06, 0.06, 0.045, 0.045, 0.035, 0.035, 0.03, 0.03], dtype=np.float64)


def make_smpl_tables(seed=0):
    """Returns a dict with the SMPL pickle keys (float64 / int64 numpy arrays)."""
    rng = np.random.RandomState(seed)
    J = _REST_JOINTS.copy()
    # vertices: points on capsule surfaces around every bone, count proportional to capsule area
    seg_a = np.where(SMPL_PARENTS[:, None] >= 0, J[np.maximum(SMPL_PARENTS, 0)], J - np.array([0, 0.05, 0]))
    seg_b = J.copy()
    seg_b[15] = J[15] + np.array([0.0, 0.12, 0.0])  # head blob extends above the head joint
    length = np.linalg.norm(seg_b - seg_a, axis=1)
    area = 2 * np.pi * _BONE_RADIUS * (length + 2 * _BONE_RADIUS)
    counts = np.floor(area / area.sum() * NUM_VERTS).astype(int)
    counts[0] += NUM_VERTS - counts.sum()
    verts = []
    for j in range(NUM_JOINTS):
        n = counts[j]
        a, b, r = seg_a[j], seg_b[j], _BONE_RADIUS[j]
        axis = (b - a) / max(length[j], 1e-9)
        # orthonormal frame
        tmp = np.array([1.0, 0, 0]) if abs(axis[0]) < 0.9 else np.array([0, 1.0, 0])
        u = np.cross(axis, tmp); u /= np.linalg.norm(u)
        v = np.cross(axis, u)
        t = rng.uniform(-r, length[j] + r, size=n)          # position along the capsule incl. caps
        ang = rng.uniform(0, 2 * np.pi, size=n)
        tc = np.clip(t, 0, length[j])
        over = t - tc                                        # signed overshoot into the caps
        rad = np.sqrt(np.maximum(r * r - over * over, 0.0))
        p = a + np.outer(tc + over, axis) + np.outer(rad * np.cos(ang), u) + np.outer(rad * np.sin(ang), v)
        verts.append(p)
    v_template = np.concatenate(verts, 0)
    assert v_template.shape == (NUM_VERTS, 3)
    perm = rng.permutation(NUM_VERTS)                        # SMPL vertex order is not spatially sorted either
    v_template = v_template[perm]

    # skinning weights: soft assignment to the 4 nearest bones
    d2 = np.empty((NUM_VERTS, NUM_JOINTS))
    for j in range(NUM_JOINTS):
        ab = seg_b[j] - seg_a[j]
        tt = np.clip(((v_template - seg_a[j]) @ ab) / max(ab @ ab, 1e-12), 0, 1)
        d2[:, j] = ((v_template - (seg_a[j] + np.outer(tt, ab))) ** 2).sum(1)
    w = np.exp(-d2 / (2 * 0.05 ** 2))
    kth = np.sort(w, axis=1)[:, -4][:, None]
    w = np.where(w >= kth, w, 0.0)
    weights = w / w.sum(1, keepdims=True)

    # joint regressor: minimum-norm affine combination of the 64 nearest vertices reproducing each rest joint
    J_regressor = np.zeros((NUM_JOINTS, NUM_VERTS))
    for j in range(NUM_JOINTS):
        idx = np.argsort(((v_template - J[j]) ** 2).sum(1))[:64]
        A = np.concatenate([v_template[idx].T, np.ones((1, 64))], 0)   # 4 x 64
        bvec = np.concatenate([J[j], [1.0]])
        J_regressor[j, idx] = A.T @ np.linalg.solve(A @ A.T + 1e-9 * np.eye(4), bvec)

    # smooth shape / pose blend shapes
    def smooth_field(n_fields, amp):
        freq = rng.normal(0, 3.0, size=(n_fields, 3))
        phase = rng.uniform(0, 2 * np.pi, size=(n_fields,))
        direction = rng.normal(0, 1.0, size=(n_fields, 3))
        direction /= np.linalg.norm(direction, axis=1, keepdims=True)
        s = np.sin(v_template @ freq.T + phase)              # (V, n_fields)
        return amp * s[:, None, :] * direction.T[None, :, :]  # (V, 3, n_fields)

    shapedirs = smooth_field(10, 0.015)
    posedirs = smooth_field(207, 0.003)

    # faces: only an attribute of the hot path (`smpl.faces`); near-neighbour triangles
    order = np.argsort(v_template[:, 1] * 7.0 + v_template[:, 0])
    f = np.stack([order[np.arange(NUM_FACES) % NUM_VERTS], order[(np.arange(NUM_FACES) + 1) % NUM_VERTS],
                  order[(np.arange(NUM_FACES) + 2) % NUM_VERTS]], 1).astype(np.int64)

    kintree = np.stack([np.where(SMPL_PARENTS < 0, 4294967295, SMPL_PARENTS), np.arange(NUM_JOINTS)], 0)
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor,
                kintree_table=kintree.astype(np.int64), weights=weights, f=f)


def make_scene(num_person=2, seed=0, H=512, W=512):
    """Seeded synthetic multi-person scene: SMPL params per person and a pinhole camera.

    Returns numpy arrays shaped like the reference's dataset items (code/lib/datasets/Hi4D.py:273-306):
      smpl_params (1,P,86) = [scale, trans3, pose72, betas10], intrinsics (1,4,4), pose (1,4,4), uv (1,H*W,2).
    """
    params = np.zeros((1, num_person, 86), dtype=np.float32)
    for p in range(num_person):
        rng = np.random.RandomState(1000 * seed + p)
        betas = rng.normal(0, 1, 10) * 0.5
        body = rng.normal(0, 0.2, 69)
        orient = np.array([np.pi, 0, 0]) + rng.normal(0, 0.1, 3)
        x = (p - (num_person - 1) / 2.0) * 0.6
        trans = np.array([x, 0.15, 0.0]) + rng.normal(0, 0.02, 3)
        params[0, p, 0] = 1.0
        params[0, p, 1:4] = trans
        params[0, p, 4:7] = orient
        params[0, p, 7:76] = body
        params[0, p, 76:] = betas
    intr = np.eye(4, dtype=np.float32)
    intr[0, 0] = intr[1, 1] = 1.5 * W
    intr[0, 2] = W / 2.0
    intr[1, 2] = H / 2.0
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 3] = [0.0, 0.0, -2.5]
    uv = np.mgrid[:H, :W].astype(np.int32)
    uv = np.flip(uv, axis=0).copy().reshape(2, -1).T.astype(np.float32)  # (x, y) order as Hi4D.py:254-255
    return dict(smpl_params=params, intrinsics=intr[None], pose=pose[None], uv=uv[None])


def write_sequence(root, n_frames=4, H=96, W=128, num_person=2, seed=0, with_edges=False):
    """Writes a small synthetic sequence in the reference's on-disk scene format (the layout `Hi4DDataset` reads,
    code/lib/datasets/Hi4D.py:92-130, written by preprocessing/preprocessing_multiple_trace.py:529-599):
    image/%04d.png, mask/<p>/%04d.png, [edge/%04d.png,] poses.npy, mean_shape.npy, normalize_trans.npy,
    cameras_normalize.npz (scale_mat_i, world_mat_i), gender.npy.  Returns the arrays it wrote."""
    import os
    from PIL import Image
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "image"), exist_ok=True)
    sc = make_scene(num_person, seed=seed, H=H, W=W)
    K, pose = sc["intrinsics"][0].astype(np.float64), sc["pose"][0].astype(np.float64)
    R, C = pose[:3, :3].T, pose[:3, 3]
    world = np.eye(4)
    world[:3, :4] = K[:3, :3] @ np.concatenate([R, (-R @ C)[:, None]], 1)
    cams, images, masks = {}, [], []
    yy, xx = np.mgrid[:H, :W]
    for f in range(n_frames):
        img = (rs.rand(H // 8 + 1, W // 8 + 1, 3) * 255).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:H, :W]
        img = (img.astype(np.int32) + rs.randint(-20, 20, (H, W, 3))).clip(0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, "image", "%04d.png" % f))
        images.append(img)
        per = []
        for p in range(num_person):
            cx, cy = W * (0.3 + 0.4 * p / max(num_person - 1, 1)) + 3 * f, H * 0.5
            m = (((xx - cx) / (0.16 * W)) ** 2 + ((yy - cy) / (0.38 * H)) ** 2) < 1.0
            os.makedirs(os.path.join(root, "mask", "%d" % p), exist_ok=True)
            Image.fromarray((m[:, :, None] * np.array([255, 255, 255])).astype(np.uint8)).save(
                os.path.join(root, "mask", "%d" % p, "%04d.png" % f))
            per.append(m)
        masks.append(np.stack(per))
        if with_edges:
            os.makedirs(os.path.join(root, "edge"), exist_ok=True)
            any_m = np.any(per, axis=0)
            edge = any_m & ~(np.roll(any_m, 2, 0) & np.roll(any_m, -2, 0) & np.roll(any_m, 2, 1) & np.roll(any_m, -2, 1))
            Image.fromarray((edge[:, :, None] * np.array([255, 255, 255])).astype(np.uint8)).save(
                os.path.join(root, "edge", "%04d.png" % f))
        s = np.eye(4)
        s[:3, :3] *= 1.0 + 0.0 * f
        cams["scale_mat_%d" % f] = s
        cams["world_mat_%d" % f] = world
    poses = np.repeat(sc["smpl_params"][0, None, :, 4:76], n_frames, 0) + rs.normal(0, 0.01, (n_frames, num_person, 72))
    trans = np.repeat(sc["smpl_params"][0, None, :, 1:4], n_frames, 0)
    shape = sc["smpl_params"][0, :, 76:]
    np.save(os.path.join(root, "poses.npy"), poses)
    np.save(os.path.join(root, "normalize_trans.npy"), trans)
    np.save(os.path.join(root, "mean_shape.npy"), shape)
    np.save(os.path.join(root, "gender.npy"), np.array(["male"] * num_person))
    np.savez(os.path.join(root, "cameras_normalize.npz"), **cams)
    return dict(images=np.stack(images), masks=np.stack(masks), poses=poses, trans=trans, shape=shape, intrinsics=K, pose=pose)
