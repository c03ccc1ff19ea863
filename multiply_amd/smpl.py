"""SMPL body model on the device: table loading and the SMPLServer module API.

Mirrors the constructor / attribute / forward contract of the reference's SMPLServer (code/lib/model/smpl.py:6-94) and
of the SMPL class it wraps (code/lib/smpl/body_models.py:60-365) for what the hot path and its callers read:
`verts_c`, `joints_c`, `tfs_c_inv`, `faces`, `smpl.faces`, `bone_parents`, `param_canonical`, and
forward(scale, transl, thetas, betas, absolute=False) -> {'smpl_verts','smpl_tfs','smpl_jnts','smpl_all_jnts',
'smpl_weights'}; the SMPL sub-module's parameters / buffers keep the reference's state-dict names.
The arithmetic (blend shapes, Rodrigues, kinematic chain, LBS) runs in csrc/geom.hip (mp_smpl_pose).
"""
import os
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import hip

NUM_VERTS, NUM_JOINTS = 6890, 24


def _to_np(a, dtype=np.float32):
    if "scipy.sparse" in str(type(a)):
        a = a.todense()
    if hasattr(a, "r"):  # chumpy arrays of the licensed pickle
        a = a.r
    return np.array(a, dtype=dtype)


def load_smpl_tables(gender="neutral", model_dir=None):
    """Reads SMPL_{GENDER}.pkl from `model_dir`, $MULTIPLY_SMPL_DIR or ./lib/smpl/smpl_model (the reference's location,
    smpl.py:13); falls back to the seeded synthetic tables only when MULTIPLY_SYNTHETIC_SMPL=1 is set explicitly."""
    dirs = [model_dir, os.environ.get("MULTIPLY_SMPL_DIR"), os.path.join(os.getcwd(), "lib/smpl/smpl_model")]
    for d in dirs:
        if d and os.path.exists(os.path.join(d, f"SMPL_{str(gender).upper()}.pkl")):
            with open(os.path.join(d, f"SMPL_{str(gender).upper()}.pkl"), "rb") as f:
                return pickle.load(f, encoding="latin1")
    if os.environ.get("MULTIPLY_SYNTHETIC_SMPL") == "1":
        from .synthetic import make_smpl_tables
        return make_smpl_tables(0)
    raise FileNotFoundError("SMPL model file not found (set MULTIPLY_SMPL_DIR, or MULTIPLY_SYNTHETIC_SMPL=1 for the "
                            "synthetic body used by the benchmarks)")


class SMPLDeviceTables:
    """fp32 device copies of the model tables in the layouts the kernels read."""

    def __init__(self, tables, device):
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.v_template = f(_to_np(tables["v_template"]))                                   # (V,3)
        self.shapedirs = f(_to_np(np.asarray(tables["shapedirs"])[:, :, :10]))               # (V,3,10)
        pd = _to_np(tables["posedirs"])
        self.posedirs = f(np.ascontiguousarray(pd.reshape(-1, pd.shape[-1]).T))              # (207, 3V)
        self.j_regressor = f(_to_np(tables["J_regressor"]))                                  # (24,V)
        self.lbs_weights = f(_to_np(tables["weights"]))                                      # (V,24)
        parents = _to_np(tables["kintree_table"], np.float64)[0].astype(np.int64)
        parents[0] = -1
        self.parents_np = parents
        self.parents = f(parents.astype(np.int32))
        self.faces = np.asarray(tables["f"]).astype(np.int64)
        # d rest-joints / d betas (J = J_regressor (v_template + shapedirs betas)): a constant table, used by the
        # pose/shape backward (mp_smpl_pose_bwd)
        self.j_shapedirs = torch.einsum("jv,vkl->jkl", self.j_regressor, self.shapedirs).contiguous()   # (24,3,10)
        assert self.v_template.shape == (NUM_VERTS, 3) and self.lbs_weights.shape == (NUM_VERTS, NUM_JOINTS)


def knn_cluster_perm(verts_np):
    """Static clustering for the exact nearest-vertex search: vertices are split kd-tree style into leaves of exactly 2 KNN_CLUSTER
    (the last one partial) so that clusters stay spatially compact under posing, and every leaf once more into two halves of at most
    KNN_CLUSTER (32): the fine clusters the eval searches scan; the leaves themselves -- the pairs (2c, 2c + 1) -- are the coarse
    clusters of the training searches (csrc/geom.hip k_knn_build).  Returns int32 [KNN_NC * KNN_CLUSTER], -1 = padding."""
    CL = hip.KNN_CLUSTER
    leaves = []

    def split(idx):
        pts = verts_np[idx]
        axis = int(np.argmax(pts.max(0) - pts.min(0)))
        order = idx[np.argsort(pts[:, axis], kind="stable")]
        if len(idx) <= 2 * CL:
            leaves.append(order[:CL])
            leaves.append(order[CL:])
            return
        nleaf = -(-len(idx) // (2 * CL))
        nl = (nleaf // 2) * 2 * CL
        split(order[:nl])
        split(order[nl:])

    split(np.arange(verts_np.shape[0]))
    assert len(leaves) <= hip.KNN_NC and hip.KNN_NC % 2 == 0
    perm = -np.ones(hip.KNN_NC * CL, dtype=np.int32)
    for c, l in enumerate(leaves):
        perm[c * CL:c * CL + len(l)] = l
    return perm


FACE_KEYPOINT_VERTS = (332, 6260, 2800, 4071, 583)   # nose, reye, leye, rear, lear (lib/smpl/vertex_ids.py 'smplh')


class _VertexJointSelector(nn.Module):
    """lib/smpl/vertex_joint_selector.py with use_hands=False, use_feet_keypoints=False (smpl.py:12-17): the five face
    keypoints appended to the 24 kinematic joints."""

    def __init__(self, device):
        super().__init__()
        self.register_buffer("extra_joints_idxs", torch.tensor(FACE_KEYPOINT_VERTS, dtype=torch.long, device=device))


class _SMPLModule(nn.Module):
    """Stands in for `smpl_server.smpl` (lib/smpl/body_models.py SMPL): callers read `.faces` / `.bone_parents`, and its
    registered parameters / buffers are part of the checkpoint contract -- the reference's state dict carries
    `smpl_server_list.N.smpl.{betas,global_orient,body_pose,transl,faces_tensor,v_template,shapedirs,J_regressor,posedirs,
    parents,lbs_weights}` and `...smpl.vertex_joint_selector.extra_joints_idxs` (body_models.py:152-249), the same again
    under `deformer_list.N.smpl.smpl.`.  The buffers ALIAS the device tables the kernels read (same shapes and layouts as
    the reference's), so a loaded checkpoint's tables are the ones used."""

    def __init__(self, tables):
        super().__init__()
        dev = tables.v_template.device
        self.faces = tables.faces
        self.bone_parents = tables.parents_np.copy()
        self.vertex_joint_selector = _VertexJointSelector(dev)
        self.register_buffer("faces_tensor", torch.from_numpy(tables.faces).to(dev))
        z = lambda n: nn.Parameter(torch.zeros(1, n, dtype=torch.float32, device=dev), requires_grad=True)
        self.betas, self.global_orient, self.body_pose, self.transl = z(10), z(3), z(69), z(3)   # body_models.py:158-211
        self.register_buffer("v_template", tables.v_template)
        self.register_buffer("shapedirs", tables.shapedirs)
        self.register_buffer("J_regressor", tables.j_regressor)
        self.register_buffer("posedirs", tables.posedirs)
        self.register_buffer("parents", torch.from_numpy(tables.parents_np.copy()).to(dev))          # long, parents[0] = -1
        self.register_buffer("lbs_weights", tables.lbs_weights)
        self._register_load_state_dict_pre_hook(self._check_tables)

    _TABLE_KEYS = ("v_template", "shapedirs", "J_regressor", "posedirs", "lbs_weights", "parents", "faces_tensor")

    def _check_tables(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Everything derived from the tables at construction (canonical vertices and inverse bone transforms, the joint
        shape basis, the nearest-vertex cluster order, the canonical meshes) is NOT rebuilt by load_state_dict: a
        checkpoint whose body-model tables differ from the ones this model was built with (another gender, synthetic
        tables) is refused here instead of leaving the model silently inconsistent."""
        for k in self._TABLE_KEYS:
            v = state_dict.get(prefix + k)
            if v is None:
                continue
            cur = getattr(self, k)
            same = tuple(v.shape) == tuple(cur.shape) and bool(
                torch.equal(v.to(cur.device).to(cur.dtype), cur) if not cur.is_floating_point()
                else torch.allclose(v.to(cur.device).to(cur.dtype), cur, rtol=0, atol=1e-7))
            if not same:
                error_msgs.append(f"{prefix}{k}: the checkpoint's SMPL table differs from the one this model was constructed with "
                                  f"(gender / betas_path / smpl_tables); construct the model with the checkpoint's body model")


class _PoseTfs(torch.autograd.Function):
    """smpl_tfs (24,4,4) of the 86 parameters with the hand-written adjoint of csrc/geom.hip (mp_smpl_pose_bwd)"""

    @staticmethod
    def forward(ctx, server, p, verts, jnts):
        prm = p.detach().contiguous()
        tfs = torch.empty(NUM_JOINTS, 4, 4, dtype=torch.float32, device=prm.device)
        server.pose_into(prm, verts, tfs, jnts)
        ctx.server, ctx.prm, ctx.rest = server, prm, server.rest_joints().contiguous()
        return tfs

    @staticmethod
    def backward(ctx, dtfs):
        sv, t = ctx.server, ctx.server.tables
        dprm = torch.empty(86, dtype=torch.float32, device=ctx.prm.device)
        hip.check(hip.lib().mp_smpl_pose_bwd(hip.ptr(t.parents), hip.ptr(ctx.prm), hip.ptr(sv.tfs_c_inv), hip.ptr(ctx.rest),
                                             hip.ptr(t.j_shapedirs), hip.ptr(dtfs.reshape(24, 16).float().contiguous()),
                                             hip.ptr(dprm), hip.stream()), "mp_smpl_pose_bwd")
        return None, dprm, None, None


class SMPLServer(nn.Module):
    def __init__(self, gender="neutral", betas=None, v_template=None, smpl_tables=None, device=None):
        super().__init__()
        if v_template is not None:
            raise NotImplementedError("custom v_template is not used by the shipped configs")
        hip.require_device()
        device = torch.device(device or "cuda")
        raw = smpl_tables if smpl_tables is not None else load_smpl_tables(gender)
        self.tables = raw if isinstance(raw, SMPLDeviceTables) else SMPLDeviceTables(raw, device)
        self.smpl = _SMPLModule(self.tables)
        self.faces = self.tables.faces
        self.bone_parents = self.tables.parents_np.astype(int)
        self.bone_ids = [[int(self.bone_parents[i]), i] for i in range(NUM_JOINTS)]
        self.v_template = None
        self.betas = None if betas is None else torch.tensor(np.asarray(betas), dtype=torch.float32, device=device)
        pc = torch.zeros(1, 86, dtype=torch.float32, device=device)
        pc[0, 0] = 1
        pc[0, 9] = np.pi / 6
        pc[0, 12] = -np.pi / 6
        if self.betas is not None:
            pc[0, -10:] = self.betas
        self.param_canonical = pc
        self._work = torch.empty(3 * NUM_VERTS + 1024, dtype=torch.float32, device=device)
        self.tfs_c_inv = None
        out = self.forward(*torch.split(pc, [1, 3, 72, 10], dim=1), absolute=True)
        self.verts_c = out["smpl_verts"]
        self.joints_c = out["smpl_jnts"]
        self.tfs_c_inv = torch.linalg.inv(out["smpl_tfs"].squeeze(0)).contiguous()   # init-time only (smpl.py:47)

    def pose_into(self, params86, verts, tfs, joints, absolute=False):
        """raw launch: params86 (86,) device -> preallocated verts (V,3), tfs (24,4,4), joints (24,3)"""
        t = self.tables
        hip.check(hip.lib().mp_smpl_pose(hip.ptr(t.v_template), hip.ptr(t.shapedirs), hip.ptr(t.posedirs),
                                         hip.ptr(t.j_regressor), hip.ptr(t.lbs_weights), hip.ptr(t.parents),
                                         hip.ptr(params86), None if absolute else hip.ptr(self.tfs_c_inv), hip.ptr(verts),
                                         hip.ptr(tfs), hip.ptr(joints), hip.ptr(self._work), hip.stream()),
                  "mp_smpl_pose")

    def rest_joints(self):
        """J (24,3) of the most recent pose_into() call (kernel work buffer, csrc/geom.hip W_J)"""
        return self._work[3 * NUM_VERTS:3 * NUM_VERTS + 3 * NUM_JOINTS].clone()

    def forward(self, scale, transl, thetas, betas, absolute=False):
        """smpl.py:50-94.  `smpl_tfs` carries gradients to scale / transl / thetas / betas when any of them requires grad
        (the mesh-space losses back-propagate into BodyModelParams through it, multiply_model.py:586-620, 969-974; betas
        through the rest joints, as in the training step); vertices and joints are returned detached."""
        dev = self.param_canonical.device
        p = torch.cat([scale.reshape(1, 1), transl.reshape(1, 3), thetas.reshape(1, 72), betas.reshape(1, 10)], 1)
        p = p.float().reshape(86)
        verts = torch.empty(NUM_VERTS, 3, dtype=torch.float32, device=dev)
        jnts = torch.empty(NUM_JOINTS, 3, dtype=torch.float32, device=dev)
        if torch.is_grad_enabled() and p.requires_grad and not absolute:
            tfs = _PoseTfs.apply(self, p, verts, jnts)
        else:
            tfs = torch.empty(NUM_JOINTS, 4, 4, dtype=torch.float32, device=dev)
            self.pose_into(p.detach().contiguous(), verts, tfs, jnts, absolute=absolute)
        # smpl_all_jnts: the 24 kinematic joints + the face keypoint vertices (body_models.py:345; smpl.py:83-84) -- the
        # vertices are already scaled / translated exactly like the joints (smpl.py:77-84)
        all_jnts = torch.cat([jnts, verts[self.smpl.vertex_joint_selector.extra_joints_idxs]], 0)
        return {"smpl_verts": verts[None], "smpl_jnts": jnts[None], "smpl_all_jnts": all_jnts[None], "smpl_tfs": tfs[None],
                "smpl_weights": self.tables.lbs_weights[None]}
