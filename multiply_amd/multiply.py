"""Multiply scene model: the reference's nn.Module API over the HIP hot path.

Drop-in for `lib.model.multiply.Multiply` (reference code/lib/model/multiply.py:23-598):
  * constructor Multiply(opt, betas_path) building the same sub-module tree, in the same order and with the same
    state-dict names (multiply.py:35-100) -- including the SMPL tables under smpl_server_list.N.smpl.* and
    deformer_list.N.smpl.smpl.* -- so the reference's checkpoints load with strict=True (tests/test_state_dict_gpu.py) and
    seeded initialisation is bit-identical;
  * forward(input, id=-1, cond_zero_shit=False, canonical_pose=False) -> the reference's output dict
    (multiply.py:566-597);
  * the attributes the Lightning module reaches into.
Everything between the input dict and the output dict runs in hand-written HIP kernels (multiply_amd/csrc) through
the C ABI of include/multiply_hip.h.  There is no PyTorch / CPU fallback: without the library or a gfx950 device the
forward raises.

Extensions that do not exist in the reference (all optional, defaults reproduce the reference):
  * input['hit_index'] : list of per-person ascending ray-id tensors replacing the box cull (parity tests; the
    reference's trimesh box, multiply.py:208-214, 256-266, is third-party code);
  * self.convergence_group : number of consecutive rays that share the sampler's convergence vote
    (ray_sampler.py:137).  None = the whole call, exactly like the reference; set it to pixel_per_batch to render a
    whole frame in one call with the results of the reference's chunked loop (multiply_model.py:1051-1055).
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .density import AbsDensity, LaplaceDensity
from .networks import ImplicitNet, RenderingNet
from .ray_sampler import ErrorBoundSampler
from .smpl import NUM_JOINTS, NUM_VERTS, SMPLDeviceTables, SMPLServer, knn_cluster_perm, load_smpl_tables
from .deformer import SMPLDeformer
from .sampler import PointInSpace


class Multiply(nn.Module):
    def __init__(self, opt, betas_path, smpl_tables=None, gender_list=None):
        super().__init__()
        hip.require_device()
        betas = np.load(betas_path) if isinstance(betas_path, str) else np.asarray(betas_path)
        self.using_nerfacc = True
        self.smpl_surface_weight = opt.loss.get("smpl_surface_weight", 0)
        self.zero_pose_weight = opt.loss.get("zero_pose_weight", 0)
        self.smpl_vertex_part = None
        if self.smpl_surface_weight > 0:      # multiply.py:112-113 (the reference's asset ./outputs/smpl_vert_segmentation.json)
            import json
            seg = os.path.abspath(opt.get("smpl_vert_segmentation_path", "./outputs/smpl_vert_segmentation.json"))
            if os.path.exists(seg):
                with open(seg) as f:
                    self.smpl_vertex_part = json.load(f)
            # (absent: train.surface_sampling_weights raises at the first training forward unless model.smpl_vertex_part is assigned)
        self.use_person_encoder = opt.get("use_person_encoder", False)
        if self.use_person_encoder:
            raise NotImplementedError("use_person_encoder (shared triplane networks) is outside the hot-path scope")
        betas2 = betas.reshape(-1, 10) if betas.ndim == 2 else betas.reshape(1, 10)
        self.num_person = betas2.shape[0]

        # same construction order as the reference (multiply.py:35-66): RNG consumption is identical
        self.foreground_implicit_network_list = nn.ModuleList()
        self.foreground_rendering_network_list = nn.ModuleList()
        for _ in range(self.num_person):
            self.foreground_implicit_network_list.append(ImplicitNet(opt.implicit_network))
            self.foreground_rendering_network_list.append(RenderingNet(opt.rendering_network))
        self.with_bkgd = opt.with_bkgd
        self.bg_implicit_network = ImplicitNet(opt.bg_implicit_network)
        self.bg_rendering_network = RenderingNet(opt.bg_rendering_network)
        self.frame_latent_encoder = nn.Embedding(opt.num_training_frames, opt.dim_frame_encoding)
        self.sampler = PointInSpace()
        self.use_smpl_deformer = opt.use_smpl_deformer
        if not self.use_smpl_deformer:
            raise NotImplementedError("only the SMPL deformer branch exists in the reference's shipped configs")

        if gender_list is None:
            # multiply.py:71: gender.npy next to mean_shape.npy.  Without it the reference fails; a silent default would
            # pick the wrong body model -- only the explicit synthetic-table route (tests, benchmarks) has no genders.
            gpath = betas_path[:-14] + "gender.npy" if isinstance(betas_path, str) else None
            if gpath and os.path.exists(gpath):
                gender_list = np.load(gpath)
            elif smpl_tables is not None:
                gender_list = ["male"] * self.num_person
            else:
                raise FileNotFoundError(f"gender.npy not found ({gpath}); pass gender_list=[...] explicitly")
        self.gender_list = gender_list
        device = torch.device("cuda")
        cache = {}

        def tables_for(gender):
            g = str(gender)
            if g not in cache:
                raw = smpl_tables if smpl_tables is not None else load_smpl_tables(g)
                cache[g] = raw if isinstance(raw, SMPLDeviceTables) else SMPLDeviceTables(raw, device)
            return cache[g]

        self.deformer_list = nn.ModuleList()
        self.smpl_server_list = nn.ModuleList()
        for i in range(self.num_person):
            server = SMPLServer(gender=self.gender_list[i], betas=betas2[i], smpl_tables=tables_for(self.gender_list[i]))
            self.smpl_server_list.append(server)
            self.deformer_list.append(SMPLDeformer(betas=betas2[i], gender=self.gender_list[i], server=server))

        self.sdf_bounding_sphere = 3.0
        self.threshold = 0.05
        self.shade_mode = hip.SHADE_MODE      # 'reverse' | 'forward' (csrc/mlp.hip: k_mlp_shade_rev | k_mlp_shade)
        self.density = LaplaceDensity(**opt.density)
        self.bg_density = AbsDensity()
        self.ray_sampler = ErrorBoundSampler(self.sdf_bounding_sphere, inverse_sphere_bg=True, **opt.ray_sampler)
        if opt.get("smpl_init", False):      # multiply.py:101-108
            path = os.path.abspath(opt.get("smpl_init_path", "./outputs/smpl_init_male_256.pth"))
            if not os.path.exists(path):
                raise FileNotFoundError(f"smpl_init is set but {path} (an asset of the reference, not shipped here) is missing; "
                                        f"set smpl_init: false to start from the geometric initialisation instead")
            state = torch.load(path, map_location=device)
            for net in self.foreground_implicit_network_list:
                net.load_state_dict(state["model_state_dict"], strict=False)
        self.mesh_v_cano_list = [s.verts_c for s in self.smpl_server_list]
        self.mesh_f_cano_list = [torch.tensor(s.smpl.faces.astype(np.int64), device=device) for s in self.smpl_server_list]
        self.mesh_face_vertices_list = [v[0][f] [None] for v, f in zip(self.mesh_v_cano_list, self.mesh_f_cano_list)]
        self.convergence_group = None
        self.obb_inflate = 1.2
        # cull box (multiply.py:208-214): "hull" = the minimum-volume box trimesh's bounding_box_oriented computes, by the published
        # algorithm (multiply_amd/obb.py: Qhull on the host, one extra device sync + ~3 ms per person and call; the candidate search
        # on the device, mp_obb_hull); "pca" = principal-axes box, device only (k_obb; conservative: identical eval pixels).
        # "auto" (default): the reference's box wherever the hit set changes the result -- TRAINING mode, where no outlier override
        # exists (multiply.py:142-143 is eval-only) and the rays a person is sampled on enter the loss -- and the device-only box
        # in eval mode, where both give the same pixels (test_forward_eval_box_cull_is_conservative).
        self.obb_mode = os.environ.get("MP_OBB_MODE", "auto")
        # eval-mode refinement of the box cull that provably leaves every pixel unchanged (mp_ray_cull_near); 0 = box only
        self.near_cull = os.environ.get("MP_NEAR_CULL", "1") != "0"
        # ray-sharded data-parallel training: a torch.distributed process group (or True = the default group) over which the
        # sampler's per-iteration convergence vote is all-reduced (MAX) -- see _sample_person; None: the vote is per process
        self.sampler_vote_group = None
        # arithmetic of the sampler's network queries (_sampler_sdf): 'auto' (default, round 6) = 'bf16x3' -- near-fp32: the depths
        # then agree with the fp32 reference to 1e-3 instead of 2e-2 and the grazing-ray tail of the render disappears -- for the
        # network shape csrc/tfuse.hip is specialised for (the shipped configs), 'f16x2' (split activations, any shape) otherwise;
        # 'f16': the fused half-precision kernel (the round 1-5 default), a third of the query time
        self.sampler_sdf_mode = os.environ.get("MP_SAMPLER_SDF", "auto")
        self.last_stats = {}
        self.profile = False
        self.phase_events = {}
        self.to(device)

    # ------------------------------------------------------------------ helpers
    class _Phase:
        """HIP-event bracket around a group of launches on the current stream (enabled by `model.profile = True`)."""

        def __init__(self, owner, name):
            self.o, self.name = owner, name

        def __enter__(self):
            if self.o.profile:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record()

        def __exit__(self, *a):
            if self.o.profile:
                self.e1.record()
                self.o.phase_events.setdefault(self.name, []).append((self.e0, self.e1))

    def _ph(self, name):
        return Multiply._Phase(self, name)

    def phase_times_ms(self):
        """{phase: (n_brackets, total ms)} of the events recorded since the last reset (synchronises)."""
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.phase_events.items()}

    def train(self, mode=True):
        """nn.Module.train, and every switch of mode drops the packed-weight caches (hip.invalidate_packed: an optimizer may have
        updated the parameters without bumping their version counters -- torch's fused Adam does)"""
        if bool(mode) != self.training:
            hip.invalidate_packed()
        return super().train(mode)

    def _obb_mode_now(self):
        if self.obb_mode not in ("auto", "hull", "pca"):
            raise ValueError(f"obb_mode {self.obb_mode!r}: expected 'auto', 'hull' or 'pca'")
        return ("hull" if self.training else "pca") if self.obb_mode == "auto" else self.obb_mode

    def _sampler_cfg(self):
        rs = self.ray_sampler
        return hip.MpSamplerCfg(rs.N_samples, rs.N_samples_eval, rs.N_samples_extra, rs.beta_iters, rs.max_total_iters,
                                rs.eps, rs.add_tiny, rs.near)

    def forward(self, input, id=-1, cond_zero_shit=False, canonical_pose=False):
        if self.training:
            from . import train
            return train.forward_train(self, input, id, cond_zero_shit, canonical_pose)
        with torch.no_grad():
            return self._forward_eval(input, id, canonical_pose)

    def _setup(self, input, id, canonical_pose, side_stream=False, _beta=None, host_hull=False):
        """Rays, SMPL posing, nearest-vertex structures and the box cull for every person of the call
        (multiply.py:177-266).  Ends with the one host sync of the call (hit counts size the workspaces).

        side_stream: the setup kernels read only the call's inputs, so they run on a stream of their own and the host waits
        for THAT stream only: it does not wait for the previous iteration's backward pass (or the previous frame) still
        running on the caller's stream, and keeps enqueueing.  The inputs must be resident and complete (produced by work
        the host has already waited for, e.g. a data loader's copies); everything allocated here is handed to the caller's
        stream (record_stream + an event wait).  What is NOT an input of the call stays on the caller's stream:
          * `density.beta` is a trained parameter -- the previous iteration's optimizer step may still be updating it on
            the caller's stream.  Training: the setup kernels do not read it (the near cull is eval-only) and the call's
            `beta` is computed on the caller's stream behind the join.  Eval: the near cull does read it, so the value is
            computed on the caller's stream once per parameter version and the side stream waits for that event;
          * body-model inputs that are being optimised (requires_grad / produced by BodyModelParams in this iteration on
            the caller's stream): the side stream is refused, the setup runs in order."""
        if side_stream and any(torch.is_tensor(input.get(k)) and (input[k].requires_grad or input[k].grad_fn is not None)
                               for k in ("smpl_params", "smpl_pose", "smpl_shape", "smpl_trans")):
            side_stream = False
        if side_stream:
            main = torch.cuda.current_stream()
            side = self.__dict__.get("_setup_stream")
            if side is None:
                side = self.__dict__["_setup_stream"] = torch.cuda.Stream()
            beta_in = None
            if not self.training:
                b = self.density.beta
                cache = self.__dict__.get("_eval_beta")
                # keyed on the parameter OBJECT, its storage and its version: writes through .data and a replaced Parameter with the
                # same version number must not leave a stale value behind (a load_state_dict bumps the version: copy_ in place)
                # (`id` is this method's person-id argument: the parameter is identified by its storage)
                key = (b.data_ptr(), b._version, str(b.device), hip._GENERATION[0])
                if cache is None or cache[0] != key:
                    val = (b.detach().abs() + self.density.beta_min).reshape(1).float().contiguous()   # caller's stream
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)          # once per parameter version: later side-stream work is ordered behind it
                    val.record_stream(side)
                    cache = self.__dict__["_eval_beta"] = (key, val)
                beta_in = cache[1]
            with torch.cuda.stream(side):
                cx = self._setup(input, id, canonical_pose, _beta=beta_in if beta_in is not None else "defer", host_hull=host_hull)

            def hand_over(o):
                if torch.is_tensor(o):
                    if o.is_cuda:
                        o.record_stream(main)
                elif isinstance(o, dict):
                    for v in o.values():
                        hand_over(v)
                elif isinstance(o, (list, tuple)):
                    for v in o:
                        hand_over(v)
            hand_over(cx)
            main.wait_stream(side)
            if beta_in is None:       # training: the parameter as the caller's stream sees it (behind the last optimizer step)
                cx["beta"] = (self.density.beta.detach().abs() + self.density.beta_min).reshape(1).float().contiguous()
            return cx
        L = hip.lib()
        dev = self.density.beta.device
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        st = hip.stream()
        uv = input["uv"].to(dev).float().reshape(-1, 2).contiguous()
        R = uv.shape[0]
        K = input["intrinsics"].to(dev).float().reshape(16).contiguous()
        pose = input["pose"].to(dev).float().reshape(16).contiguous()
        smpl_params = input["smpl_params"].detach().to(dev).float()
        smpl_pose = input["smpl_pose"].detach().to(dev).float()
        smpl_shape = input["smpl_shape"].detach().to(dev).float()
        smpl_trans = input["smpl_trans"].detach().to(dev).float()
        P = smpl_trans.shape[1]
        persons = list(id) if isinstance(id, (list, tuple)) else (list(range(P)) if id == -1 else [id])
        rs = self.ray_sampler
        group = int(self.convergence_group or R)
        if torch.is_tensor(_beta):
            beta = _beta
        elif _beta == "defer":     # side-stream training setup: filled in on the caller's stream; no setup kernel reads it
            assert self.training, "a deferred beta is only valid where the near cull (eval) does not run"
            beta = None
        else:
            beta = (self.density.beta.detach().abs() + self.density.beta_min).reshape(1).float().contiguous()

        # rays (rend_util.get_camera_params)
        dirs = torch.empty(R, 3, **f32)
        far = torch.empty(R, **f32)
        hip.check(L.mp_ray_setup(hip.ptr(uv), hip.ptr(K), hip.ptr(pose), R, C.c_float(self.sdf_bounding_sphere),
                                 hip.ptr(dirs), hip.ptr(far), st), "mp_ray_setup")

        # SMPL posing, nearest-vertex structures, box cull  (multiply.py:196-214, 256-266)
        per = {}
        zp = hip.ZeroPool(dev, 1 << 16)             # the setup's device counters: one fill (on the setup's stream)
        counts = zp.take(len(persons), dtype=torch.int32)
        hull_status = None
        scan_tmp = torch.empty(R + (R + 1023) // 1024 + 8, **i32)
        verts_all = torch.empty(len(persons), NUM_VERTS, 3, **f32)
        given_hits = "hit_index" in input and input["hit_index"] is not None
        device_hull = not given_hits and self._obb_mode_now() == "hull" and not host_hull
        for n, p in enumerate(persons):
            server = self.smpl_server_list[p]
            prm = torch.cat([smpl_params[0, p, 0:1], smpl_trans[0, p], smpl_pose[0, p], smpl_shape[0, p]]).contiguous()
            if canonical_pose:   # multiply.py:197-202
                prm = prm.clone()
                prm[1:4] = 0
                prm[4:76] = 0
                prm[4 + 5] = np.pi / 6
                prm[4 + 8] = -np.pi / 6
            verts = verts_all[n]
            tfs = torch.empty(NUM_JOINTS, 4, 4, **f32)
            jnts = torch.empty(NUM_JOINTS, 3, **f32)
            server.pose_into(prm, verts, tfs, jnts)
            d = self.deformer_list[p]
            vsorted = torch.empty(hip.KNN_NC * hip.KNN_CLUSTER, 4, **f32)
            cbound = torch.empty(hip.KNN_CB_ROWS, 4, **f32)
            hip.check(L.mp_knn_build(hip.ptr(verts), hip.ptr(d.knn_perm), hip.ptr(vsorted), hip.ptr(cbound), st),
                      "mp_knn_build")
            btab = torch.empty(NUM_VERTS, 12, **f32)        # per-vertex inverse blended transform of this pose
            hip.check(L.mp_blend_table(hip.ptr(server.tables.lbs_weights), hip.ptr(tfs), NUM_VERTS, hip.ptr(btab), st),
                      "mp_blend_table")
            hit_index = torch.empty(R, **i32)
            inv_index = torch.empty(R, **i32)
            cond = (smpl_pose[0, p, 3:] / np.pi).contiguous()          # multiply.py:270
            per[p] = dict(verts=verts, tfs=tfs, btab=btab, vsorted=vsorted, cbound=cbound, hit_index=hit_index, obb=None,
                          inv_index=inv_index, count=counts[n:n + 1], cond=cond, prm=prm,
                          rest_joints=server.rest_joints() if self.training else None)
        obb_all = None
        if device_hull:
            # the convex hulls (gift wrapping), the candidate searches and the boxes of ALL bodies in one batch on the device: no
            # host round trip; the status words are read with the hit counts below (a failure -- exactly coplanar vertices
            # tying into a non-manifold patch -- repeats the setup with the host-side hull)
            hull_status = zp.take(len(persons), 8, dtype=torch.int32)
            work = torch.empty(len(persons), int(L.mp_obb_hull_device_work_bytes()), dtype=torch.uint8, device=dev)
            obb_all = torch.empty(len(persons), 16, **f32)
            hip.check(L.mp_obb_hull_device(hip.ptr(verts_all), NUM_VERTS, len(persons), C.c_float(self.obb_inflate), hip.ptr(work),
                                           hip.ptr(obb_all), hip.ptr(hull_status), st), "mp_obb_hull_device")
        for n, p in enumerate(persons):
            q = per[p]
            verts, hit_index, inv_index, cbound = q["verts"], q["hit_index"], q["inv_index"], q["cbound"]
            obb = None
            if given_hits:
                hi = input["hit_index"][p].to(dev).to(torch.int32).contiguous()
                if hi.numel() == 0:      # multiply.py:262-263: no ray meets the box -> ray 0
                    hi = torch.zeros(1, dtype=torch.int32, device=dev)
                hit_index[:hi.numel()] = hi
                hip.check(L.mp_ray_hits_from_index(hip.ptr(hit_index), hi.numel(), R, hip.ptr(counts[n:n + 1]),
                                                   hip.ptr(inv_index), st), "mp_ray_hits_from_index")
            else:
                if device_hull:
                    obb = obb_all[n]
                elif self._obb_mode_now() == "hull":
                    # hull on the host (Qhull, ~3 ms; one extra device sync), search + box on the device
                    from .obb import hull_search_inputs, obb_record
                    vhost = verts.cpu().numpy()
                    buf, nh, nn_, ne = hull_search_inputs(vhost)
                    if nh > 4096:      # beyond the kernel's LDS tile (a body's hull has a few hundred vertices): the host statement
                        obb = torch.from_numpy(obb_record(vhost, self.obb_inflate)).to(dev)
                        buf = None
                    hb = torch.from_numpy(buf).to(dev) if buf is not None else None
                    if hb is not None:
                        o_n, o_e = 3 * nh, 3 * (nh + nn_)
                        work = torch.empty(2 * nn_, dtype=torch.float64, device=dev)
                        obb = torch.empty(16, **f32)
                        hip.check(L.mp_obb_hull(hip.ptr(hb), nh, hip.ptr(hb[o_n:]), nn_, hip.ptr(hb[o_e:]), hip.ptr(hb[o_e + 3 * ne:]),
                                                hip.ptr(hb[o_e + 6 * ne:]), ne, C.c_float(self.obb_inflate), hip.ptr(work),
                                                hip.ptr(obb), st), "mp_obb_hull")
                else:
                    obb = torch.empty(16, **f32)
                    hip.check(L.mp_obb(hip.ptr(verts), C.c_float(self.obb_inflate), hip.ptr(obb), st), "mp_obb")
                if self.near_cull and not self.training:
                    # eval: rays of the box that never come within the outlier radius of the body are background, bit for bit
                    # (csrc/geom.hip k_ray_near_body); they are dropped before the sampler
                    hip.check(L.mp_ray_cull_near(hip.ptr(dirs), hip.ptr(pose), hip.ptr(obb), hip.ptr(cbound), hip.ptr(far),
                                                 hip.ptr(beta), C.c_float(rs.near), R, group, hip.ptr(hit_index),
                                                 hip.ptr(counts[n:n + 1]), hip.ptr(inv_index), hip.ptr(scan_tmp), st),
                              "mp_ray_cull_near")
                else:
                    hip.check(L.mp_ray_cull(hip.ptr(dirs), hip.ptr(pose), hip.ptr(obb), R, group, hip.ptr(hit_index),
                                            hip.ptr(counts[n:n + 1]), hip.ptr(inv_index), hip.ptr(scan_tmp), st),
                              "mp_ray_cull")
            q["obb"] = obb
        if hull_status is not None:      # the one host sync of the call: sizes the per-person workspaces (+ the hulls' status)
            both = torch.cat([counts, hull_status[:, 3]]).tolist()
            n_hit = both[:len(persons)]
            if any(both[len(persons):]):
                import warnings
                # counted (advisor, round 5): a regression of the device wrap -- e.g. its grid barrier giving up under GPU sharing --
                # shows as a growing `hull_host_fallbacks` in `last_stats` / the bench line, not only as a warning
                self.hull_host_fallbacks = getattr(self, "hull_host_fallbacks", 0) + 1
                warnings.warn("device convex hull failed (degenerate vertex configuration): falling back to the host-side hull")
                return self._setup(input, id, canonical_pose, _beta=_beta, host_hull=True)
        else:
            n_hit = counts.tolist()
        return dict(dev=dev, R=R, uv=uv, K=K, pose=pose, dirs=dirs, far=far, per=per, persons=persons, n_hit=n_hit,
                    group=group, beta=beta, counts=counts, hull_status=hull_status)

    def _vote_groups_check(self, n_groups, grp):
        """Every rank of the vote's process group must contribute the same number of convergence-group flags (uneven ray shards
        with convergence_group set would mismatch the collective's sizes -- undefined behaviour on RCCL).  A FIXED-size collective,
        issued by every rank on every call in which the flag count is not 1 by construction (convergence_group set), whatever
        its own n_groups: a rank that skipped it would desynchronise the collective sequence it is meant to protect."""
        import torch.distributed as dist
        t = torch.tensor([n_groups, -n_groups], device=self.density.beta.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
        hi, lo = int(t[0]), int(-t[1])
        if hi != n_groups or lo != n_groups:
            raise RuntimeError(f"sampler vote: the ranks hold different numbers of convergence groups (this rank {n_groups}, "
                               f"range {lo}..{hi}); shard the rays at multiples of convergence_group, equally many per rank")

    # ---- ErrorBoundSampler.get_z_vals (ray_sampler.py:66-220) in four steps, so that the persons of a call can advance
    #      iteration by iteration together (one convergence-vote collective per iteration for ALL persons, _sample_persons)
    def _sampler_open(self, cx, n, p, draws=None):
        """workspaces of person p's sampler + mp_sampler_init.  draws = None: eval-mode determinism; else the training
        randomness {t_rand [R_p,NE], u_final [R_p,N], extra_idx [max_iters,N_extra] int32}."""
        L = hip.lib()
        dev = cx["dev"]
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        st = hip.stream()
        rs = self.ray_sampler
        cfg = self._sampler_cfg()
        NE, NS, NX = rs.N_samples_eval, rs.N_samples, rs.N_samples_extra
        NZ = NS + NX + 2
        ZM = NE * rs.max_total_iters
        R, group = cx["R"], cx["group"]
        n_groups = (R + group - 1) // group
        pp = cx["per"][p]
        Rp = max(int(cx["n_hit"][n]), 1)
        imp = self.foreground_implicit_network_list[p]
        pk_sdf = hip.packed(imp, "sdf", 2)
        pk_sdf.refresh(pp["cond"], force=self.training)
        zs = torch.empty(Rp, ZM, **f32); sdfs = torch.empty(Rp, ZM, **f32)
        nz = torch.empty(Rp, **i32); znew = torch.empty(Rp, NE, **f32); sdfnew = torch.empty(Rp, NE, **f32)
        betar = torch.empty(Rp, **f32); active = torch.empty(Rp, **i32)
        gflag = torch.empty((rs.max_total_iters + 1) * n_groups, **i32)
        zp = cx.get("_zp_sampler")                   # the samplers' device counters of ALL persons of the call: one fill
        if zp is None:
            zp = cx["_zp_sampler"] = hip.ZeroPool(dev, 1 << 16)
        zfinal = torch.empty(Rp, NZ, **f32); iters = zp.take(n_groups, dtype=torch.int32)
        any_active = zp.take(rs.max_total_iters + 1, dtype=torch.int32)
        state = hip.MpSamplerState(zs.data_ptr(), sdfs.data_ptr(), nz.data_ptr(), znew.data_ptr(),
                                   sdfnew.data_ptr(), betar.data_ptr(), active.data_ptr(), gflag.data_ptr(),
                                   zfinal.data_ptr(), iters.data_ptr(), any_active.data_ptr())
        train = draws is not None
        t_rand = hip.ptr(draws["t_rand"]) if train else None
        hip.check(L.mp_sampler_init(C.byref(cfg), C.byref(state), hip.ptr(cx["far"]), hip.ptr(pp["hit_index"]),
                                    hip.ptr(pp["count"]), Rp, group, R, t_rand, st), "mp_sampler_init")
        xc_new = torch.empty(Rp * NE, 3, **f32)
        work = torch.empty(Rp * NE, **i32)
        wcount = zp.take(rs.max_total_iters + 1, dtype=torch.int32)
        # training: the rays are random pixels -- the warp first groups a call's samples by their nearest vertex cluster
        bin_work = torch.empty(int(L.mp_warp_bin_work_bytes(Rp * NE)), dtype=torch.uint8, device=dev) if train else None
        return dict(cx=cx, p=p, pp=pp, Rp=Rp, NE=NE, cfg=cfg, state=state, train=train, draws=draws, pk_sdf=pk_sdf, n_groups=n_groups,
                    zs=zs, sdfs=sdfs, nz=nz, znew=znew, sdfnew=sdfnew, betar=betar, active=active, gflag=gflag, zfinal=zfinal,
                    iters=iters, any_active=any_active, xc_new=xc_new, work=work, wcount=wcount, bin_work=bin_work)

    def _sampler_query(self, s, it):
        """iteration `it`, first half: warp the new samples, query the SDF net, evaluate the error bound (sets the group flags)"""
        L, st = hip.lib(), hip.stream()
        cx, pp, Rp, NE, train = s["cx"], s["pp"], s["Rp"], s["NE"], s["train"]
        pk_sdf, any_active, wcount = s["pk_sdf"], s["any_active"], s["wcount"]
        with self._ph("sampler_warp"):
            hip.check(L.mp_warp_inverse(None, hip.ptr(cx["dirs"]), hip.ptr(cx["pose"]), hip.ptr(pp["hit_index"]),
                                        hip.ptr(pp["count"]), hip.ptr(s["znew"]), NE, NE, Rp, hip.ptr(pp["vsorted"]),
                                        hip.ptr(pp["cbound"]), hip.ptr(pp["btab"]), 0 if train else 1,
                                        hip.ptr(s["active"]), hip.ptr(any_active[it:it + 1]), hip.ptr(s["xc_new"]), None,
                                        hip.ptr(s["sdfnew"]), hip.ptr(s["work"]),
                                        hip.ptr(wcount[it:it + 1]), hip.ptr(s["bin_work"]) if train else None, st),
                      "mp_warp_inverse")
        with self._ph("sampler_mlp_sdf"):
            self._sampler_sdf(s, it)
        with self._ph("sampler_bound"):
            hip.check(L.mp_sampler_bound(C.byref(s["cfg"]), C.byref(s["state"]), hip.ptr(cx["beta"]), hip.ptr(pp["hit_index"]),
                                         hip.ptr(pp["count"]), Rp, cx["group"], cx["R"], it, st), "mp_sampler_bound")

    def resolved_sampler_sdf_mode(self, p=0):
        """`sampler_sdf_mode` with 'auto' resolved for person p's network (see __init__)"""
        mode = getattr(self, "sampler_sdf_mode", "auto")
        if mode != "auto":
            return mode
        from . import train as T
        return "bf16x3" if T.fused_sdf_supported(self.foreground_implicit_network_list[p]) else "f16x2"

    def _sampler_sdf(self, s, it):
        """the sampler's network queries of iteration `it` (`self.sampler_sdf_mode`, MP_SAMPLER_SDF):
        'bf16x3' (what 'auto' resolves to for the shipped network shape): the value sweep of the training path's layer-fused
        kernel (mp_tf_sdf_val: split-bfloat16 products, fp32 activations, ~2^-16 per product) -- near-fp32 queries, 3x the time
        of the half-precision kernel; 'f16x2': split activations on the half-precision weights (mp_mlp_sdf_x2, 2x the time, a
        quarter of the mean depth error, any network shape); 'f16': the fused half-precision kernel (csrc/mlp.hip k_mlp_sdf);
        'bf16x3-layerwise': the bf16x3 arithmetic layer by layer (the independent implementation tools/sampler_precision.py
        first measured with; reads the worklist count on the host)."""
        L, st = hip.lib(), hip.stream()
        pk_sdf, wcount = s["pk_sdf"], s["wcount"]
        mode = self.resolved_sampler_sdf_mode(s["p"])
        if mode == "bf16x3":
            # the value sweep of the training path's layer-fused kernel (csrc/tfuse.hip k_tf_sdf_val): same worklist, device-side count
            fs = s.get("fs")
            if fs is None:
                from . import train as T
                imp = self.foreground_implicit_network_list[s["p"]]
                if not T.fused_sdf_supported(imp):
                    raise NotImplementedError("sampler_sdf_mode 'bf16x3' needs the network shape csrc/tfuse.hip is specialised for")
                # once per call and person; inside a training forward (TrainGraph.run: TrainState.begin has just resolved the
                # iteration's weights) the shared layers are used, anywhere else the state resolves the weights itself
                lins = T.train_state(self).lins[id(imp)] if self.__dict__.get("_mp_in_train_graph") else None
                fs = s["fs"] = T.fused_sdf_state(imp, lins).refresh(s["pp"]["cond"])
            hip.check(L.mp_tf_sdf_val(hip.ptr(fs.wpack), hip.ptr(fs.bias_all), hip.ptr(s["xc_new"]), hip.ptr(s["work"]),
                                      hip.ptr(wcount[it:it + 1]), s["Rp"] * s["NE"], hip.ptr(s["sdfnew"]), st), "mp_tf_sdf_val")
            return
        if mode == "bf16x3-layerwise":          # the measurement path of tools/sampler_precision.py (host read per iteration)
            from . import train as T
            n = int(wcount[it])
            if n > 0:
                idx = s["work"][:n].long()
                x = s["xc_new"][idx].contiguous()
                imp = self.foreground_implicit_network_list[s["p"]]
                lins = [T.LinW(l) for l in imp.layers()]
                parts = [T.ImplicitTrain(imp, x[c0:c0 + (1 << 18)], s["pp"]["cond"], fwd=False, lins=lins).out[:, 0].clone()
                         for c0 in range(0, n, 1 << 18)]
                s["sdfnew"].view(-1)[idx] = torch.cat(parts)
            return
        if mode not in ("f16", "f16x2"):
            raise ValueError(f"sampler_sdf_mode {mode!r}: expected 'f16x2', 'f16', 'bf16x3' or 'bf16x3-layerwise'")
        # 'f16x2': split activations on the same packed half-precision weights (csrc/mlp.hip k_mlp_sdf_x2)
        fn = L.mp_mlp_sdf_x2 if mode == "f16x2" else L.mp_mlp_sdf
        hip.check(fn(C.byref(pk_sdf.net), hip.ptr(pk_sdf.wpack), hip.ptr(pk_sdf.bias),
                     hip.ptr(s["xc_new"]), hip.ptr(s["work"]), hip.ptr(wcount[it:it + 1]), s["Rp"] * s["NE"],
                     hip.ptr(s["sdfnew"]), st), "mp_mlp_sdf_x2" if mode == "f16x2" else "mp_mlp_sdf")

    def _sampler_resample(self, s, it):
        """iteration `it`, second half: new samples where the bound is not met (or, converged, the final inverse-CDF draw)"""
        L, st = hip.lib(), hip.stream()
        cx, pp, draws = s["cx"], s["pp"], s["draws"]
        u_final = hip.ptr(draws["u_final"]) if s["train"] else None
        extra_idx = hip.ptr(draws["extra_idx"]) if s["train"] else None
        with self._ph("sampler_resample"):
            hip.check(L.mp_sampler_resample(C.byref(s["cfg"]), C.byref(s["state"]), hip.ptr(cx["beta"]), hip.ptr(cx["far"]),
                                            hip.ptr(pp["hit_index"]), hip.ptr(pp["count"]), s["Rp"], cx["group"], cx["R"], it,
                                            u_final, extra_idx, st), "mp_sampler_resample")

    def _sampler_close(self, s):
        s["pp"]["_sampler_keep"] = tuple(s[k] for k in ("zs", "sdfs", "nz", "znew", "sdfnew", "betar", "active", "gflag",
                                                       "any_active", "xc_new", "work", "draws"))
        return s["zfinal"], s["iters"], s["wcount"]

    def _sample_persons(self, cx, draws_by_person=None, persons=None):
        """The sampler of EVERY person of the call, advancing iteration by iteration together -> {p: (zfinal, iters, wcount)}.
        The persons' samplers are independent (same launches as one after the other, other order); what the interleaving buys
        is the data-parallel convergence vote: the reference's `not_converge = beta.max() > beta0` (ray_sampler.py:137) spans
        ALL rays of the call -- here the rays of every rank -- and with `sampler_vote_group` set ONE MAX all-reduce per sampler
        iteration carries the flags of all persons (P x n_groups ints; P x max_total_iters collectives before), between the
        bound and the resampling kernels: the N-rank step samples exactly like the single-process step."""
        persons = list(cx["persons"]) if persons is None else list(persons)
        order = {p: n for n, p in enumerate(cx["persons"])}
        states = [self._sampler_open(cx, order[p], p, None if draws_by_person is None else draws_by_person[p]) for p in persons]
        vote = self.sampler_vote_group is not None
        if vote and states:
            import torch.distributed as dist
            grp = None if self.sampler_vote_group is True else self.sampler_vote_group
            ng = states[0]["n_groups"]
            if self.convergence_group is not None:      # (None: one flag per person and call on every rank, by construction)
                self._vote_groups_check(ng, grp)
        for it in range(self.ray_sampler.max_total_iters):
            for s in states:
                self._sampler_query(s, it)
            if vote and states:
                flags = [s["gflag"][it * ng:(it + 1) * ng] for s in states]
                if len(flags) == 1:
                    dist.all_reduce(flags[0], op=dist.ReduceOp.MAX, group=grp)
                else:
                    packed = torch.cat(flags)
                    dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=grp)
                    torch._foreach_copy_(flags, list(packed.split(ng)))
                self.vote_collectives = getattr(self, "vote_collectives", 0) + 1
            for s in states:
                self._sampler_resample(s, it)
        return {s["p"]: self._sampler_close(s) for s in states}

    def _sample_person(self, cx, n, p, draws=None):
        """ErrorBoundSampler.get_z_vals for person p's rays (ray_sampler.py:66-220): returns zfinal [R_p][N+N_extra+2],
        the iteration counters and the per-iteration SDF worklist counts."""
        return self._sample_persons(cx, None if draws is None else {p: draws}, persons=[p])[p]

    def sample_rays(self, ray_dirs, cam_loc, cond, smpl_tfs, smpl_verts, person_id, draws=None):
        """The sampler on explicit rays, outside forward(): what ErrorBoundSampler.get_z_vals(ray_dirs, cam_loc, model, cond,
        smpl_tfs, eval_mode, smpl_verts, person_id) does in the reference (ray_sampler.py:66-220) for ONE person (draws = None: eval mode):
        every ray is sampled (no box cull), the convergence vote spans the call.  ray_dirs (R,3) unit vectors, cam_loc (3,)
        or (R,3) with equal rows, cond the pose conditioning (69,) / {'smpl': (1,69)}, smpl_tfs (1,24,4,4), smpl_verts
        (1,6890,3) posed vertices.  -> z_vals (R, N_samples + N_samples_extra + 2) sorted depths."""
        L = hip.lib()
        st = hip.stream()
        dev = self.density.beta.device
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        dirs = ray_dirs.detach().to(dev).float().reshape(-1, 3).contiguous()
        R = dirs.shape[0]
        cam = cam_loc.detach().to(dev).float().reshape(-1, 3)[0].contiguous()
        pose = torch.eye(4, **f32)
        pose[:3, 3] = cam
        # far end: the ray / bounding-sphere intersection (rend_util.get_sphere_intersections, rend_util.py:131-147)
        od = (dirs * cam).sum(-1)
        far = (-od + torch.sqrt((od * od - (cam @ cam - self.sdf_bounding_sphere ** 2)).clamp_min(0.0))).contiguous()
        verts = smpl_verts.detach().to(dev).float().reshape(-1, 3).contiguous()
        tfs = smpl_tfs.detach().to(dev).float().reshape(24, 16).contiguous()
        d = self.deformer_list[person_id]
        vsorted = torch.empty(hip.KNN_NC * hip.KNN_CLUSTER, 4, **f32)
        cbound = torch.empty(hip.KNN_CB_ROWS, 4, **f32)
        hip.check(L.mp_knn_build(hip.ptr(verts), hip.ptr(d.knn_perm), hip.ptr(vsorted), hip.ptr(cbound), st), "mp_knn_build")
        btab = torch.empty(verts.shape[0], 12, **f32)
        hip.check(L.mp_blend_table(hip.ptr(self.smpl_server_list[person_id].tables.lbs_weights), hip.ptr(tfs), verts.shape[0],
                                   hip.ptr(btab), st), "mp_blend_table")
        if isinstance(cond, dict):
            cond = cond["smpl"]
        cvec = cond.detach().to(dev).float().reshape(-1).contiguous()
        beta = (self.density.beta.detach().abs() + self.density.beta_min).reshape(1).float().contiguous()
        per = {person_id: dict(verts=verts, tfs=tfs, btab=btab, vsorted=vsorted, cbound=cbound,
                               hit_index=torch.arange(R, **i32), count=torch.full((1,), R, **i32), cond=cvec)}
        cx = dict(dev=dev, R=R, pose=pose.reshape(16).contiguous(), dirs=dirs, far=far, per=per, persons=[person_id], n_hit=[R],
                  group=R, beta=beta)
        with torch.no_grad():      # draws (training mode): {t_rand [R,NE], u_final [R,N], extra_idx [max_iters,N_extra] int32}
            zfinal, _, _ = self._sample_person(cx, 0, person_id, draws)
        return zfinal

    def _forward_eval(self, input, id, canonical_pose, composite=True):
        """composite=False: stop after the per-person sampling + shading and return the per-person sample arrays (in hit
        order) -- the person-sharded multi-GPU mode composites them elsewhere (parallel.render_person_sharded)."""
        L = hip.lib()
        # async_setup (inputs resident and complete, e.g. a render loop over preloaded frames): the setup's host sync waits
        # for the setup kernels only, not for the previous frame still in flight on this stream
        cx = self._setup(input, id, canonical_pose, side_stream=bool(getattr(self, "async_setup", False)))
        dev, R, dirs, far, pose, beta = cx["dev"], cx["R"], cx["dirs"], cx["far"], cx["pose"], cx["beta"]
        per, persons, n_hit = cx["per"], cx["persons"], cx["n_hit"]
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        st = hip.stream()
        rs = self.ray_sampler
        NZ = rs.N_samples + rs.N_samples_extra + 2
        S = NZ - 1
        stats = {"n_hit": n_hit, "iters": [], "n_sdf_evals": [], "n_shaded": [], "hull_host_fallbacks": getattr(self, "hull_host_fallbacks", 0)}

        for n, p in enumerate(persons):
            pp = per[p]
            Rp = max(int(n_hit[n]), 1)
            imp, ren, dfm = self.foreground_implicit_network_list[p], self.foreground_rendering_network_list[p], \
                self.deformer_list[p]
            skin_w = self.smpl_server_list[p].tables.lbs_weights
            zfinal, iters, wcount = self._sample_person(cx, n, p)
            # ---- shading of the final samples (multiply.py:294-308, 403-405)
            npts = Rp * S
            xc = torch.empty(npts, 3, **f32)
            sdf = torch.empty(npts, **f32)
            nrm = torch.zeros(npts, 3, **f32)
            rgb = torch.zeros(npts, 3, **f32)
            work2 = torch.empty(npts, **i32)
            need = torch.empty(npts, dtype=torch.uint8, device=dev)
            nn_posed = torch.empty(npts, **i32)
            wc2 = wcount[rs.max_total_iters:]
            ph = self._ph("shade_warp"); ph.__enter__()
            hip.check(L.mp_warp_inverse_shade(hip.ptr(dirs), hip.ptr(pose), hip.ptr(pp["hit_index"]), hip.ptr(pp["count"]),
                                              hip.ptr(zfinal), NZ, S, Rp, hip.ptr(pp["vsorted"]), hip.ptr(pp["cbound"]),
                                              hip.ptr(pp["btab"]), 1, hip.ptr(beta), hip.ptr(xc), None,
                                              hip.ptr(need), hip.ptr(sdf), hip.ptr(work2), hip.ptr(wc2), hip.ptr(nn_posed), None, st),
                      "mp_warp_inverse_shade")
            ph.__exit__()
            jinv = torch.empty(npts, 9, **f32)
            ph = self._ph("shade_jacobian"); ph.__enter__()
            hip.check(L.mp_warp_jacobian(hip.ptr(xc), hip.ptr(need), hip.ptr(pp["count"]), Rp, S, 0,
                                         hip.ptr(dfm.vsorted_c), hip.ptr(dfm.cbound_c), hip.ptr(pp["btab"]),
                                         hip.ptr(jinv), None, hip.ptr(nn_posed), hip.ptr(dfm.verts_c_flat), st), "mp_warp_jacobian")
            ph.__exit__()
            pk_full = hip.packed(imp, "full", 2)
            pk_full.refresh(pp["cond"])
            pk_col = hip.packed(ren, "color", 2)
            pe = ren.__dict__.get("_mp_pose_embed") or hip.PoseEmbed(ren)
            ren.__dict__["_mp_pose_embed"] = pe
            pk_col.refresh(pe(pp["cond"]))
            feat = torch.empty(((npts + 255) // 256) * 4 * 8 * 4 * 1024, dtype=torch.uint8, device=dev)
            if self.shade_mode == "reverse":     # value sweep (parks the sigmoids) + reverse sweep for the normals
                with self._ph("mlp_shade"):
                    hip.shade_rev_launch(pk_full, hip.grad_net(imp), xc, jinv, work2, wc2, npts, sdf, nrm, feat)
            else:
                with self._ph("mlp_shade"):
                    hip.check(L.mp_mlp_shade(C.byref(pk_full.net), hip.ptr(pk_full.wpack), hip.ptr(pk_full.bias), hip.ptr(xc),
                                             hip.ptr(jinv), hip.ptr(work2), hip.ptr(wc2), npts, hip.ptr(sdf), hip.ptr(nrm),
                                             hip.ptr(feat), st), "mp_mlp_shade")
            with self._ph("mlp_color"):
                hip.check(L.mp_mlp_color(C.byref(pk_col.net), hip.ptr(pk_col.wpack), hip.ptr(pk_col.bias), hip.ptr(xc),
                                         hip.ptr(nrm), hip.ptr(feat), hip.ptr(work2), hip.ptr(wc2), npts, hip.ptr(rgb),
                                         st), "mp_mlp_color")
            stats["iters"].append(iters); stats["n_sdf_evals"].append(wcount)
            per[p].update(zfinal=zfinal, sdf=sdf, rgb=rgb, nrm=nrm, xc=xc, work2=work2, wc2=wc2)

        if not composite:
            stats["n_shaded"] = [per[p]["wc2"] for p in persons]
            self.last_stats = stats
            self._last = dict(per=per, dirs=dirs, far=far, persons=persons, cx=cx)
            return {p: dict(z=per[p]["zfinal"], sdf=per[p]["sdf"], rgb=per[p]["rgb"], nrm=per[p]["nrm"],
                            hit_index=per[p]["hit_index"][:max(int(n_hit[n]), 1)], n_hit=int(n_hit[n]))
                    for n, p in enumerate(persons)}

        stats["n_shaded"] = [per[p]["wc2"] for p in persons]
        self.last_stats = stats
        bg_rgb = self._background(input, cx)
        out, bg_T, keep = self._composite(cx, persons, bg_rgb)
        self._last = dict(per=per, dirs=dirs, far=far, bg_T=bg_T, bg_rgb=bg_rgb, persons=persons, keep=keep)
        return out

    def _background(self, input, cx):
        """NeRF++ background colour of every ray of the call, or None without a frame index (multiply.py:482-484,
        514-539)."""
        if input.get("idx", None) is None:
            return None
        rs = self.ray_sampler
        dev = cx["dev"]
        key = "image_id" if "image_id" in input else "idx"      # multiply.py:407-410
        w_lat = self.frame_latent_encoder.weight.detach()        # (row looked up on the device: no device -> host wait)
        code = w_lat.index_select(0, torch.as_tensor(input[key]).reshape(-1)[:1].to(w_lat.device, torch.long))[0]
        t = torch.linspace(0.0, 1.0, rs.N_samples_inverse_sphere, device=dev)
        z_bg = torch.flip(t * (1.0 / rs.scene_bounding_sphere), dims=[0]).contiguous()
        with self._ph("background"):
            return hip.background(self.bg_implicit_network, self.bg_rendering_network, cx["dirs"],
                                  cx["pose"].reshape(4, 4)[:3, 3].contiguous(), z_bg, code,
                                  radius=self.sdf_bounding_sphere)

    def _composite(self, cx, persons, bg_rgb):
        """Packed multi-person compositing of the per-person sample arrays in cx['per'] for the subset `persons`
        (multiply.py:425-480, 544-545) -> (output dict, background transmittance, tensors to keep alive)."""
        L = hip.lib()
        dev, R, per = cx["dev"], cx["R"], cx["per"]
        f32 = dict(dtype=torch.float32, device=dev)
        NZ = self.ray_sampler.N_samples + self.ray_sampler.N_samples_extra + 2

        def table(key):
            return hip.device_ints([per[p][key].data_ptr() for p in persons], dev)
        t_inv, t_z, t_sdf, t_rgb, t_nrm = table("inv_index"), table("zfinal"), table("sdf"), table("rgb"), table("nrm")
        rgb_values = torch.empty(R, 3, **f32); fg_rgb_values = torch.empty(R, 3, **f32)
        normal_values = torch.empty(R, 3, **f32); acc_map = torch.empty(R, **f32)
        acc_person = torch.empty(R, len(persons), **f32); bg_T = torch.empty(R, **f32)
        with self._ph("composite"):
            hip.check(L.mp_composite(R, len(persons), NZ, hip.ptr(t_inv), hip.ptr(t_z), hip.ptr(t_sdf), hip.ptr(t_rgb),
                                     hip.ptr(t_nrm), hip.ptr(cx["beta"]), hip.ptr(bg_rgb) if bg_rgb is not None else None,
                                     hip.ptr(rgb_values), hip.ptr(fg_rgb_values), hip.ptr(normal_values), hip.ptr(acc_map),
                                     hip.ptr(acc_person), hip.ptr(bg_T), hip.stream()), "mp_composite")
        out = {"acc_map": acc_map, "acc_person_list": acc_person, "rgb_values": rgb_values,
               "fg_rgb_values": fg_rgb_values, "normal_values": normal_values}
        return out, bg_T, (t_inv, t_z, t_sdf, t_rgb, t_nrm)

    def render_views(self, input, ids=None, canonical_pose=False):
        """Every view the reference's caller renders of one frame -- all persons (id -1) and each person alone
        (multiply_model.py:982-989, 1183-1190: P + 1 full chunk loops of forward(s, id)) -- from ONE sampling + shading
        pass: a person's samples do not depend on who else is rendered (multiply.py:254-423 loops persons independently),
        so view `id` is a compositing pass over that person's arrays plus the shared background.
        Returns {id: the eval output dict of forward(input, id)}; bit-identical to the separate calls."""
        assert not self.training, "render_views is an eval-mode entry point"
        with torch.no_grad():
            samples = self._forward_eval(input, -1, canonical_pose, composite=False)
            cx = self._last["cx"]
            persons = cx["persons"]
            if ids is None:
                ids = [-1] + (persons if len(persons) > 1 else [])
            bg_rgb = self._background(input, cx)
            views, keep = {}, []
            for i in ids:
                out, bg_T, k = self._composite(cx, persons if i == -1 else [int(i)], bg_rgb)
                views[i] = out
                keep.append((bg_T, k))
            self._last.update(bg_rgb=bg_rgb, keep=keep, samples=samples)
            return views
