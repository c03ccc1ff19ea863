"""SMPLDeformer with the reference's attribute surface (code/lib/model/deformer.py:6-50).

The canonical<->deformed warp of the hot path (nearest posed vertex -> its skinning weights -> inverse of the blended
bone transform) runs in csrc/geom.hip (mp_warp_inverse / mp_warp_jacobian).  This module owns the static data those
kernels need: the canonical vertices of this person's shape, the skinning weights and the nearest-neighbour cluster
order (computed once from the canonical vertices; clusters stay compact under posing because nearby vertices move
almost rigidly together)."""
import torch
import torch.nn as nn

from . import hip
from .smpl import knn_cluster_perm


class SMPLDeformer(nn.Module):
    def __init__(self, max_dist=0.05, K=1, gender="male", betas=None, server=None):
        super().__init__()
        if server is None:
            from .smpl import SMPLServer
            server = SMPLServer(gender=gender, betas=betas)
        self.max_dist = max_dist
        self.K = K
        # deformer.py:11: the deformer owns ITS OWN SMPLServer(gender) (no betas) -- a second set of SMPL parameters in the
        # checkpoint (deformer_list.N.smpl.smpl.*); the device tables are shared with the scene's server
        from .smpl import SMPLServer as _Server
        self.smpl = _Server(gender=gender, smpl_tables=server.tables)
        # canonical ("A-pose") vertices of this shape = SMPLServer(betas).verts_c (deformer.py:12-18)
        self.smpl_verts = server.verts_c
        self.smpl_weights = server.tables.lbs_weights[None]
        self.verts_c_flat = self.smpl_verts[0].detach().float().contiguous()      # (V,3), original vertex order
        dev = self.smpl_verts.device
        self.knn_perm = torch.from_numpy(knn_cluster_perm(self.smpl_verts[0].cpu().numpy())).to(dev)
        self.vsorted_c = torch.empty(hip.KNN_NC * hip.KNN_CLUSTER, 4, dtype=torch.float32, device=dev)
        self.cbound_c = torch.empty(hip.KNN_CB_ROWS, 4, dtype=torch.float32, device=dev)
        hip.check(hip.lib().mp_knn_build(hip.ptr(self.smpl_verts[0].contiguous()), hip.ptr(self.knn_perm),
                                         hip.ptr(self.vsorted_c), hip.ptr(self.cbound_c), hip.stream()), "mp_knn_build")

    def forward(self, x, smpl_tfs, return_weights=True, inverse=False, smpl_verts=None):
        """deformer.py:19-30.  The hot-path call pattern (return_weights=False, inverse=True, K=1) takes the fused warp
        kernel; every other combination (weights only, forward skinning, K up to 8) goes through mp_query_weights /
        mp_skinning.  Returns weights (1,N,24), or (x_transformed (N,3), outlier_mask (N,))."""
        if x.shape[0] == 0:
            return x
        L = hip.lib()
        dev = x.device
        x = x.detach().float().contiguous()
        verts = (self.smpl_verts if smpl_verts is None else smpl_verts)[0].detach().float().contiguous()
        n = x.shape[0]
        if return_weights or not inverse or self.K != 1:
            w, outl = self._query(x, verts)
            if return_weights:
                return w[None]
            tfs = smpl_tfs.detach().float().reshape(24, 16).contiguous()
            out = torch.empty(n, 3, dtype=torch.float32, device=dev)
            hip.check(L.mp_skinning(hip.ptr(x), hip.ptr(w), n, hip.ptr(tfs), int(bool(inverse)), hip.ptr(out), hip.stream()),
                      "mp_skinning")
            return out, outl.bool()
        vs = torch.empty(hip.KNN_NC * hip.KNN_CLUSTER, 4, dtype=torch.float32, device=dev)
        cb = torch.empty(hip.KNN_CB_ROWS, 4, dtype=torch.float32, device=dev)
        hip.check(L.mp_knn_build(hip.ptr(verts), hip.ptr(self.knn_perm), hip.ptr(vs), hip.ptr(cb), hip.stream()),
                  "mp_knn_build")
        xc = torch.empty(n, 3, dtype=torch.float32, device=dev)
        outl = torch.empty(n, dtype=torch.uint8, device=dev)
        hip.check(L.mp_warp_inverse(hip.ptr(x), None, None, None, None, None, 0, 1, n, hip.ptr(vs), hip.ptr(cb),
                                    hip.ptr(self._blend_table(smpl_tfs)), 0, None, None, hip.ptr(xc),
                                    hip.ptr(outl), None, None, None, None, hip.stream()), "mp_warp_inverse")
        return xc, outl.bool()

    def _blend_table(self, smpl_tfs):
        """per-vertex inverse blended transforms of one pose (mp_blend_table), (V,12)"""
        tfs = smpl_tfs.detach().float().reshape(24, 16).contiguous()
        w = self.smpl_weights[0].contiguous()
        tab = torch.empty(w.shape[0], 12, dtype=torch.float32, device=w.device)
        hip.check(hip.lib().mp_blend_table(hip.ptr(w), hip.ptr(tfs), w.shape[0], hip.ptr(tab), hip.stream()), "mp_blend_table")
        return tab

    def _query(self, x, verts):
        n = x.shape[0]
        w = torch.empty(n, 24, dtype=torch.float32, device=x.device)
        outl = torch.empty(n, dtype=torch.uint8, device=x.device)
        hip.check(hip.lib().mp_query_weights(hip.ptr(x), n, hip.ptr(verts), verts.shape[0],
                                             hip.ptr(self.smpl_weights[0].contiguous()), int(self.K), hip.ptr(w),
                                             hip.ptr(outl), hip.stream()), "mp_query_weights")
        return w, outl

    def query_skinning_weights_smpl_multi(self, pts, smpl_verts, smpl_weights=None):
        """deformer.py:37-50: pts (1,N,3), smpl_verts (V,3) -> weights (1,N,24) (detached), outlier_mask (N,)"""
        w, outl = self._query(pts[0].detach().float().contiguous(), smpl_verts.detach().float().contiguous())
        return w[None], outl.bool()

    def query_weights(self, xc):
        """deformer.py:52-54: skinning weights of canonical points (K nearest canonical vertices)"""
        return self.forward(xc, None, return_weights=True, inverse=False)

    def forward_skinning(self, xc, cond, smpl_tfs):
        """deformer.py:31-35: canonical -> deformed, xc (1,N,3) -> (1,N,3)"""
        x = xc[0].detach().float().contiguous()
        w, _ = self._query(x, self.smpl_verts[0].contiguous())
        tfs = smpl_tfs.detach().float().reshape(24, 16).contiguous()
        out = torch.empty_like(x)
        hip.check(hip.lib().mp_skinning(hip.ptr(x), hip.ptr(w), x.shape[0], hip.ptr(tfs), 0, hip.ptr(out), hip.stream()),
                  "mp_skinning")
        return out[None]

    def forward_skinning_jacobian_inverse(self, xc, smpl_tfs):
        """inverse of d(forward_skinning)/d x_c at canonical points (deformer.py:31-35 + multiply.py:625-641)."""
        xc = xc.detach().float().contiguous()
        n = xc.shape[0]
        jinv = torch.empty(n, 9, dtype=torch.float32, device=xc.device)
        hip.check(hip.lib().mp_warp_jacobian(hip.ptr(xc), None, None, 0, 0, n, hip.ptr(self.vsorted_c),
                                             hip.ptr(self.cbound_c), hip.ptr(self._blend_table(smpl_tfs)),
                                             hip.ptr(jinv), None, None, None, hip.stream()), "mp_warp_jacobian")
        return jinv.reshape(n, 3, 3)
