"""Minimum-volume oriented bounding box of a point cloud -- what the reference's cull asks trimesh for
(`smpl_mesh.bounding_box_oriented`, code/lib/model/multiply.py:208-214; trimesh is a third-party dependency that is absent here,
its version unpinned by the reference: requirement.txt:10).

Published algorithm (trimesh.bounds.oriented_bounds; O'Rourke's theorem restricted to face-flush boxes): the box has one face
flush with a facet of the convex hull; for every hull-facet normal n the points are projected onto the plane perpendicular to n
and the minimum-AREA rectangle of the projection is found -- it has a side collinear with an edge of the 2-D hull (rotating
calipers); volume = area x extent along n; the smallest wins.

Host side (numpy + scipy's Qhull wrapper for the 3-D hull), O(facets x silhouette edges x hull vertices).  The 2-D hulls are
never built: an edge of the projection's hull is the projection of a SILHOUETTE edge of the 3-D hull (an edge whose two facets
face opposite ways with respect to n), so those edges are the candidate directions -- a superset of the 2-D hull's edges cannot
beat the optimum, which the theorem places ON a hull edge.

`Multiply.obb_mode = "hull"` (the default in training mode, where the hit set decides which samples exist): the HULL is built
on the host (Qhull, ~3 ms for a posed body; needs the posed vertices there: one device sync per person and call, like the
reference's own trimesh call), the candidate search and the box run on the device in fp64 (csrc/geom.hip mp_obb_hull; the
numpy statement below, `min_volume_obb`, takes 50 ms and stays as the host-side cross-check of that kernel).  `"pca"`
(csrc/geom.hip k_obb) never leaves the device; it is proven conservative for eval renders (identical pixels)."""
import numpy as np


def _hull(points):
    from scipy.spatial import ConvexHull
    return ConvexHull(points)


def _hull_parts(points):
    """hull vertices (H, 3), distinct facet normals in order of first appearance (N, 3; one of +n / -n), and per hull edge its
    vector and the normals of its two facets (E, 3 each) -- float64"""
    p = np.asarray(points, dtype=np.float64)
    hull = _hull(p)
    hv = p[hull.vertices]
    # unique facet normals (Qhull triangulates coplanar facets: merge them; keep one of +n / -n)
    n_all = hull.equations[:, :3]
    flip = (n_all[:, 0] < 0) | ((n_all[:, 0] == 0) & (n_all[:, 1] < 0)) | ((n_all[:, 0] == 0) & (n_all[:, 1] == 0) & (n_all[:, 2] < 0))
    n_c = np.where(flip[:, None], -n_all, n_all)
    _, first = np.unique(np.round(n_c, 9), axis=0, return_index=True)
    normals = n_c[np.sort(first)]
    # hull edges with their two facets
    simp = hull.simplices
    e = np.sort(np.concatenate([simp[:, [0, 1]], simp[:, [1, 2]], simp[:, [2, 0]]]), axis=1)
    fid = np.tile(np.arange(simp.shape[0]), 3)
    order = np.lexsort((e[:, 1], e[:, 0]))
    e, fid = e[order], fid[order]
    assert e.shape[0] % 2 == 0 and np.array_equal(e[0::2], e[1::2]), "hull is not a closed 2-manifold"
    edges, fa, fb = e[0::2], fid[0::2], fid[1::2]
    evec = p[edges[:, 1]] - p[edges[:, 0]]                           # (E, 3)
    fn = hull.equations[:, :3]
    return hv, normals, evec, fn[fa], fn[fb]


def hull_search_inputs(points):
    """the five fp64 arrays mp_obb_hull reads (include/multiply_hip.h), as ONE contiguous float64 buffer + their row counts:
    [hull_verts | normals | edge_vec | edge_na | edge_nb] -- one host-to-device copy per person"""
    hv, normals, evec, ena, enb = _hull_parts(points)
    buf = np.ascontiguousarray(np.concatenate([hv, normals, evec, ena, enb]).reshape(-1), dtype=np.float64)
    return buf, hv.shape[0], normals.shape[0], evec.shape[0]


def min_volume_obb(points):
    """points (V, 3) -> (centre (3,), axes (3, 3) rows = unit box axes, half_extents (3,)), float64.
    The first axis is the winning hull-facet normal, the other two span its plane (the rectangle's sides)."""
    hv, normals, evec, ena, enb = _hull_parts(points)
    best = (np.inf, None)
    for n in normals:
        sa, sb = ena @ n, enb @ n
        sil = (sa * sb <= 1e-12)                                     # facets facing opposite ways (or edge-on): silhouette
        d = evec[sil]
        d = d - np.outer(d @ n, n)                                   # candidate side directions in the plane
        ln = np.linalg.norm(d, axis=1)
        d = d[ln > 1e-12] / ln[ln > 1e-12, None]
        if d.shape[0] == 0:
            continue
        w = np.cross(n, d)                                           # the other side direction, (S, 3)
        pu, pw = hv @ d.T, hv @ w.T                                  # (H, S)
        area = (pu.max(0) - pu.min(0)) * (pw.max(0) - pw.min(0))
        k = int(np.argmin(area))
        h = hv @ n
        vol = area[k] * (h.max() - h.min())
        if vol < best[0]:
            best = (vol, (n, d[k], w[k]))
    n, u, w = best[1]
    axes = np.stack([n, u, w])
    proj = hv @ axes.T                                               # (H, 3)
    lo, hi = proj.min(0), proj.max(0)
    centre = ((lo + hi) * 0.5) @ axes
    return centre, axes, (hi - lo) * 0.5


def obb_record(points, inflate):
    """the 15 floats mp_ray_cull reads (include/multiply_hip.h mp_obb): centre, axes (3 rows), half extents x inflate
    (multiply.py:212: Box(extents * 1.2, transform))"""
    c, a, h = min_volume_obb(points)
    return np.concatenate([c, a.reshape(-1), h * float(inflate), [0.0]]).astype(np.float32)
