"""CPU restatement of the reference's MISE octree extractor and mesh-extraction driver -- TEST INFRASTRUCTURE ONLY
(tests/; the product path is multiply_amd/mesh.py + csrc/mise.hip).

Follows code/lib/libmise/mise.pyx (:33-368: __cinit__, update, query, to_dense, subdivide_voxels, subdivide_voxel,
get_voxel_idx) and the driver generate_mesh (code/lib/utils/mesh.py:78-131).  Pinned by tests/golden/mise_golden.npz,
produced by the reference's OWN extractor built from its source with Cython (oracle/Makefile ->
oracle/_ref/mise*.so, tests/golden/make_mise_golden.py).  skimage's marching cubes and trimesh's component split, which
generate_mesh calls afterwards, are third party and absent here: the surface extraction is checked geometrically
(tests/test_mesh_*), parity with skimage's vertex / face order is UNPINNED.
"""
import numpy as np


class MISE:
    """Multiresolution IsoSurface Extraction state: resolution_0^3 voxels of edge 2^depth lattice units, refined where the
    known lattice values straddle `threshold`."""

    def __init__(self, resolution_0, depth, threshold):
        self.resolution_0, self.depth, self.threshold = resolution_0, depth, threshold
        self.voxel_size_0 = 1 << depth
        self.resolution = resolution_0 * self.voxel_size_0
        # voxels: list of [x, y, z, level, is_leaf, children(dict)] ; the first resolution_0^3 are the level-0 grid (x-major)
        self.voxels = []
        for i in range(resolution_0):
            for j in range(resolution_0):
                for k in range(resolution_0):
                    self.voxels.append([i * self.voxel_size_0, j * self.voxel_size_0, k * self.voxel_size_0, 0, True, None])
        self.points = []            # [x, y, z] in insertion order
        self.values = []
        self.known = []
        self.index = {}             # (x, y, z) -> position in self.points
        for i in range(resolution_0 + 1):
            for j in range(resolution_0 + 1):
                for k in range(resolution_0 + 1):
                    self._add_point((i * self.voxel_size_0, j * self.voxel_size_0, k * self.voxel_size_0))

    def _add_point(self, loc):
        self.index[loc] = len(self.points)
        self.points.append(loc)
        self.values.append(0.0)
        self.known.append(False)

    def query(self):
        """lattice points whose value is still unknown, in insertion order (mise.pyx:99-120)"""
        pts = [p for p, k in zip(self.points, self.known) if not k]
        return np.asarray(pts, dtype=np.int64).reshape(-1, 3)

    def update(self, points, values):
        """mise.pyx:80-97"""
        for p, v in zip(np.asarray(points), np.asarray(values, dtype=np.float64)):
            i = self.index[(int(p[0]), int(p[1]), int(p[2]))]
            self.values[i] = float(v)
            self.known[i] = True
        self._subdivide_voxels()

    def _voxel_idx(self, x, y, z):
        """index of the LEAF voxel containing fine voxel (x, y, z), -1 outside (mise.pyx:275-322)"""
        R = self.resolution
        if not (0 <= x < R and 0 <= y < R and 0 <= z < R):
            return -1
        r0 = self.resolution_0
        idx = r0 * r0 * (x >> self.depth) + r0 * (y >> self.depth) + (z >> self.depth)
        level = 0
        while not self.voxels[idx][4]:
            level += 1
            shift = self.depth - level
            v = self.voxels[idx]
            cx, cy, cz = ((x - v[0]) >> shift) & 1, ((y - v[1]) >> shift) & 1, ((z - v[2]) >> shift) & 1
            idx = v[5][(cx, cy, cz)]
        return idx

    def _subdivide_voxels(self):
        """mise.pyx:172-222: a leaf is active when the KNOWN lattice points touching it include a value >= threshold and
        a value <= threshold; all active leaves above the finest level are split"""
        n = len(self.voxels)
        pos, neg = [False] * n, [False] * n
        for (x, y, z), v, k in zip(self.points, self.values, self.known):
            if not k:
                continue
            for i in (-1, 0):
                for j in (-1, 0):
                    for l in (-1, 0):
                        idx = self._voxel_idx(x + i, y + j, z + l)
                        if idx == -1:
                            continue
                        if v >= self.threshold:
                            pos[idx] = True
                        if v <= self.threshold:
                            neg[idx] = True
        for idx in range(n):
            v = self.voxels[idx]
            if v[4] and v[3] < self.depth and pos[idx] and neg[idx]:
                self._subdivide_voxel(idx)

    def _subdivide_voxel(self, idx):
        """mise.pyx:224-273"""
        v = self.voxels[idx]
        new_level = v[3] + 1
        size = 1 << (self.depth - new_level)
        v[4] = False
        v[5] = {}
        for i in range(2):
            for j in range(2):
                for k in range(2):
                    v[5][(i, j, k)] = len(self.voxels)
                    self.voxels.append([v[0] + i * size, v[1] + j * size, v[2] + k * size, new_level, True, None])
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    loc = (v[0] + i * size, v[1] + j * size, v[2] + k * size)
                    if loc not in self.index:
                        self._add_point(loc)

    def to_dense(self):
        """(resolution+1)^3 float64: known values, holes filled from the previous index along x, then y, then z
        (mise.pyx:122-154)"""
        n = self.resolution + 1
        out = np.full((n, n, n), np.nan)
        for (x, y, z), v in zip(self.points, self.values):
            out[x, y, z] = v
        for i in range(1, n):
            hole = np.isnan(out[i])
            out[i][hole] = out[i - 1][hole]
        for j in range(1, n):
            hole = np.isnan(out[:, j])
            out[:, j][hole] = out[:, j - 1][hole]
        for k in range(1, n):
            hole = np.isnan(out[:, :, k])
            out[:, :, k][hole] = out[:, :, k - 1][hole]
        return out


def lattice_to_world(points, resolution, gt_scale, gt_center, scale=1.1):
    """mesh.py:96-98 in float32, as the reference evaluates it"""
    p = np.asarray(points).astype(np.float32)
    p = (p / resolution - 0.5) * scale
    return p * np.float32(gt_scale) + np.asarray(gt_center, dtype=np.float32)


def value_grid(func, verts, level_set=0.0, res_init=32, res_up=3):
    """the MISE part of generate_mesh (mesh.py:80-111): func(world points (n,3) float32) -> values (n,)"""
    verts = np.asarray(verts)
    lo, hi = verts.min(axis=0), verts.max(axis=0)
    gt_center, gt_scale = (lo + hi) * 0.5, (hi - lo).max()
    ex = MISE(res_init, res_up, level_set)
    pts = ex.query()
    n_queries = []
    while pts.shape[0] != 0:
        vals = np.asarray(func(lattice_to_world(pts, ex.resolution, gt_scale, gt_center))).astype(np.float64).reshape(-1)
        ex.update(pts, vals)
        n_queries.append(pts.shape[0])
        pts = ex.query()
    return ex.to_dense(), ex.resolution, gt_scale, gt_center, n_queries


# ---- surface extraction check (the reference calls skimage's marching cubes: third party, absent here) ---------------
CORNER = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]])
EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def marching_cubes_np(volume, level, tri_table):
    """plain-loop marching cubes with the product's case table (multiply_amd/mesh.py build_tri_table) and vertex rule
    (linear interpolation from the edge's lower lattice end, fp32): -> triangle corner positions (T, 3, 3) float32 in
    lattice units and the lattice-edge ids (T, 3) -- a restatement of csrc/mise.hip k_mc_count / k_mc_emit for small grids"""
    vol = np.asarray(volume, dtype=np.float32)
    n = vol.shape[0]
    lvl = np.float32(level)
    tris, ids = [], []
    for x in range(n - 1):
        for y in range(n - 1):
            for z in range(n - 1):
                case = 0
                for k in range(8):
                    if vol[x + CORNER[k][0], y + CORNER[k][1], z + CORNER[k][2]] < lvl:
                        case |= 1 << k
                row = tri_table[case]
                for t in range(5):
                    if row[3 * t] < 0:
                        break
                    tri, tid = [], []
                    for e in row[3 * t:3 * t + 3]:
                        a, b = np.array([x, y, z]) + CORNER[EDGE[e][0]], np.array([x, y, z]) + CORNER[EDGE[e][1]]
                        lo, hi = (a, b) if tuple(a) < tuple(b) else (b, a)
                        v0, v1 = vol[tuple(lo)], vol[tuple(hi)]
                        tpar = np.float32(np.float32(lvl - v0) / np.float32(v1 - v0))
                        axis = int(np.argmax(hi - lo))
                        p = lo.astype(np.float32)
                        p[axis] = np.float32(p[axis] + tpar)
                        tri.append(p)
                        tid.append(3 * ((int(lo[0]) * n + int(lo[1])) * n + int(lo[2])) + axis)
                    tris.append(tri)
                    ids.append(tid)
    return np.asarray(tris, dtype=np.float32).reshape(-1, 3, 3), np.asarray(ids, dtype=np.int64).reshape(-1, 3)


def mesh_topology(ids):
    """from per-corner lattice-edge ids (T,3): (#vertices, #edges, #faces, every directed edge has exactly one opposite
    partner = closed, consistently oriented surface)"""
    ids = np.asarray(ids)
    ids = ids[(ids[:, 0] != ids[:, 1]) & (ids[:, 1] != ids[:, 2]) & (ids[:, 0] != ids[:, 2])]
    directed = {}
    for t in ids:
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            directed[(a, b)] = directed.get((a, b), 0) + 1
    closed = all(c == 1 and directed.get((b, a), 0) == 1 for (a, b), c in directed.items())
    return len(np.unique(ids)), len(directed) // 2, len(ids), closed
