"""Canonical-mesh extraction, CPU side: the MISE restatement (oracle/mise_oracle.py) against golden grids produced by the
reference's own extractor (built from code/lib/libmise/mise.pyx by oracle/Makefile, tests/golden/make_mise_golden.py),
and the derived marching-cubes table."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "mise_golden.npz")


def fields():
    src = open(os.path.join(HERE, "golden", "make_mise_golden.py")).read()
    ns = {}
    exec(src[src.index("def fields()"):src.index("out = {}")], {"np": np}, ns)     # the analytic fields of the golden script
    return ns["fields"]()


def run_extractor(ex, f):
    pts, nq = ex.query(), []
    while pts.shape[0]:
        world = ((np.asarray(pts).astype(np.float32) / ex.resolution - 0.5) * 1.1).astype(np.float64)
        ex.update(pts, f(world).astype(np.float32).astype(np.float64))
        nq.append(pts.shape[0])
        pts = ex.query()
    return nq


def test_oracle_mise_reproduces_the_reference_extractor():
    from oracle.mise_oracle import MISE
    g, F = np.load(GOLD), fields()
    keys = [k[:-6] for k in g.files if k.endswith("_dense")]
    assert len(keys) == 12
    for key in keys:
        name, res0, depth = key.rsplit("_", 2)
        ex = MISE(int(res0), int(depth), 0.0)
        nq = run_extractor(ex, F[name])
        assert nq == g[key + "_nq"].tolist(), key                    # same points queried in every pass
        assert np.array_equal(ex.to_dense(), g[key + "_dense"]), key  # same dense grid, bit for bit


def test_oracle_matches_the_live_reference_build_when_present():
    ref_dir = os.path.join(HERE, "..", "oracle", "_ref")
    sys.path.insert(0, ref_dir)
    try:
        import mise as ref_mise
    except ImportError:
        pytest.skip("oracle/_ref not built (make -C oracle; needs /root/reference)")
    from oracle.mise_oracle import MISE
    rs = np.random.RandomState(3)
    centers, radii = rs.uniform(-0.35, 0.35, (5, 3)), rs.uniform(0.05, 0.2, 5)
    f = lambda p: np.min(np.linalg.norm(p[:, None, :] - centers[None], axis=2) - radii[None], axis=1)
    a, b = ref_mise.MISE(5, 3, 0.0), MISE(5, 3, 0.0)
    assert run_extractor(a, f) == run_extractor(b, f)
    assert np.array_equal(a.to_dense(), b.to_dense())


def test_marching_cubes_table_closes_the_surface():
    from multiply_amd.mesh import build_tri_table
    from oracle.mise_oracle import EDGE, marching_cubes_np, mesh_topology
    table = build_tri_table()
    assert table.shape == (256, 16) and (table[0] < 0).all() and (table[255] < 0).all()
    for case in range(256):
        used = set(int(e) for e in table[case] if e >= 0)
        crossing = {i for i, (a, b) in enumerate(EDGE) if ((case >> a) & 1) != ((case >> b) & 1)}
        assert used == crossing, case                 # a triangle corner on every crossed edge and only there
    # random smooth fields that stay away from the boundary: closed, consistently oriented surfaces, including the
    # ambiguous-face configurations (many small blobs on a coarse lattice)
    rs = np.random.RandomState(0)
    n = 14
    grid = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).astype(np.float64)
    for trial in range(6):
        c, r = rs.uniform(3, n - 4, (7, 3)), rs.uniform(0.8, 2.2, 7)
        vol = np.min(np.linalg.norm(grid[..., None, :] - c, axis=-1) - r, axis=-1) + 0.013 * trial
        tris, ids = marching_cubes_np(vol, 0.0, table)
        V, E, Fc, closed = mesh_topology(ids)
        assert closed and Fc > 50, trial
        # outward orientation: normals point towards increasing values
        nrm = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
        cen = tris.mean(1)
        eps = 1e-2
        f = lambda p: np.min(np.linalg.norm(p[:, None, :] - c[None], axis=2) - r[None], axis=1) + 0.013 * trial
        keep = np.linalg.norm(nrm, axis=1) > 1e-6
        nu = nrm[keep] / np.linalg.norm(nrm[keep], axis=1, keepdims=True)
        rising = f(cen[keep] + eps * nu) - f(cen[keep] - eps * nu)
        assert (rising > 0).mean() > 0.97, (trial, (rising > 0).mean())
    # a sphere: Euler characteristic 2 and the right area
    vol = np.linalg.norm(grid - 6.4, axis=-1) - 4.3
    tris, ids = marching_cubes_np(vol, 0.0, table)
    V, E, Fc, closed = mesh_topology(ids)
    assert closed and V - E + Fc == 2
    area = 0.5 * np.linalg.norm(np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]), axis=1).sum()
    assert abs(area / (4 * np.pi * 4.3 ** 2) - 1) < 0.03


def test_extracted_mesh_object_on_the_host():
    """the trimesh-like surface of generate_mesh's return value (lib/utils/mesh.py:117-131 callers), with host tensors: two
    tetrahedra in one vertex / face list -> components, areas, watertightness, PLY round trip"""
    import torch
    from multiply_amd.mesh import ExtractedMesh
    tet = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float32)
    tf = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]])
    v = np.concatenate([tet, 2.0 * tet + 5.0])
    f = np.concatenate([tf, tf + 4])
    m = ExtractedMesh(torch.tensor(v), torch.tensor(f), resolution=7)
    assert m.vertices.dtype == np.float32 and m.faces.dtype == np.int64 and m.resolution == 7 and m["faces"].shape == (8, 3)
    assert m.is_watertight
    a1 = 1.5 + np.sqrt(3) / 2
    assert abs(m.area - 5 * a1) < 1e-5
    parts = m.split(only_watertight=False)
    assert len(parts) == 2 and sorted(p.vertices.shape[0] for p in parts) == [4, 4]
    big = max(parts, key=lambda p: p.area)
    assert abs(big.area - 4 * a1) < 1e-5 and big.faces.max() == 3 and np.allclose(big.vertices.min(0), 5.0)
    open_mesh = ExtractedMesh(torch.tensor(v), torch.tensor(f[:-1]))
    assert not open_mesh.is_watertight and len(open_mesh.split(only_watertight=True)) == 1
    import os, tempfile
    path = m.export(os.path.join(tempfile.mkdtemp(), "two.ply"))
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n")
    assert b"element vertex 8" in head and b"element face 8" in head
    vv = np.frombuffer(body[:8 * 12], dtype="<f4").reshape(8, 3)
    ff = np.frombuffer(body[8 * 12:], dtype=[("n", "u1"), ("i", "<i4", (3,))])
    assert np.array_equal(vv, v) and np.array_equal(ff["i"], f) and (ff["n"] == 3).all()
    try:
        m.nonexistent
        assert False
    except AttributeError:
        pass
