"""Canonical-mesh extraction on the GPU: the dense-lattice MISE (csrc/mise.hip) against golden grids from the reference's
own extractor, the marching-cubes kernels against the plain-loop restatement, and generate_mesh end to end."""
import os

import numpy as np
import pytest
import torch

from tests.test_mesh_cpu import GOLD, fields

pytestmark = pytest.mark.gpu


def _run(ex, f):
    pts = ex.query()
    while pts.shape[0]:
        p = pts.cpu().numpy()
        world = ((p.astype(np.float32) / ex.resolution - 0.5) * 1.1).astype(np.float64)
        ex.update(pts, torch.from_numpy(f(world).astype(np.float32)))
        pts = ex.query()
    return ex.n_queried


def test_device_mise_reproduces_the_reference_extractor():
    from multiply_amd.mesh import MISE
    g, F = np.load(GOLD), fields()
    for key in [k[:-6] for k in g.files if k.endswith("_dense")]:
        name, res0, depth = key.rsplit("_", 2)
        ex = MISE(int(res0), int(depth), 0.0)
        nq = _run(ex, F[name])
        torch.cuda.synchronize()
        assert nq == g[key + "_nq"].tolist(), key
        dense = ex.to_dense().cpu().numpy()
        assert dense.dtype == np.float32 and np.array_equal(dense.astype(np.float64), g[key + "_dense"]), key


def test_marching_cubes_kernels_match_the_plain_loop_restatement():
    from multiply_amd.mesh import build_tri_table, marching_cubes
    from oracle.mise_oracle import marching_cubes_np, mesh_topology
    rs = np.random.RandomState(1)
    n = 13
    grid = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).astype(np.float64)
    c, r = rs.uniform(3, n - 4, (6, 3)), rs.uniform(0.8, 2.0, 6)
    vol = (np.min(np.linalg.norm(grid[..., None, :] - c, axis=-1) - r, axis=-1)).astype(np.float32)
    verts, faces = marching_cubes(torch.from_numpy(vol).cuda(), 0.0)
    torch.cuda.synchronize()
    want, ids = marching_cubes_np(vol, 0.0, build_tri_table())
    got = verts[faces].cpu().numpy()                                  # (T, 3, 3)
    ids = ids[(ids[:, 0] != ids[:, 1]) & (ids[:, 1] != ids[:, 2]) & (ids[:, 0] != ids[:, 2])]
    keep = np.ones(len(want), bool)
    assert got.shape[0] == ids.shape[0]
    key = lambda t: sorted(map(tuple, t.reshape(-1, 9).tolist()))
    wk = marching_cubes_np(vol, 0.0, build_tri_table())[0]
    wid = marching_cubes_np(vol, 0.0, build_tri_table())[1]
    wk = wk[(wid[:, 0] != wid[:, 1]) & (wid[:, 1] != wid[:, 2]) & (wid[:, 0] != wid[:, 2])]
    assert key(got) == key(wk)                                        # the same triangles, bit for bit
    V, E, Fc, closed = mesh_topology(ids)
    assert closed and verts.shape[0] == V and faces.shape[0] == Fc
    # welded mesh: every undirected edge is shared by exactly two faces
    f = faces.cpu().numpy()
    und = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(und, axis=0, return_counts=True)
    assert (cnt == 2).all()


def test_generate_mesh_sphere_and_torus():
    from multiply_amd.mesh import generate_mesh
    box = torch.tensor([[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]]).cuda()
    c = torch.tensor([0.03, -0.02, 0.05]).cuda()
    m = generate_mesh(lambda p: (p - c).norm(dim=1) - 0.31, box, 0.0, res_init=16, res_up=2)
    v, f = m["vertices"], m["faces"]
    torch.cuda.synchronize()
    assert m["resolution"] == 64 and len(m["n_queried"]) >= 3 and m["n_queried"][0] == 17 ** 3
    assert sum(m["n_queried"]) < 0.15 * 65 ** 3                       # refinement only near the surface
    rad = (v - c).norm(dim=1)
    assert float((rad - 0.31).abs().max()) < 2e-4                     # linear interpolation of a smooth field at h = 1.1/64
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(dim=1).values.unique(dim=0)
    assert v.shape[0] - e.shape[0] + f.shape[0] == 2                  # a sphere
    nrm = torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=1)
    out = ((v[f].mean(1) - c) * nrm).sum(1)
    assert bool((out >= 0).all()) and float((out > 0).double().mean()) > 0.999     # outward winding
    area = 0.5 * nrm.norm(dim=1).sum()
    assert abs(float(area) / (4 * np.pi * 0.31 ** 2) - 1) < 5e-3
    # torus: genus 1; plus a far-away small sphere that the largest-component filter drops
    def field(p):
        q = torch.stack([p[:, [0, 2]].norm(dim=1) - 0.25, p[:, 1]], 1).norm(dim=1) - 0.07
        s = (p - torch.tensor([0.4, 0.4, 0.4], device=p.device)).norm(dim=1) - 0.05
        return torch.minimum(q, s)
    m = generate_mesh(field, box, 0.0, res_init=16, res_up=2, point_batch=10000)
    v, f = m["vertices"], m["faces"]
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(dim=1).values.unique(dim=0)
    assert v.shape[0] - e.shape[0] + f.shape[0] == 0
    assert float(v[:, 1].abs().max()) < 0.08                          # the small sphere at y = 0.4 is gone


def test_canonical_mesh_of_the_model_feeds_the_surface_flags():
    """refresh_canonical_meshes (multiply_model.py:497-506) through the fused SDF kernel, then a training forward at an epoch
    < 250 that reads mesh_face_vertices_list (multiply.py:153-167)"""
    from multiply_amd import hip
    from multiply_amd.mesh import canonical_mesh, refresh_canonical_meshes
    from tests.test_render_gpu import build
    model, oracle, inp = build(H=9, W=9)
    m = canonical_mesh(model, 0, res_up=2)
    v, f = m["vertices"], m["faces"]
    assert m["resolution"] == 128 and f.shape[0] > 1000
    sdf = hip.implicit_sdf(model.foreground_implicit_network_list[0], v.contiguous(), torch.zeros(69, device=v.device))
    torch.cuda.synchronize()
    print(f"[parity] canonical mesh: {v.shape[0]} vertices, {f.shape[0]} faces, passes {m['n_queried']}, "
          f"max |sdf(vertex)| {float(sdf.abs().max()):.2e}")
    assert float(sdf.abs().max()) < 2e-3                              # on the zero level set of the (f16) network
    # manifold: interior edges are shared by exactly two faces; the only open edges lie on the box boundary (the
    # geometric-initialisation sphere is larger than the synthetic body's bounding cube along y)
    e, cnt = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(dim=1).values.unique(dim=0, return_counts=True)
    assert int(cnt.max()) == 2
    lv = m["lattice_vertices"]
    on_box = ((lv == 0) | (lv == m["resolution"])).any(dim=1)
    assert bool(on_box[e[cnt == 1]].all())
    vs, fs = refresh_canonical_meshes(model, res_up=1)
    assert len(vs) == 2 and model.mesh_face_vertices_list[1].shape[1:] == (fs[1].shape[0], 3, 3)
    model.train()
    gin = {k: (t.cuda() if torch.is_tensor(t) else t) for k, t in inp.items()}
    gin.update(current_epoch=100, index_outside=torch.zeros(81, dtype=torch.bool))
    out = model(gin)
    torch.cuda.synchronize()
    assert out["index_off_surface"].dtype == torch.bool and out["index_in_surface"].shape == (81,)
    assert bool(out["index_in_surface"].any()) and torch.isfinite(out["rgb_values"]).all()


def test_generate_mesh_returns_the_reference_mesh_interface(tmp_path):
    """lib/utils/mesh.py:117-131: callers read `.vertices` / `.faces` (numpy), split into components, pick the largest area,
    export a .ply; the object also keeps the dictionary access of the device tensors.  Watertightness on the model's own
    canonical SDF at the validation resolution (res_up = 4, 513^3 lattice)."""
    import numpy as np
    from multiply_amd.mesh import ExtractedMesh, canonical_mesh, generate_mesh
    box = torch.tensor([[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]])

    def two_blobs(p):                                            # a big and a small sphere: two components
        a = (p - torch.tensor([-0.15, 0.0, 0.0], device=p.device)).norm(dim=1) - 0.2
        b = (p - torch.tensor([0.3, 0.0, 0.0], device=p.device)).norm(dim=1) - 0.08
        return torch.minimum(a, b)
    m = generate_mesh(two_blobs, box, 0.0, res_init=16, res_up=2)
    assert isinstance(m, ExtractedMesh) and isinstance(m.vertices, np.ndarray) and m.faces.dtype == np.int64
    assert m.vertices.shape[1] == 3 and m.faces.shape[1] == 3 and m.faces.max() == m.vertices.shape[0] - 1
    assert torch.is_tensor(m["vertices"]) and m["vertices"].is_cuda and m.resolution == 64       # earlier dictionary access
    assert m.is_watertight and len(m.split(only_watertight=False)) == 1                          # the largest component only
    assert abs(m.area - 4 * np.pi * 0.2 ** 2) < 0.01 * 4 * np.pi * 0.2 ** 2
    assert np.abs(np.linalg.norm(m.vertices - np.array([-0.15, 0, 0]), axis=1) - 0.2).max() < 2e-3
    path = m.export(str(tmp_path / "m.ply"))
    head = open(path, "rb").read(200).decode("latin1")
    assert head.startswith("ply") and f"element vertex {m.vertices.shape[0]}" in head and f"element face {m.faces.shape[0]}" in head
    # the model's canonical surface at the validation / test resolution of the reference (res_up = 4)
    from tests.test_render_gpu import build
    model, _, _ = build(H=8, W=8)
    cm = canonical_mesh(model, 0, res_up=4)
    assert cm.resolution == 512 and cm.value_grid.shape == (513, 513, 513)
    # the geometric-init sphere (radius ~0.6 about the origin) pokes through the top of the canonical body's bounding cube:
    # the surface is closed and consistently oriented EXCEPT where it leaves the box -- every unmatched edge lies on a box face
    f, V = cm.faces, cm.vertices.shape[0]
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key, rev = e[:, 0] * (V + 1) + e[:, 1], e[:, 1] * (V + 1) + e[:, 0]
    assert np.unique(key).shape[0] == key.shape[0]                                # no edge is traversed twice the same way
    open_edges = e[~np.isin(key, rev)]
    lv = cm.lattice_vertices.cpu().numpy()
    on_face = ((lv <= 1e-4) | (lv >= cm.resolution - 1e-4)).any(axis=1)
    assert on_face[open_edges].all(), "an open edge inside the box: the surface has a hole"
    # vertices lie on the level set of the network they were extracted from
    from multiply_amd import hip
    sdf_v = hip.implicit_sdf(model.foreground_implicit_network_list[0], cm["vertices"][~torch.from_numpy(on_face).cuda()],
                             torch.zeros(69, device="cuda"))
    assert float(sdf_v.abs().max()) < 2e-3                                        # lattice spacing 3.7e-3, linear interpolation
    print(f"[parity] canonical mesh at 513^3: {V} vertices, {f.shape[0]} faces, {open_edges.shape[0]} open edges, all on the box")
