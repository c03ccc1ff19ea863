"""The minimum-volume oriented box of the parity cull (multiply_amd/obb.py) against the brute-force restatement of the
published algorithm (oracle/obb_oracle.py: explicit 2-D hulls, every hull facet, float64).  CPU only."""
import numpy as np

from multiply_amd.obb import min_volume_obb, obb_record
from oracle.obb_oracle import min_volume_obb_bruteforce, rays_hitting_box


def _same_box(a, b, tol=1e-7):
    """boxes as point sets: same centre, same volume, and every axis of one is an axis (up to sign) of the other with the
    same half extent"""
    ca, aa, ha = a
    cb, ab, hb = b[:3]
    assert np.abs(ca - cb).max() < tol, (ca, cb)
    assert abs(np.prod(ha) - np.prod(hb)) < tol * max(1.0, np.prod(hb))
    m = np.abs(aa @ ab.T)
    for i in range(3):
        j = int(np.argmax(m[i]))
        if abs(ha[i] - hb[j]) > 10 * tol or m[i, j] < 1 - 1e-6:
            # a square cross-section leaves the in-plane rotation free: compare the corner sets instead
            break
    else:
        return
    def corners(c, A, h):
        s = np.array([[i, j, k] for i in (-1, 1) for j in (-1, 1) for k in (-1, 1)], dtype=np.float64)
        return c + (s * h) @ A
    ka, kb = corners(ca, aa, ha), corners(cb, ab, hb)
    d = np.linalg.norm(ka[:, None] - kb[None], axis=2).min(1)
    assert d.max() < 1e-5, d.max()


def test_matches_bruteforce_on_posed_bodies_and_clouds(smpl_tables):
    from oracle import multiply_oracle as O
    from multiply_amd.synthetic import make_scene
    import torch
    sc = make_scene(2, seed=0, H=8, W=8)
    sp = torch.tensor(sc["smpl_params"], dtype=torch.float32)
    T = O.SMPLTables(smpl_tables)
    clouds = []
    for p in range(2):
        sv = O.SMPLServerOracle(T, sp[0, p, 76:].numpy())
        clouds.append(sv.forward(sp[0, p, 0], sp[0, p, 1:4], sp[0, p, 4:76], sp[0, p, 76:])["smpl_verts"].numpy())
    rng = np.random.RandomState(0)
    clouds.append(rng.normal(0, 1, (500, 3)) * np.array([3.0, 1.0, 0.3]))
    R = np.linalg.qr(rng.normal(0, 1, (3, 3)))[0]
    clouds.append((rng.uniform(-1, 1, (2000, 3)) * np.array([0.5, 1.5, 0.25])) @ R)      # a rotated cuboid: the box IS the cuboid
    for c in clouds:
        got = min_volume_obb(c)
        want = min_volume_obb_bruteforce(c)
        assert abs(np.prod(got[2]) * 8 - want[3]) < 1e-9 * max(1.0, want[3])               # same (minimal) volume
        _same_box(got, want)
        proj = (np.asarray(c, np.float64) - got[0]) @ got[1].T
        assert (np.abs(proj) <= got[2] + 1e-9).all()                                       # it bounds the cloud
    # the cuboid is recovered (up to the sampling of its faces)
    c, a, h = min_volume_obb(clouds[3])
    assert np.allclose(np.sort(h), np.sort([0.5, 1.5, 0.25]), atol=0.02)


def test_record_layout_and_ray_hits(smpl_tables):
    rng = np.random.RandomState(1)
    cloud = rng.normal(0, 1, (300, 3)) * np.array([0.3, 0.8, 0.2]) + np.array([0.1, -0.2, 0.4])
    rec = obb_record(cloud, 1.2)
    c, a, h = min_volume_obb(cloud)
    assert rec.dtype == np.float32 and rec.shape == (16,)
    assert np.allclose(rec[:3], c, atol=1e-6) and np.allclose(rec[3:12].reshape(3, 3), a, atol=1e-6) and np.allclose(rec[12:15], 1.2 * h, atol=1e-6)
    cam = np.array([0.0, 0.0, -2.5])
    d = rng.normal(0, 1, (4000, 3)); d[:, 2] = np.abs(d[:, 2]) + 1.0; d /= np.linalg.norm(d, axis=1, keepdims=True)
    hit, margin = rays_hitting_box(cam, d, c, a, 1.2 * h)
    assert 0 < len(hit) < 4000
    # a ray towards a point of the cloud always hits; a ray pointing away never does
    toward = (cloud[:50] - cam); toward /= np.linalg.norm(toward, axis=1, keepdims=True)
    assert len(rays_hitting_box(cam, toward, c, a, 1.2 * h)[0]) == 50
    assert len(rays_hitting_box(cam, -toward, c, a, 1.2 * h)[0]) == 0
