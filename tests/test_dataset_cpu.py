"""Input producer, CPU side: the oracle (oracle/dataset_oracle.py) against golden vectors produced by the reference's own
functions (tests/golden/make_dataset_golden.py), the host logic of multiply_amd/datasets.py, the camera decomposition."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _data(g):
    H, W = g["img"].shape[:2]
    uv = np.flip(np.mgrid[:H, :W].astype(np.int32), axis=0).copy().transpose(1, 2, 0).astype(np.float32)
    return {"rgb": g["img"] / 255, "uv": uv, "object_mask": g["person"].sum(0), "sam_mask": g["sam"]}, (H, W)


def test_oracle_weighted_sampling_reproduces_the_reference(gold):
    from oracle import dataset_oracle as O
    data, size = _data(gold)
    np.random.seed(int(gold["seed"]))
    out, outside = O.weighted_sampling(data, size, int(gold["n"]))
    assert np.array_equal(outside, gold["index_outside"])
    for k in ("rgb", "uv", "object_mask", "sam_mask"):
        assert out[k].shape == gold[k].shape and np.abs(out[k] - gold[k]).max() < 1e-12, k


def test_position_draws_consume_the_random_stream_like_the_reference(gold):
    from multiply_amd.datasets import draw_positions
    data, size = _data(gold)
    where = np.asarray(np.where(data["object_mask"]))
    np.random.seed(int(gold["seed"]))
    pos, outside = draw_positions(where.min(1), where.max(1), size, int(gold["n"]))
    assert np.array_equal(pos, gold["pos"]) and np.array_equal(outside, gold["index_outside"])
    assert pos[:, 0].max() < size[0] - 1 and pos[:, 1].max() < size[1] - 1      # x1 + 1, y1 + 1 stay inside


def test_camera_decomposition_recomposes_the_projection():
    from multiply_amd.datasets import load_K_Rt_from_P as product
    from oracle.dataset_oracle import load_K_Rt_from_P as oracle
    rs = np.random.RandomState(0)
    for trial in range(20):
        A = rs.normal(size=(3, 3))
        q, _ = np.linalg.qr(A)
        R = q * np.sign(np.linalg.det(q))
        K = np.array([[700 + 100 * rs.rand(), 2 * rs.normal(), 250 + 20 * rs.rand()],
                      [0, 690 + 100 * rs.rand(), 260 + 20 * rs.rand()], [0, 0, 1.0]])
        C = rs.normal(size=3)
        P = K @ np.concatenate([R, (-R @ C)[:, None]], 1) * (1.0 if trial % 2 else -2.5)    # P is homogeneous
        for fn in (product, oracle):
            intr, pose = fn(P)
            assert np.abs(intr[:3, :3] - K).max() < 1e-6 * 700
            assert np.abs(pose[:3, :3] - R.T).max() < 1e-6 and np.abs(pose[:3, 3] - C).max() < 1e-5
            assert intr[0, 0] > 0 and intr[1, 1] > 0 and abs(intr[2, 2] - 1) < 1e-12


def test_gray_threshold_follows_the_fixed_point_luma():
    from multiply_amd.datasets import gray_nonzero
    from oracle.dataset_oracle import bgr2gray
    px = np.array([[[0, 0, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0], [255, 255, 255], [3, 0, 0]]], dtype=np.uint8)   # RGB
    assert gray_nonzero(px).tolist() == [[False, False, False, True, True, True]]
    assert (bgr2gray(px) > 0).tolist() == gray_nonzero(px).tolist()
    assert int(bgr2gray(px)[0, 4]) == 255


def test_oracle_dataset_on_a_written_sequence(tmp_path):
    from multiply_amd.synthetic import write_sequence
    from oracle.dataset_oracle import Hi4DDatasetOracle
    w = write_sequence(str(tmp_path), n_frames=3, H=40, W=48)
    ds = Hi4DDatasetOracle(str(tmp_path), 1, 3, 64)
    assert len(ds) == 2 and ds.img_size == (40, 48) and ds.num_person == 2
    img, mask = ds.frame(0)
    assert np.array_equal((img * 255).round().astype(np.uint8), w["images"][1])
    assert np.array_equal(mask, w["masks"][1].sum(0))
    np.random.seed(5)
    inputs, images = ds[0]
    assert inputs["uv"].shape == (64, 2) and images["rgb"].shape == (64, 3) and inputs["uv"].dtype == np.float32
    assert np.allclose(inputs["smpl_params"][:, 4:76], w["poses"][1], atol=1e-6) and np.all(inputs["smpl_params"][:, 0] == 1)
    assert np.abs(inputs["intrinsics"] - w["intrinsics"]).max() < 1e-4 and np.abs(inputs["pose"] - w["pose"]).max() < 1e-6
    full = Hi4DDatasetOracle(str(tmp_path), 0, 3, 0)[2]
    assert full[0]["uv"].shape == (40 * 48, 2) and np.array_equal(full[0]["uv"][49], [1, 1])


def test_novel_view_cameras_match_the_restatement_and_a_directly_built_camera():
    """Hi4D.py:398-425.  Studio frame = a rigid motion of the training frame; the novel camera must image a point exactly
    like the target studio camera images the same physical point, at the training image scale."""
    from multiply_amd.datasets import novel_view_camera
    from oracle.dataset_oracle import novel_view_camera as want_fn
    rs = np.random.RandomState(5)

    def rot():
        q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        return q * np.sign(np.linalg.det(q))
    K_tr = np.array([[800.0, 0, 310.0], [0, 800.0, 250.0], [0, 0, 1.0]])
    R_tr, t_tr = rot(), rs.normal(size=3) + [0, 0, 4.0]
    world = np.eye(4); world[:3, :4] = K_tr @ np.concatenate([R_tr, t_tr[:, None]], 1)
    scale_mat = np.diag([1.3, 1.3, 1.3, 1.0]); scale_mat[:3, 3] = [0.1, -0.2, 0.05]
    # studio coordinates x_s = A x_w + b ; studio cameras are 2x larger images (zoom = 2)
    A, b = rot(), rs.normal(size=3)
    K_cur = np.diag([2.0, 2.0, 1.0]) @ K_tr
    E_cur = np.concatenate([R_tr @ A.T, (t_tr - R_tr @ A.T @ b)[:, None]], 1)            # same physical camera
    K_tgt = np.array([[1700.0, 0, 600.0], [0, 1650.0, 520.0], [0, 0, 1.0]])
    R_t, t_t = rot(), rs.normal(size=3) + [0, 0, 5.0]
    E_tgt = np.concatenate([R_t, t_t[:, None]], 1)
    got = novel_view_camera(scale_mat, world, (K_cur, E_cur), (K_tgt, E_tgt))
    want = want_fn(scale_mat, world, K_cur, E_cur, K_tgt, E_tgt)
    for g, w, name in zip(got, want, ("P", "C", "intrinsics", "pose")):
        assert np.allclose(g, w, rtol=1e-9, atol=1e-9), name
    # a point in normalised coordinates x_n -> training world x_w = scale_mat x_n -> studio -> target pixel / zoom
    xn = rs.normal(size=(5, 3)) * 0.3
    xw = xn @ scale_mat[:3, :3].T + scale_mat[:3, 3]
    xs = xw @ A.T + b
    pix = (xs @ R_t.T + t_t) @ K_tgt.T
    pix = pix[:, :2] / pix[:, 2:3] / 2.0
    mine = np.concatenate([xn, np.ones((5, 1))], 1) @ got[0][:3].T
    assert np.allclose(mine[:, :2] / mine[:, 2:3], pix, atol=1e-8)
    # novel view == current view: the training camera itself
    same = novel_view_camera(scale_mat, world, (K_cur, E_cur), (K_cur, E_cur))
    P0 = world @ scale_mat
    assert np.allclose(same[0][:3] / same[0][2, 3], P0[:3] / P0[2, 3], atol=1e-8)
