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
