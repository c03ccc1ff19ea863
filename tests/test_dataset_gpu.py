"""Input producer on the GPU: mp_sample_pixels against golden vectors from the reference's own weighted_sampling /
bilinear_interpolation / edge_sampling (tests/golden/make_dataset_golden.py), and the Hi4DDataset mirror against the CPU
oracle on a sequence written in the reference's on-disk format."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_golden.npz")


def _opt(root, **kw):
    from multiply_amd.config import to_config
    base = dict(data_root=os.path.dirname(root), data_dir=os.path.basename(root), start_frame=0, end_frame=3, num_sample=64,
                using_SAM=False, pixel_per_batch=512)
    base.update(kw)
    return to_config(base)


def test_sample_pixels_matches_the_reference_functions():
    from multiply_amd import hip
    g = np.load(GOLD)
    H, W = g["img"].shape[:2]
    dev = torch.device("cuda")
    img = torch.from_numpy(g["img"]).to(dev)
    mask = torch.from_numpy(g["person"].sum(0).astype(np.uint8)).to(dev)
    sam = torch.from_numpy(g["sam"]).to(dev)
    pos = torch.from_numpy(g["pos"]).to(dev)
    n = pos.shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    rgb, uv, om, ex = torch.empty(n, 3, **f32), torch.empty(n, 2, **f32), torch.empty(n, **f32), torch.empty(n, 2, **f32)
    hip.check(hip.lib().mp_sample_pixels(hip.ptr(img), hip.ptr(mask), hip.ptr(sam), 2, hip.ptr(pos), n, H, W, hip.ptr(rgb),
                                         hip.ptr(uv), hip.ptr(om), hip.ptr(ex), hip.stream()), "mp_sample_pixels")
    torch.cuda.synchronize()
    for name, got, want in (("rgb", rgb, g["rgb"]), ("uv", uv, g["uv"]), ("object_mask", om, g["object_mask"]),
                            ("sam_mask", ex, g["sam_mask"])):
        want32 = want.astype(np.float32)
        err = np.abs(got.cpu().numpy() - want32).max()
        print(f"[parity] sample_pixels {name}: max |err| {err:.3e} (fp32 ulp of the value range {np.spacing(np.float32(np.abs(want32).max())):.1e})")
        assert err <= 2 * np.spacing(np.float32(np.abs(want32).max())), name      # double arithmetic, one rounding
    # argument errors, empty input
    assert hip.lib().mp_sample_pixels(hip.ptr(img), hip.ptr(mask), None, 0, hip.ptr(pos), 0, H, W, hip.ptr(rgb), None, None,
                                      None, hip.stream()) == 0
    assert hip.lib().mp_sample_pixels(hip.ptr(img), hip.ptr(mask), None, 2, hip.ptr(pos), n, H, W, hip.ptr(rgb), None, None,
                                      None, hip.stream()) == -1


def test_dataset_items_match_the_cpu_oracle(tmp_path):
    from multiply_amd.datasets import Hi4DDataset, Hi4DTestDataset, Hi4DValDataset
    from multiply_amd.synthetic import write_sequence
    from oracle.dataset_oracle import Hi4DDatasetOracle
    root = str(tmp_path / "seq")
    w = write_sequence(root, n_frames=4, H=72, W=96, with_edges=True)
    ds = Hi4DDataset(_opt(root, start_frame=1, end_frame=4), rng=np.random.RandomState(11))
    ora = Hi4DDatasetOracle(root, 1, 4, 64)
    assert len(ds) == 3 and tuple(ds.img_size) == (72, 96) and ds.num_person == 2
    assert ds.store.images.dtype == torch.uint8 and ds.store.images.is_cuda and ds.store.images.shape == (3, 72, 96, 3)
    rs = np.random.RandomState(11)
    for idx in (2, 0):
        inputs, images = ds[idx]
        want_in, want_im = ora.__getitem__(idx, rng=rs)
        torch.cuda.synchronize()
        assert inputs["uv"].is_cuda and images["rgb"].is_cuda
        assert np.abs(inputs["uv"].cpu().numpy() - want_in["uv"]).max() <= 1e-5
        assert np.abs(images["rgb"].cpu().numpy() - want_im["rgb"]).max() <= 1.2e-7
        assert np.array_equal(inputs["index_outside"], want_in["index_outside"]) and inputs["idx"] == idx
        assert torch.equal(inputs["smpl_params"], torch.from_numpy(want_in["smpl_params"]))
        assert np.abs(inputs["intrinsics"].numpy() - want_in["intrinsics"]).max() < 1e-4
        assert np.abs(inputs["pose"].numpy() - want_in["pose"]).max() < 1e-6
        assert np.allclose(inputs["P"], want_in["P"]) and np.allclose(inputs["C"], want_in["C"])
        assert inputs["org_img"].shape == (72, 96, 3) and inputs["is_certain"] is True
    # full-frame items (num_sample = 0), the validation and test wrappers
    full = Hi4DDataset(_opt(root, num_sample=0))
    fin, fim = full[1]
    oin, oim = Hi4DDatasetOracle(root, 0, 3, 0)[1]
    assert torch.equal(fin["uv"].cpu(), torch.from_numpy(oin["uv"]))
    assert np.abs(fim["rgb"].cpu().numpy() - oim["rgb"]).max() <= 1.2e-7
    assert np.array_equal(fin["org_object_mask"].cpu().numpy(), oin["org_object_mask"])
    val = Hi4DValDataset(_opt(root, num_sample=0), rng=np.random.RandomState(3))
    vin, vim = val[0]
    assert len(val) == 1 and vin["image_id"] == vin["idx"] and vim["pixel_per_batch"] == 512 and vim["total_pixels"] == 72 * 96
    test = Hi4DTestDataset(_opt(root, num_sample=0))
    tin, tim, ppb, total, i = test[2]
    assert (ppb, total, i) == (512, 72 * 96, 2) and tin["img_size"].tolist() == [72, 96] and "org_img" in tim
    # the model consumes an item as it is (batch dimension added by the DataLoader's collate in the reference)
    assert fin["uv"].shape == (72 * 96, 2) and fin["smpl_params"].shape == (2, 86)


def test_edge_sampling_picks_like_the_reference():
    """edge_sampling (Hi4D.py:28-56): same randint stream -> same pixels; checked through the dataset's code path on the
    golden frame (the reference function's outputs are in the golden file)."""
    g = np.load(GOLD)
    H, W = g["img"].shape[:2]
    n = int(g["edge_n"])
    mask = g["person"].sum(0)
    edge = np.logical_and(mask, g["edge"])
    rng = np.random.RandomState(int(g["edge_seed"]))
    n_mask, n_edge = int(n * 0.5), int(n * 0.4)
    mask_loc, edge_loc = np.where(mask.reshape(-1))[0], np.where(edge.reshape(-1))[0]
    pick = np.concatenate([mask_loc[rng.randint(0, len(mask_loc), n_mask)], edge_loc[rng.randint(0, len(edge_loc), n_edge)],
                           rng.randint(0, H * W, n - n_mask - n_edge)])
    dev = torch.device("cuda")
    img = torch.from_numpy(g["img"]).to(dev).float() / 255
    rows, cols = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    uv = torch.stack([cols, rows], -1).float().reshape(-1, 2)
    p = torch.from_numpy(pick).to(dev)
    assert np.abs(uv[p].cpu().numpy() - g["edge_uv"]).max() == 0
    assert np.abs(img.reshape(-1, 3)[p].cpu().numpy() - g["edge_rgb"].astype(np.float32)).max() <= 1.2e-7


def test_sequence_on_disk_to_training_step(tmp_path):
    """disk -> Hi4DDataset -> DataLoader(num_workers=0) -> the reference's training_step input preparation
    (multiply_model.py:162-192) -> Multiply.forward (train) -> Loss -> backward"""
    import warnings
    warnings.filterwarnings("ignore")
    from multiply_amd.config import load_config
    from multiply_amd.datasets import Hi4DDataset
    from multiply_amd.loss import Loss
    from multiply_amd.multiply import Multiply
    from multiply_amd.synthetic import make_smpl_tables, write_sequence
    root = str(tmp_path / "seq")
    w = write_sequence(root, n_frames=3, H=64, W=64)
    ds = Hi4DDataset(_opt(root, num_sample=96), rng=np.random.RandomState(2))
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)
    opt = load_config()
    torch.manual_seed(0)
    model = Multiply(opt, w["shape"], smpl_tables=make_smpl_tables(0)).train()
    loss_fn = Loss(opt.loss)
    inputs, targets = next(iter(loader))
    assert inputs["uv"].shape == (1, 96, 2) and inputs["uv"].is_cuda and targets["rgb"].shape == (1, 96, 3)
    assert inputs["smpl_params"].shape == (1, 2, 86) and inputs["intrinsics"].shape == (1, 4, 4)
    inputs["smpl_pose"] = inputs["smpl_params"][..., 4:76]
    inputs["smpl_shape"] = inputs["smpl_params"][..., 76:]
    inputs["smpl_trans"] = inputs["smpl_params"][..., 1:4]
    inputs["smpl_pose_last"] = inputs["smpl_pose"]      # the opt_smpl branch provides it (multiply_model.py:172-180)
    inputs["current_epoch"] = 301
    out = model(inputs)
    loss = loss_fn(out, targets)["loss"]
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and out["rgb_values"].shape == (96, 3)
    g = model.foreground_implicit_network_list[0].lin0.weight_v.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
    assert out["index_outside"] is not None


def test_novel_view_test_items_and_the_single_person_layout(tmp_path):
    """Hi4DTestDataset's novel-view branch (Hi4D.py:373-450) on a written sequence + a written studio-camera file, and the
    ThreeDPW* variants (threedpw.py:60-243) on the same frames stored in the single-person layout; create_dataset's factory."""
    import shutil
    from multiply_amd.datasets import (Hi4DDataset, Hi4DTestDataset, ThreeDPWDataset, ThreeDPWTestDataset, ThreeDPWValDataset,
                                       create_dataset, find_dataset_using_name)
    from multiply_amd.synthetic import write_sequence
    from oracle.dataset_oracle import Hi4DDatasetOracle, novel_view_camera
    root = str(tmp_path / "seq")
    write_sequence(root, n_frames=3, H=48, W=64, num_person=1)
    gt = tmp_path / "gt" / "pair00" / "dance" / "cameras"
    gt.mkdir(parents=True)
    rs = np.random.RandomState(1)
    K = np.stack([np.array([[190.0 + 10 * i, 0, 60.0], [0, 190.0, 50.0 + i], [0, 0, 1.0]]) for i in range(3)])
    E = []
    for i in range(3):
        q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        E.append(np.concatenate([q * np.sign(np.linalg.det(q)), rs.normal(size=(3, 1)) + [[0], [0], [3.0]]], 1))
    np.savez(str(gt / "rgb_cameras.npz"), ids=np.array([4, 16, 28]), intrinsics=K, extrinsics=np.stack(E))
    test = Hi4DTestDataset(_opt(root, num_sample=0, novel_view=28, current_view=4, pair="pair00", action="dance",
                                GT_DIR=str(tmp_path / "gt")))
    tin, tim, ppb, total, i = test[1]
    cams = np.load(os.path.join(root, "cameras_normalize.npz"))
    P, C, intr, pose = novel_view_camera(cams["scale_mat_1"].astype(np.float32), cams["world_mat_1"].astype(np.float32),
                                         K[0], E[0], K[2], E[2])
    assert np.allclose(tin["P"], P) and np.allclose(tin["C"], C) and tin["novel_view"] == 28 and set(tim) == {"rgb", "img_size"}
    assert np.abs(tin["intrinsics"].numpy() - intr).max() < 1e-4 and np.abs(tin["pose"].numpy() - pose).max() < 1e-6
    assert (ppb, total, i) == (512, 48 * 64, 1) and tin["uv"].shape == (48 * 64, 2)
    plain = Hi4DTestDataset(_opt(root, num_sample=0))
    assert plain.novel_view is None and "org_img" in plain[1][1]
    # single-person layout: mask/*.png, (F,72) poses, (F,3) translations, (10,) shape
    root1 = str(tmp_path / "seq1")
    shutil.copytree(root, root1)
    for f in sorted(os.listdir(os.path.join(root1, "mask", "0"))):
        shutil.move(os.path.join(root1, "mask", "0", f), os.path.join(root1, "mask", f))
    os.rmdir(os.path.join(root1, "mask", "0"))
    for name in ("poses", "normalize_trans"):
        np.save(os.path.join(root1, name + ".npy"), np.load(os.path.join(root1, name + ".npy"))[:, 0])
    np.save(os.path.join(root1, "mean_shape.npy"), np.load(os.path.join(root1, "mean_shape.npy"))[0])
    ds = ThreeDPWDataset(_opt(root1), rng=np.random.RandomState(4))
    ref = Hi4DDataset(_opt(root), rng=np.random.RandomState(4))
    (i1, t1), (i2, t2) = ds[2], ref[2]
    torch.cuda.synchronize()
    assert torch.equal(i1["uv"], i2["uv"]) and torch.equal(t1["rgb"], t2["rgb"])            # same draws, same frame
    assert i1["smpl_params"].shape == (86,) and torch.equal(i1["smpl_params"], i2["smpl_params"][0])
    assert set(i1) == {"uv", "P", "C", "intrinsics", "pose", "smpl_params", "index_outside", "idx"} and set(t1) == {"rgb"}
    want_in, want_im = Hi4DDatasetOracle(root, 0, 3, 64).__getitem__(2, rng=np.random.RandomState(4))
    assert np.abs(t1["rgb"].cpu().numpy() - want_im["rgb"]).max() <= 1.2e-7
    v_in, v_im = ThreeDPWValDataset(_opt(root1, num_sample=0), rng=np.random.RandomState(0))[0]
    assert v_in["uv"].shape == (48 * 64, 2) and v_im["total_pixels"] == 48 * 64 and "image_id" in v_in
    t_in, t_im, ppb, total, i = ThreeDPWTestDataset(_opt(root1, num_sample=0))[0]
    assert set(t_in) == {"uv", "P", "C", "intrinsics", "pose", "smpl_params", "idx"} and set(t_im) == {"rgb", "img_size"}
    assert find_dataset_using_name("Hi4DVal").__name__ == "Hi4DValDataset"
    with pytest.raises(ValueError):
        find_dataset_using_name("nope")
    loader = create_dataset(_opt(root1, dataset="ThreeDPW", batch_size=1, drop_last=False, shuffle=False, worker=0))
    b_in, b_im = next(iter(loader))
    assert b_in["uv"].shape == (1, 64, 2) and b_in["smpl_params"].shape == (1, 86) and b_im["rgb"].is_cuda
