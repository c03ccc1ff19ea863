"""multiply_amd.loss.Loss against the reference's Loss.forward outputs stored in the golden fixture (G9)."""
import numpy as np
import torch

from multiply_amd.config import load_config
from multiply_amd.loss import Loss


def test_loss_matches_reference_golden(golden):
    G = golden
    t = lambda k: torch.tensor(G[k])
    mo = dict(fg_rgb_values_each_person_list=[], rgb_values=t("g9_in_rgb_values"), grad_theta=t("g9_in_grad_theta"),
              acc_map=t("g9_in_acc_map"), index_in_surface=t("g9_in_index_in_surface"), index_off_surface=None, epoch=120,
              temporal_loss=t("g9_in_temporal_loss"), smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1),
              sam_mask=t("g9_in_sam_mask"), acc_person_list=t("g9_in_acc_person_list"))
    lo = Loss(load_config().loss)(mo, dict(rgb=t("g9_gt_rgb")))
    assert set(lo) == {k[7:] for k in G.files if k.startswith("g9_out_")}
    for k, v in lo.items():
        want = G["g9_out_" + k].reshape(-1)
        got = np.asarray(v.detach().numpy(), dtype=np.float32).reshape(-1)
        assert np.allclose(got, want, rtol=1e-6, atol=1e-7), (k, got, want)


def test_loss_nan_rays_are_filtered_and_schedules():
    lf = Loss(load_config().loss)
    R = 16
    rgb = torch.rand(R, 3)
    rgb[3] = float("nan")
    mo = dict(fg_rgb_values_each_person_list=[], rgb_values=rgb, grad_theta=torch.randn(1, 64, 3),
              acc_map=torch.rand(R) * 0.9 + 0.05, index_in_surface=None, index_off_surface=None, epoch=300,
              temporal_loss=torch.tensor([0.5]), smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1),
              acc_person_list=torch.rand(R, 2))
    lo = lf(mo, dict(rgb=torch.rand(1, R, 3)))
    assert torch.isfinite(lo["loss"]).all()
    assert float(lo["in_shape_loss"]) == 0.0 and float(lo["sam_mask_loss"]) == 0.0      # no mask given / no flags
    want = lo["rgb_loss"] + 0.1 * lo["eikonal_loss"] + 5e-3 * lo["bce_loss"] + lo["temporal_loss"]
    assert torch.allclose(lo["loss"], want)


def test_nan_guards_send_no_nan_backwards():
    """loss.py:124-139 replaces a NaN bce / in-shape term by a fresh zero: nothing flows back through it.  The sync-free
    guards must do the same -- an opacity above 1 + eps (or NaN) zeroes the term AND its gradient, finite everywhere."""
    lf = Loss(load_config().loss)
    R = 16
    for poison in (1.00001, float("nan"), -1e-3):
        acc = (torch.rand(R) * 0.9 + 0.05).requires_grad_(True)
        with torch.no_grad():
            acc[5] = poison
        rgb = torch.rand(R, 3, requires_grad=True)
        in_surf = torch.zeros(R, dtype=torch.bool)
        in_surf[4:7] = True                                             # the poisoned ray is inside the in-shape mask
        mo = dict(fg_rgb_values_each_person_list=[], rgb_values=rgb, grad_theta=torch.randn(1, 64, 3),
                  acc_map=acc, index_in_surface=in_surf, index_off_surface=None, epoch=20,
                  temporal_loss=torch.tensor([0.5]), smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1),
                  acc_person_list=torch.rand(R, 2))
        lo = lf(mo, dict(rgb=torch.rand(1, R, 3)))
        assert float(lo["bce_loss"]) == 0.0
        if poison != poison:
            assert float(lo["in_shape_loss"]) == 0.0                    # a NaN inside the mean: zeroed like the reference's
        assert torch.isfinite(lo["loss"]).all()
        lo["loss"].sum().backward()
        assert torch.isfinite(acc.grad).all() and torch.isfinite(rgb.grad).all(), (poison, acc.grad)
        if poison != poison:
            assert float(acc.grad.abs().sum()) == 0.0                   # both opacity terms were replaced
    # clean inputs: the guarded terms are the plain ones, value and gradient
    acc = (torch.rand(R) * 0.9 + 0.05).requires_grad_(True)
    in_surf = torch.zeros(R, dtype=torch.bool)
    in_surf[2:9] = True
    mo = dict(fg_rgb_values_each_person_list=[], rgb_values=torch.rand(R, 3), grad_theta=torch.randn(1, 64, 3), acc_map=acc,
              index_in_surface=in_surf, index_off_surface=None, epoch=20, temporal_loss=torch.zeros(1),
              smpl_surface_loss=torch.zeros(1), zero_pose_loss=torch.zeros(1), acc_person_list=torch.rand(R, 2))
    lo = lf(mo, dict(rgb=torch.rand(1, R, 3)))
    a = acc.detach()
    assert torch.allclose(lo["bce_loss"], -2 * (a * (a + 1e-6).log() + (1 - a) * (1 - a + 1e-6).log()).mean().reshape(1))
    assert torch.allclose(lo["in_shape_loss"], (a[in_surf] - 1).abs().mean().reshape(1))
    empty = dict(mo, index_in_surface=torch.zeros(R, dtype=torch.bool))
    assert float(lf(empty, dict(rgb=torch.rand(1, R, 3)))["in_shape_loss"]) == 0.0      # empty mean -> NaN -> zero


def test_split_input_and_merge_output_round_trip():
    """idr_utils mirrors (reference lib/utils/idr_utils.py:3-30): chunks cover the pixels in order, merge restores them"""
    from multiply_amd.idr_utils import merge_output, split_input
    R = 150
    inp = {"uv": torch.arange(R * 2, dtype=torch.float32).reshape(1, R, 2), "pose": torch.eye(4)[None]}
    chunks = split_input(inp, R, n_pixels=64)
    assert [c["uv"].shape[1] for c in chunks] == [64, 64, 22] and all(c["pose"] is inp["pose"] for c in chunks)
    res = [{"rgb_values": c["uv"][0].repeat(1, 2)[:, :3], "acc_map": c["uv"][0, :, 0], "skipped": None} for c in chunks]
    m = merge_output(res, R, 1)
    assert sorted(m) == ["acc_map", "rgb_values"]
    assert m["acc_map"].shape == (R,) and torch.equal(m["acc_map"], inp["uv"][0, :, 0])
    assert m["rgb_values"].shape == (R, 3) and torch.equal(m["rgb_values"][:, 0], inp["uv"][0, :, 0])
