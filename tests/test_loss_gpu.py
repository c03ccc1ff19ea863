"""The fused objective kernel (csrc/loss.hip mp_loss_fused, multiply_amd/loss.py _forward_fused) against the torch statement of
the same terms on the same device tensors -- the statement tests/test_loss_cpu.py pins with the reference's own Loss.forward
outputs (golden G9 / T3): every entry of the output dict and the gradient w.r.t. every differentiable model output, over the
schedules and the reference's NaN / empty-set / "keep the first element" branches (loss.py:61-78, 120-139)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _outputs(R=512, P=2, N=1024, seed=0, epoch=301, sam=True, in_surface=False, nan_ray=False, nan_acc=False, sam_all_agree=False,
             empty_in=False):
    g = torch.Generator().manual_seed(seed)
    dev = "cuda"
    rgb = torch.rand(R, 3, generator=g)
    accp = torch.rand(R, P, generator=g) * 0.5
    accp[::7] = torch.tensor([0.99] + [0.0] * (P - 1))
    accp[1::7] = 0.0
    acc = accp.sum(1).clamp(0, 1)
    gth = torch.randn(1, N, 3, generator=g)
    gth[0, 5] = 0.0                                              # a zero gradient vector: the norm's subgradient
    if nan_ray:
        rgb[3, 1] = float("nan")
        rgb[9, 0] = float("inf")
    if nan_acc:
        acc[11] = float("nan")
    mo = {"rgb_values": rgb, "acc_map": acc, "acc_person_list": accp, "grad_theta": gth, "epoch": epoch,
          "fg_rgb_values_each_person_list": [], "index_in_surface": None, "temporal_loss": torch.full((1,), 3e-4),
          "smpl_surface_loss": torch.zeros(1), "zero_pose_loss": torch.zeros(1)}
    if sam:
        logits = torch.randn(R, P, generator=g) * 4
        if sam_all_agree:                                        # every selected element agrees: the "first element" fallback
            logits = torch.where(accp > 0.5, torch.full_like(accp, 9.0), torch.full_like(accp, -9.0))
            mo["acc_person_list"] = accp = torch.where(accp > 0.5, torch.full_like(accp, 0.99), torch.full_like(accp, 0.01))
        mo["sam_mask"] = logits
    if in_surface:
        m = torch.rand(R, generator=g) < (0.0 if empty_in else 0.3)
        mo["index_in_surface"] = m
    mo = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in mo.items()}
    gt = {"rgb": torch.rand(1, R, 3, generator=g).to(dev)}
    return mo, gt


CASES = {
    "epoch301_sam": dict(),
    "epoch100_in_surface_sam": dict(epoch=100, in_surface=True),
    "epoch10_no_sam_yet": dict(epoch=10, in_surface=True),
    "nan_ray_and_inf": dict(nan_ray=True),
    "nan_in_acc_zeroes_bce": dict(nan_acc=True, in_surface=True, epoch=100),
    "empty_in_surface_set": dict(in_surface=True, empty_in=True, epoch=50),
    "sam_all_agree_first_element": dict(sam_all_agree=True),
    "three_persons": dict(P=3, seed=4),
    "no_sam_key": dict(sam=False),
}


@pytest.mark.parametrize("name", list(CASES))
def test_fused_loss_matches_the_torch_statement(name, monkeypatch):
    from multiply_amd import loss as LM
    from multiply_amd.config import load_config
    loss_fn = LM.Loss(load_config().loss)
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(LM, "FUSED", fused)
        mo, gt = _outputs(**CASES[name])
        leaves = {k: mo[k].clone().requires_grad_(True) for k in ("rgb_values", "acc_map", "acc_person_list", "grad_theta")}
        mo.update(leaves)
        out = loss_fn(mo, gt)
        out["loss"].sum().backward()
        res[fused] = ({k: v.detach().float().reshape(-1).cpu() for k, v in out.items() if torch.is_tensor(v)},
                      {k: (v.grad.detach().cpu() if v.grad is not None else torch.zeros_like(v).cpu()) for k, v in leaves.items()})
    (want, gwant), (got, ggot) = res[False], res[True]
    assert set(got) == set(want)
    for k in want:
        a, b = got[k], want[k]
        assert a.shape == b.shape or a.numel() == b.numel(), k
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-7, equal_nan=True), (name, k, a, b)
    for k in gwant:
        a, b = ggot[k], gwant[k]
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-9, equal_nan=True), (name, k, float((a - b).abs().nan_to_num().max()))
    print(f"[parity] fused loss, case {name}: loss {float(got['loss']):.6f} (torch {float(want['loss']):.6f}), "
          f"max |d rgb| diff {float((ggot['rgb_values'] - gwant['rgb_values']).abs().nan_to_num().max()):.1e}")
