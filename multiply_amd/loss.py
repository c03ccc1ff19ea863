"""Training objective (reference code/lib/model/loss.py:6-177), same constructor keys, same output dict.

These are a handful of reductions over <= 512 rays / 1024 eikonal points, so they are plain torch ops; their autograd
adjoints (d rgb_values, d acc_map, d acc_person_list, d grad_theta) are what `Multiply.forward`'s hand-written
backward (multiply_amd/train.py) consumes.  Terms:

  rgb_loss       mean |rgb - gt| over rays without NaN                                  (loss.py:31-33, 120-122)
  eikonal_loss   mean (|grad_theta| - 1)^2                                              (loss.py:36-38)
  bce_loss       -2 mean(a log(a+eps) + (1-a) log(1-a+eps)),  zeroed if NaN            (loss.py:41-43, 124-128)
  in_shape_loss  mean |acc[index_in_surface] - 1|, weight fades to 0 at epoch 200       (loss.py:51-53, 131-139, 161)
  sam_mask_loss  clipped L1 between acc_person and sigmoid(sam logits)                  (loss.py:61-78, 143-146)
  temporal/smpl_surface/zero_pose/depth_order: passed through with their schedules     (loss.py:141-158)
"""
import ctypes as C
import os

import torch
from torch import nn

# The per-ray terms and their adjoints in ONE HIP launch (csrc/loss.hip mp_loss_fused) when the model outputs live on the GPU
# (the training path always does); MP_FUSED_LOSS=0: the torch statement below everywhere (the cross-check, tests/test_loss_gpu.py).
FUSED = os.environ.get("MP_FUSED_LOSS", "1") != "0"


class MpLossArgs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("rgb", "rgb_gt", "acc", "accp", "gth", "sam", "in_mask", "d_rgb", "d_acc", "d_accp",
                                           "d_gth", "terms")] + \
               [("n_rays", C.c_int), ("n_persons", C.c_int), ("n_eik", C.c_int)] + \
               [(k, C.c_float) for k in ("w_eik", "w_bce", "w_in", "w_sam", "eps")]


class _FusedTerms(torch.autograd.Function):
    """(rgb_values, acc_map, acc_person_list, grad_theta) -> (total, terms): the kernel evaluates the terms AND the gradient of
    their weighted sum; backward scales the stored gradient by d loss / d total.  `terms` (the individual losses the trainer
    logs) is not differentiable: only `loss` is ever back-propagated (multiply_model.py:212-217)."""

    @staticmethod
    def forward(ctx, rgb, acc, accp, gth, rgb_gt, sam, in_mask, w):
        from . import hip
        ctx.shapes = (rgb.shape, acc.shape, accp.shape, gth.shape)
        f = lambda t: t.detach().contiguous().float()
        rgb, acc, accp, gth, rgb_gt = f(rgb).reshape(-1, 3), f(acc).reshape(-1), f(accp), f(gth).reshape(-1, 3), f(rgb_gt).reshape(-1, 3)
        R, N = rgb.shape[0], gth.shape[0]
        accp = accp.reshape(R, -1)
        P = accp.shape[1]
        sam = None if sam is None else f(sam).reshape(R, P)
        in_mask = None if in_mask is None else in_mask.detach().reshape(-1).to(torch.uint8).contiguous()
        buf = torch.empty(8 + 4 * R + R * P + 3 * N, dtype=torch.float32, device=rgb.device)      # terms + the four gradients
        terms, d_rgb, d_acc = buf[:8], buf[8:8 + 3 * R].view(R, 3), buf[8 + 3 * R:8 + 4 * R]
        d_accp, d_gth = buf[8 + 4 * R:8 + 4 * R + R * P].view(R, P), buf[8 + 4 * R + R * P:].view(N, 3)
        p = lambda t: None if t is None else t.data_ptr()
        args = MpLossArgs(p(rgb), p(rgb_gt), p(acc), p(accp), p(gth), p(sam), p(in_mask), p(d_rgb), p(d_acc), p(d_accp), p(d_gth),
                          p(terms), R, P, N, *w)
        hip.check(hip.lib().mp_loss_fused(C.byref(args), hip.stream()), "mp_loss_fused")
        ctx.grads = (d_rgb, d_acc, d_accp, d_gth)
        ctx.mark_non_differentiable(terms)
        return terms[0].reshape(1), terms

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        try:
            scaled = torch._foreach_mul(list(ctx.grads), g_total.reshape(()))   # one launch for the four gradients
        except (AttributeError, TypeError, RuntimeError):                        # (a torch without the list x tensor overload)
            scaled = [g * g_total.reshape(()) for g in ctx.grads]
        return tuple(g.reshape(sh) for g, sh in zip(scaled, ctx.shapes)) + (None, None, None, None)


def _get(opt, key, default):
    try:
        return opt.get(key, default)
    except AttributeError:
        return getattr(opt, key, default)


class Loss(nn.Module):
    def __init__(self, opt):
        super().__init__()
        for key in ("eikonal_weight", "bce_weight", "opacity_sparse_weight", "in_shape_weight", "sam_mask_weight",
                    "smpl_surface_milestone"):
            setattr(self, key, opt[key] if isinstance(opt, dict) else getattr(opt, key))
        self.smpl_surface_weight = _get(opt, "smpl_surface_weight", 0)
        self.zero_pose_weight = _get(opt, "zero_pose_weight", 0)
        self.sam_start_epoch = _get(opt, "sam_start_epoch", 200)
        self.increase_sam = _get(opt, "increase_sam", False)
        self.temporal_loss_weight = _get(opt, "temporal_loss_weight", 1.0)
        self.eps = 1e-6
        self.milestone = 200
        self.sam_milestone = 1000
        self.depth_loss_milestone = 1000

    # -- individual terms (names kept: the reference's trainer logs them one by one)
    # The reductions below never read a device value on the host (no boolean-mask indexing, no `if tensor:`): a training
    # iteration enqueues forward, loss and backward without waiting for the GPU.  Masked means are sum(mask * x) / sum(mask),
    # the same numbers as the reference's x[mask].mean() (0/0 = NaN where the reference's empty mean is NaN).
    def get_rgb_loss(self, rgb_values, rgb_gt, keep=None):
        d = (rgb_values - rgb_gt).abs()
        if keep is None:
            return d.mean()
        return torch.where(keep[:, None], d, torch.zeros_like(d)).sum() / (keep.sum() * d.shape[1])

    def get_eikonal_loss(self, grad_theta):
        return (grad_theta.norm(2, dim=-1) - 1).square().mean()

    def get_bce_los(self, acc_map):
        a = acc_map
        return -2.0 * (a * (a + self.eps).log() + (1 - a) * (1 - a + self.eps).log()).mean()

    @staticmethod
    def _masked_mean(x, mask):
        return torch.where(mask, x, torch.zeros_like(x)).sum() / mask.sum()

    def get_opacity_sparse(self, acc_map, index_off_surface):
        return self._masked_mean(acc_map.abs(), index_off_surface)

    def get_in_shape_loss(self, acc_map, index_in_surface):
        return self._masked_mean((acc_map - 1).abs(), index_in_surface)

    def get_sam_mask_loss(self, sam_mask, acc_person):
        prob = torch.sigmoid(sam_mask)
        ok = (prob.sum(dim=1) <= 1.01)[:, None].expand_as(prob)
        return self._masked_mean((acc_person - prob).abs(), ok)

    def get_sam_mask_clip_loss(self, sam_mask, acc_person):
        n_ray, n_person = sam_mask.shape[0], sam_mask.shape[1]
        prob = torch.sigmoid(sam_mask)
        ok = (prob.sum(dim=1) <= 1.01)[:, None].expand_as(prob)        # rays whose SAM masks do not overlap
        a, m = acc_person, prob
        agree = ((a < 0.04) & (m < 0.04)) | ((a > 0.96) & (m > 0.96))
        keep = ok & ~agree
        # loss.py:72-75: if nothing is kept the reference keeps the FIRST element of the selected rays ("clip_mask is all
        # False"): position of the first selected element, computed on the device
        first = torch.zeros_like(keep).reshape(-1)
        okf = ok.reshape(-1)
        idx = torch.argmax(okf.to(torch.int32))                        # first True (0 if none)
        first[idx] = True
        first = first.reshape(keep.shape) & ok
        keep = torch.where(keep.any(), keep, first)
        return torch.where(keep, (a - m).abs(), torch.zeros_like(a)).sum() / (n_ray * n_person)

    def get_depth_order_loss_samGT(self, t_list, mean_hitted_vertex_list, sam_mask, cam_loc):
        import numpy as np
        front = np.argmin(t_list, axis=0)
        correct = np.argmax(sam_mask.cpu().numpy(), axis=1)
        cols = torch.arange(mean_hitted_vertex_list.shape[1])
        d_front = (mean_hitted_vertex_list[front, cols, :] - cam_loc).norm(dim=-1)
        d_correct = (mean_hitted_vertex_list[correct, cols, :] - cam_loc).norm(dim=-1)
        return torch.log(1 + torch.exp(d_correct - d_front)).sum()

    def _forward_fused(self, mo, ground_truth):
        """Loss.forward with the ray / point reductions in csrc/loss.hip (same keys, same numbers to fp32 summation order)."""
        dev = mo["acc_map"].device
        epoch = mo["epoch"]
        e200 = min(self.milestone, epoch)
        use_sam = "sam_mask" in mo and epoch >= self.sam_start_epoch
        sam_ramp = min(1.0, epoch / 100) if self.increase_sam else 1.0
        in_mask = mo["index_in_surface"]
        w = (float(self.eikonal_weight), float(self.bce_weight), float(self.in_shape_weight * (1 - e200 / self.milestone)),
             float(self.sam_mask_weight * sam_ramp), float(self.eps))
        gth = mo["grad_theta"]
        accp = mo["acc_person_list"]
        total, terms = _FusedTerms.apply(mo["rgb_values"], mo["acc_map"], accp.reshape(mo["acc_map"].numel(), -1), gth,
                                         ground_truth["rgb"][0].to(dev), mo["sam_mask"] if use_sam else None, in_mask, w)
        zero = terms[6:7]                                                  # a device zero (no fill of its own)
        temporal_loss = mo["temporal_loss"]
        smpl_surface_loss = mo["smpl_surface_loss"] * self.smpl_surface_weight if self.smpl_surface_weight else zero
        zp_w = self.zero_pose_weight * (1 - min(1000, epoch) / 1000)
        loss = total + self.temporal_loss_weight * temporal_loss
        if self.smpl_surface_weight:
            loss = loss + smpl_surface_loss * (1 - min(self.smpl_surface_milestone, epoch) / self.smpl_surface_milestone)
        if zp_w:
            loss = loss + mo["zero_pose_loss"] * zp_w
        return {"loss": loss, "rgb_loss": terms[1], "depth_order_loss": zero, "eikonal_loss": terms[2], "bce_loss": terms[3:4],
                "opacity_sparse_loss": zero, "in_shape_loss": terms[4:5], "temporal_loss": temporal_loss,
                "sam_mask_loss": terms[5] if use_sam else zero, "smpl_surface_loss": smpl_surface_loss,
                "zero_pose_loss": mo["zero_pose_loss"]}

    def forward(self, model_outputs, ground_truth):
        mo = model_outputs
        dev = mo["acc_map"].device
        if (FUSED and dev.type == "cuda" and isinstance(mo["fg_rgb_values_each_person_list"], list)
                and all(torch.is_tensor(mo[k]) and mo[k].is_cuda for k in ("rgb_values", "acc_person_list", "grad_theta"))):
            return self._forward_fused(mo, ground_truth)
        zero = lambda: torch.zeros(1, device=dev)
        epoch = mo["epoch"]

        if isinstance(mo["fg_rgb_values_each_person_list"], list):        # loss.py:109-110 (always, for this model)
            depth_order_loss = zero()
        else:
            sam = mo["sam_mask"][mo["hitted_mask_idx"]]
            depth_order_loss = self.get_depth_order_loss_samGT(mo["t_list"], mo["mean_hitted_vertex_list"], sam,
                                                               mo["cam_loc"][mo["hitted_mask_idx"]])

        finite = ~torch.any(mo["rgb_values"].isnan(), dim=1)           # loss.py:120-122: rays with a NaN pixel are left out
        rgb_gt = ground_truth["rgb"][0].to(dev)
        rgb_loss = self.get_rgb_loss(torch.nan_to_num(mo["rgb_values"]), rgb_gt, keep=finite)
        eikonal_loss = self.get_eikonal_loss(mo["grad_theta"])
        # loss.py:124-128 / 131-139: a NaN term is REPLACED by a fresh zero, so nothing flows back through it.  Without a host
        # round trip that means: find the elements that would make the term NaN, evaluate it on sanitised values (no NaN
        # node in the graph -- zeroing only the result would still send 0 * NaN = NaN backwards into every weight), and
        # select the zero on the device.  (The reference's "Nan: bce_loss" print would need the value on the host.)
        acc = mo["acc_map"]
        bad_el = acc.isnan() | ((acc + self.eps) < 0) | ((1 - acc + self.eps) < 0)      # where the logarithms give NaN
        bce_loss = self.get_bce_los(torch.where(bad_el, torch.full_like(acc, 0.5), acc))
        bce_loss = torch.where(bad_el.any(), torch.zeros_like(bce_loss), bce_loss).reshape(1)
        opacity_sparse_loss = zero()
        if mo["index_in_surface"] is not None:
            mask = mo["index_in_surface"]
            bad = (mask.sum() == 0) | (acc.isnan() & mask).any()                          # empty mean or a NaN inside it
            in_shape_loss = torch.where(mask, (torch.nan_to_num(acc) - 1).abs(), torch.zeros_like(acc)).sum() / mask.sum().clamp(min=1)
            in_shape_loss = torch.where(bad, torch.zeros_like(in_shape_loss), in_shape_loss).reshape(1)
        else:
            in_shape_loss = zero()

        e200 = min(self.milestone, epoch)
        temporal_loss = mo["temporal_loss"]
        smpl_surface_loss = mo["smpl_surface_loss"] * self.smpl_surface_weight
        if "sam_mask" in mo and epoch >= self.sam_start_epoch:
            sam_mask_loss = self.get_sam_mask_clip_loss(mo["sam_mask"], mo["acc_person_list"])
        else:
            sam_mask_loss = zero()
        if epoch >= self.sam_start_epoch:
            depth_order_loss = depth_order_loss * (1 - min(self.depth_loss_milestone, epoch) / self.depth_loss_milestone)
        else:
            depth_order_loss = zero()
        zero_pose_loss = mo["zero_pose_loss"] * self.zero_pose_weight * (1 - min(1000, epoch) / 1000)
        sam_ramp = min(1.0, epoch / 100) if self.increase_sam else 1.0

        loss = (rgb_loss
                + self.eikonal_weight * eikonal_loss
                + self.bce_weight * bce_loss
                + self.opacity_sparse_weight * (1 + e200 ** 2 / 40) * opacity_sparse_loss
                + self.in_shape_weight * (1 - e200 / self.milestone) * in_shape_loss
                + self.temporal_loss_weight * temporal_loss
                + self.sam_mask_weight * sam_ramp * sam_mask_loss
                + smpl_surface_loss * (1 - min(self.smpl_surface_milestone, epoch) / self.smpl_surface_milestone)
                + depth_order_loss + zero_pose_loss)
        return {"loss": loss, "rgb_loss": rgb_loss, "depth_order_loss": depth_order_loss, "eikonal_loss": eikonal_loss,
                "bce_loss": bce_loss, "opacity_sparse_loss": opacity_sparse_loss, "in_shape_loss": in_shape_loss,
                "temporal_loss": temporal_loss, "sam_mask_loss": sam_mask_loss, "smpl_surface_loss": smpl_surface_loss,
                "zero_pose_loss": mo["zero_pose_loss"]}
