"""The torch arithmetic of the mesh-space losses (multiply_amd/mesh_losses.py: front depth, instance masks, depth-order loss,
the silhouette target, differentiable skinning, body-model rows) on the CPU against the numpy restatement
(oracle/raster_oracle.py) and plain formulas; the device parts (z-buffer, inside test) are covered by tests/test_raster_gpu.py."""
import numpy as np
import torch

from multiply_amd import mesh_losses as ML
from oracle import raster_oracle as RO


def random_maps(rs, P=3, H=17, W=23):
    depth = []
    for p in range(P):
        d = rs.uniform(2.0, 5.0, (H, W))
        d[rs.uniform(size=(H, W)) < 0.4] = -1.0                     # holes
        depth.append(d)
    return depth, rs.normal(0, 3.0, (H, W, P))


def test_front_depth_masks_and_loss_match_the_restatement():
    rs = np.random.RandomState(0)
    for trial in range(5):
        depth, sam = random_maps(rs, P=2 + trial % 3)
        td = [torch.tensor(d) for d in depth]
        mx, front = ML.front_depth(td)
        wmx, wfront, wmasks = RO.front_depth_and_masks(depth)
        assert np.array_equal(mx.numpy(), wmx) and np.array_equal(front.numpy(), wfront)
        assert np.array_equal(ML.instance_masks(td).numpy(), wmasks)
        for epoch, wgt in ((0, 0.005), (300, 0.1), (1000, 0.1), (2000, 0.1)):
            got = float(ML.depth_order_loss(td, torch.tensor(sam)[None], epoch, wgt))
            want = RO.depth_order_loss(depth, sam, epoch, wgt)
            assert abs(got - want) <= 1e-12 * max(1.0, abs(want)), (trial, epoch)
    # nothing out of order -> exactly zero, with a graph attached (the reference returns a fresh zero there)
    d = [torch.full((4, 5), 2.0, dtype=torch.float64, requires_grad=True), torch.full((4, 5), 3.0, dtype=torch.float64)]
    sam = torch.zeros(1, 4, 5, 2); sam[..., 0], sam[..., 1] = 9.0, -9.0
    z = ML.depth_order_loss(d, sam, 10, 0.1)
    assert float(z.detach()) == 0.0 and z.requires_grad


def test_depth_order_loss_gradient_is_the_logistic_of_the_depth_gap():
    rs = np.random.RandomState(1)
    depth, sam = random_maps(rs, P=2, H=9, W=11)
    td = [torch.tensor(d, requires_grad=True) for d in depth]
    loss = ML.depth_order_loss(td, torch.tensor(sam), 250, 0.1)
    loss.backward()
    mx, front, _ = RO.front_depth_and_masks(depth)
    s = 1 / (1 + np.exp(-sam))
    valid = (front < 999) & (s.sum(-1) <= 1.01) & (s.sum(-1) >= 0.7)
    lab = s.argmax(-1)
    gt = np.take_along_axis(mx, lab[..., None], -1)[..., 0]
    use = valid & (gt < 999) & (gt != front)
    scale = 0.1 * (1 - 250 / 1000)
    sig = 1 / (1 + np.exp(-(gt - front)))
    for p in range(2):
        want = np.zeros_like(depth[p])
        want += np.where(use & (lab == p), scale * sig, 0.0)                       # the labelled (hidden) surface is pulled forward
        want -= np.where(use & (mx[..., p] == front) & (lab != p), scale * sig, 0.0)   # the wrongly-front one pushed back
        assert np.allclose(td[p].grad.numpy(), want, atol=1e-12)


def test_silhouette_target_and_skinning_and_body_rows():
    sam = torch.tensor([[[[5.0, -5.0], [-5.0, 5.0], [-5.0, -5.0]]]])               # person 0, person 1, background
    cm = ML.gt_instance_map(sam, 2)
    assert cm.shape == (1, 3, 3) and cm[0].tolist() == [[255.0, 0.0, 0.0], [0.0, 255.0, 0.0], [0.0, 0.0, 0.0]]
    # skinning: x' = sum_j w_j T_j [x; 1], and its inverse
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(1, 7, 3, generator=g), torch.softmax(torch.randn(1, 7, 24, generator=g), -1)
    tfs = torch.eye(4).repeat(1, 24, 1, 1) + 0.05 * torch.randn(1, 24, 4, 4, generator=g)
    tfs[:, :, 3] = torch.tensor([0.0, 0, 0, 1])
    y = ML.skinning(x, w, tfs)
    T = (w[0, :, :, None, None] * tfs[0][None]).sum(1)
    want = (T @ torch.cat([x[0], torch.ones(7, 1)], 1)[:, :, None])[:, :3, 0]
    assert torch.allclose(y[0], want, atol=1e-6) and torch.allclose(ML.skinning(y, w, tfs, inverse=True), x, atol=1e-5)
    from multiply_amd.body_model_params import BodyModelParams
    bml = []
    for p in range(2):
        bm = BodyModelParams(3)
        bm.init_parameters("transl", torch.arange(9.0).reshape(3, 3) + 10 * p)
        bm.init_parameters("body_pose", torch.ones(3, 69) * (p + 1))
        bml.append(bm)
    trans, shape, pose = ML.body_model_inputs(bml, torch.tensor([2]))
    assert trans.shape == (1, 2, 3) and shape.shape == (1, 2, 10) and pose.shape == (1, 2, 72)
    assert trans[0, 1].tolist() == [16.0, 17.0, 18.0] and float(pose[0, 1, 3:].mean()) == 2.0 and float(pose[0, :, :3].abs().sum()) == 0.0
