"""GPU parity of the fp32 training building blocks (layer-wise forward with stash + hand-written backward) against torch
autograd on the oracle's formulas (same device, fp32).  Tolerances are fp32-roundoff class."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import multiply_oracle as O
from tests.util import seeded_networks

pytestmark = pytest.mark.gpu


def rel(name, got, want):
    got, want = got.double().cpu(), want.double().cpu()
    e = (got - want).abs().max().item() / (want.abs().max().item() + 1e-30)
    print(f"[grad parity] {name}: rel-to-max err {e:.3e} (|want|max {want.abs().max().item():.3e})")
    return e


def test_gemms():
    from multiply_amd import train as T
    torch.manual_seed(0)
    for (M, N, K) in [(1000, 257, 39), (4097, 256, 256), (300, 3, 256), (129, 130, 17)]:
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        Cm = torch.empty(M, N, device="cuda")
        T.gemm_nt(T._p(A), K, T._p(B), K, T._p(Cm), N, M, N, K, T._p(b), M // 2)
        want = A @ B.T
        want[:M // 2] += b
        assert rel(f"gemm_nt {M}x{N}x{K}", Cm, want) < 2e-6
        T.gemm_nt(T._p(A), K, T._p(B), K, T._p(Cm), N, M, N, K, None, 0, accumulate=True, relu=True)
        assert rel("gemm_nt accumulate+relu", Cm, torch.relu(want + A @ B.T)) < 2e-6
    for (M, N, K) in [(256, 256, 50000), (257, 39, 3000), (3, 128, 777)]:
        A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda")
        Cm = torch.ones(M, N, device="cuda")
        T.gemm_tn(T._p(A), M, T._p(B), N, T._p(Cm), N, M, N, K)
        assert rel(f"gemm_tn {M}x{N}x{K}", Cm, 1.0 + A.T @ B) < 2e-5


def _implicit_torch(sd, prefix, x, cond, multires):
    """sdf, feat and d sdf/dx with a differentiable graph (create_graph) on the oracle's formula"""
    xg = x.clone().requires_grad_(True)
    out = O.implicit_forward(sd, prefix, xg, cond, multires)
    g = torch.autograd.grad(out[:, 0].sum(), xg, create_graph=True)[0]
    return out, g


def test_implicit_forward_mode_and_second_order_backward():
    from multiply_amd import train as T
    m, _ = seeded_networks(1, 0)
    m = m.cuda()
    net = m.foreground_implicit_network_list[0]
    torch.manual_seed(1)
    P = 700
    x = (torch.rand(P, 3, device="cuda") - 0.5) * 1.6
    cond = torch.randn(69, device="cuda") * 0.1
    it = T.ImplicitTrain(net, x, cond, fwd=True)
    params = {k: v for k, v in m.named_parameters()}
    sd = {k: v for k, v in params.items()}
    out, g = _implicit_torch(sd, "foreground_implicit_network_list.0.", x, cond, 6)
    Z8 = it.out
    assert rel("sdf+feat", Z8[:P], out.detach()) < 1e-5
    for k in range(3):
        assert rel(f"d sdf/dx{k}", Z8[(k + 1) * P:(k + 2) * P, 0], g[:, k].detach()) < 2e-5
    # random adjoints on sdf, features and the spatial gradient (the normal / eikonal paths)
    a_out = torch.randn(P, 257, device="cuda")
    a_g = torch.randn(P, 3, device="cuda")
    loss = (out * a_out).sum() + (g * a_g).sum()
    plist = [p for n, p in m.named_parameters() if n.startswith("foreground_implicit_network_list.0.")]
    want = torch.autograd.grad(loss, plist, allow_unused=True)
    dZ8 = torch.zeros(4 * P, 257, device="cuda")
    dZ8[:P] = a_out
    for k in range(3):
        dZ8[(k + 1) * P:(k + 2) * P, 0] = a_g[:, k]
    it.backward(dZ8)
    got = dict(zip([id(p) for p in it.params()], it.param_grads()))
    names = [n for n, p in m.named_parameters() if n.startswith("foreground_implicit_network_list.0.")]
    assert len(got) == len(want) == len(names)
    for n, p, ww in zip(names, plist, want):
        assert rel(n, got[id(p)].reshape(ww.shape), ww) < 5e-4, n


def test_rendering_net_backward():
    from multiply_amd import train as T
    m, _ = seeded_networks(1, 0)
    m = m.cuda()
    ren = m.foreground_rendering_network_list[0]
    torch.manual_seed(2)
    n = 900
    XA = torch.randn(n, 6, device="cuda")
    Z8 = torch.randn(n, 257, device="cuda") * 0.3
    cond = torch.randn(69, device="cuda") * 0.1
    rt = T.RenderTrain(ren, XA, T.off(Z8, 1), 257, n, cond)
    sd = {k: v for k, v in m.named_parameters()}
    XAg, featg = XA.clone().requires_grad_(True), Z8[:, 1:].clone().requires_grad_(True)
    want_rgb = O.rendering_forward_pose_no_view(sd, "foreground_rendering_network_list.0.", XAg[:, :3], XAg[:, 3:], cond, featg)
    assert rel("rgb", rt.rgb, want_rgb.detach()) < 1e-5
    a = torch.randn(n, 3, device="cuda")
    plist = [p for nme, p in m.named_parameters() if nme.startswith("foreground_rendering_network_list.0.")]
    names = [nme for nme, p in m.named_parameters() if nme.startswith("foreground_rendering_network_list.0.")]
    want = torch.autograd.grad((want_rgb * a).sum(), plist + [XAg, featg])
    dXA = torch.empty(n, 6, device="cuda")
    dZ8 = torch.zeros(n, 257, device="cuda")
    rt.backward(a, dXA, T.off(dZ8, 1), 257)
    got = dict(zip([id(p) for p in rt.params()], rt.param_grads()))
    for nme, p, ww in zip(names, plist, want[:len(plist)]):
        # ReLU masks of pre-activations within fp32 round-off of 0 can differ between two summation orders: one flipped
        # sample moves an element of a 900-sample gradient by ~1e-3 of the maximum
        assert rel(nme, got[id(p)].reshape(ww.shape), ww) < 1e-2, nme
    assert rel("d XA", dXA, want[-2]) < 1e-2
    assert rel("d feat", dZ8[:, 1:], want[-1]) < 1e-2
