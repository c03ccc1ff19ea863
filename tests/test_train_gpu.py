"""GPU parity of the fp32 training building blocks (layer-wise forward with stash + hand-written backward) against torch
autograd on the oracle's formulas (same device, fp32).  Tolerances are fp32-roundoff class."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import multiply_oracle as O
from tests.util import seeded_networks

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _exact_fp32_gemms(monkeypatch):
    """These are tests of the ALGEBRA of the hand-written forward / adjoint kernels against torch autograd at fp32-roundoff
    tolerances, so the GEMMs between them run on the exact-fp32 matrix instruction (MP_TRAIN_PRECISION=f32); test_gemms checks
    both GEMM arithmetics themselves, tests/test_train_step_gpu.py the whole training step in the default split-bf16 mode."""
    from multiply_amd import train as T
    monkeypatch.setattr(T, "TRAIN_PRECISION", "f32")


def rel(name, got, want):
    got, want = got.double().cpu(), want.double().cpu()
    e = (got - want).abs().max().item() / (want.abs().max().item() + 1e-30)
    print(f"[grad parity] {name}: rel-to-max err {e:.3e} (|want|max {want.abs().max().item():.3e})")
    return e


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_gemms(precision, monkeypatch):
    """exact-fp32 MFMA (bitwise an fmaf chain: 2e-6 / 2e-5 relative to the largest entry) and the split-bf16 path (three 16-bit
    MFMAs per product, ~2^-16 per product: measured 4e-6 / 6e-6, asserted 3e-5) against a float64 product"""
    from multiply_amd import train as T
    monkeypatch.setattr(T, "TRAIN_PRECISION", precision)
    tol_nt, tol_tn = (2e-6, 2e-5) if precision == "f32" else (3e-5, 3e-5)
    torch.manual_seed(0)
    # the last two: one person's sample rows of a training iteration (ragged last row tile; a partly filled column block)
    for (M, N, K) in [(1000, 257, 39), (4097, 256, 256), (300, 3, 256), (129, 130, 17), (63001, 256, 256), (33000, 217, 256)]:
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        Cm = torch.empty(M, N, device="cuda")
        T.gemm_nt(T._p(A), K, T._p(B), K, T._p(Cm), N, M, N, K, T._p(b), M // 2)
        want = (A.double() @ B.double().T)
        want[:M // 2] += b.double()
        assert rel(f"gemm_nt[{precision}] {M}x{N}x{K}", Cm, want) < tol_nt
        T.gemm_nt(T._p(A), K, T._p(B), K, T._p(Cm), N, M, N, K, None, 0, accumulate=True, relu=True)
        assert rel("gemm_nt accumulate+relu", Cm, torch.relu(want + A.double() @ B.double().T)) < tol_nt
    for (M, N, K) in [(256, 256, 50000), (257, 39, 3000), (3, 128, 777), (256, 257, 100001)]:
        A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda")
        Cm = torch.ones(M, N, device="cuda")
        cs = torch.zeros(M, device="cuda")
        T.gemm_tn(T._p(A), M, T._p(B), N, T._p(Cm), N, M, N, K, T._p(cs), K // 2)
        assert rel(f"gemm_tn[{precision}] {M}x{N}x{K}", Cm, 1.0 + A.double().T @ B.double()) < tol_tn
        assert rel("gemm_tn fused column sum", cs, A[:K // 2].double().sum(0)) < 2e-5
    if precision == "bf16x3":      # the split keeps the RANGE of fp32: operands far below half precision's subnormals
        A = torch.randn(2000, 256, device="cuda") * 1e-9; B = torch.randn(256, 256, device="cuda")
        Cm = torch.empty(2000, 256, device="cuda")
        T.gemm_nt(T._p(A), 256, T._p(B), 256, T._p(Cm), 256, 2000, 256, 256)
        assert rel("gemm_nt[bf16x3] operands of magnitude 1e-9", Cm, A.double() @ B.double().T) < tol_nt


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


@pytest.mark.parametrize("P", [700, 129, 5])
def test_fused_sdf_kernels_against_autograd(P):
    """The layer-fused training kernels (csrc/tfuse.hip: value sweep + gradient sweep in one launch, the adjoint of both in a
    second one, split-bf16 products inside) against torch autograd with create_graph on the oracle's formula, and against the
    layer-wise path on exact-fp32 GEMMs: outputs, d sdf / dx, every parameter gradient, the conditioning adjoint.  P = 700: five
    full tiles and a ragged one; 129: a tile with a single point; 5: less than a wave.  The weights are perturbed away from the
    geometric initialisation so that hidden units sit in the softplus transition (sigma'' != 0)."""
    from multiply_amd import train as T
    m, _ = seeded_networks(1, 0)
    m = m.cuda()
    net = m.foreground_implicit_network_list[0]
    torch.manual_seed(11)
    with torch.no_grad():
        for prm in net.parameters():
            prm.add_(torch.randn_like(prm) * 0.02 * prm.abs().mean().clamp_min(1e-2))
    assert T.fused_sdf_supported(net)
    x = (torch.rand(P, 3, device="cuda") - 0.5) * 1.6
    cond = torch.randn(69, device="cuda") * 0.1
    fus = T.ImplicitTrainFused(net, x, cond)
    ref = T.ImplicitTrainRev(net, x, cond)
    sd = {k: v for k, v in m.named_parameters()}
    out, g = _implicit_torch(sd, "foreground_implicit_network_list.0.", x, cond, 6)
    assert rel("fused sdf+feat vs autograd", fus.out, out.detach()) < 3e-5
    assert rel("fused d sdf/dx vs autograd", fus.grad, g.detach()) < 1e-4
    assert rel("fused sdf+feat vs layer-wise", fus.out, ref.out) < 3e-5
    assert rel("fused d sdf/dx vs layer-wise", fus.grad, ref.grad) < 1e-4
    a_out = torch.randn(P, 257, device="cuda")
    a_g = torch.randn(P, 3, device="cuda")
    loss = (out * a_out).sum() + (g * a_g).sum()
    names = [n for n, p in m.named_parameters() if n.startswith("foreground_implicit_network_list.0.")]
    plist = [p for n, p in m.named_parameters() if n.startswith("foreground_implicit_network_list.0.")]
    want = torch.autograd.grad(loss, plist, allow_unused=True)
    dc_f = fus.backward(a_out[:, 1:].contiguous(), a_out[:, 0].contiguous(), a_g.clone())
    dc_r = ref.backward(a_out.clone(), a_g.clone())
    assert rel("d cond, fused vs layer-wise", dc_f, dc_r) < 2e-4
    got = dict(zip([id(p) for p in fus.params()], fus.param_grads()))
    gref = dict(zip([id(p) for p in ref.params()], ref.param_grads()))
    assert len(got) == len(want) == len(names)
    for n, p, ww in zip(names, plist, want):
        assert rel(n + " vs layer-wise", got[id(p)].reshape(ww.shape), gref[id(p)].reshape(ww.shape)) < 2e-4, n
        assert rel(n + " vs autograd", got[id(p)].reshape(ww.shape), ww) < 5e-4, n
    # a second forward on the same network object re-uses the persistent state (weights re-packed, accumulators zeroed)
    fus2 = T.ImplicitTrainFused(net, x, cond)
    assert torch.equal(fus2.out, fus.out) and torch.equal(fus2.grad, fus.grad)
    fus2.backward(a_out[:, 1:].contiguous(), a_out[:, 0].contiguous(), a_g.clone())
    for g1, g2 in zip(fus2.param_grads(), got.values()):
        assert rel("second iteration, same gradients", g1, g2) < 1e-5


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
    # per-sample data gradients: a flipped ReLU mask touches exactly the sample it belongs to -> all but a couple of rows
    # must agree tightly
    for nme, g, w in (("d XA", dXA, want[-2]), ("d feat", dZ8[:, 1:], want[-1])):
        row_err = (g - w).abs().amax(1) / w.abs().max()
        bad = int((row_err > 1e-5).sum())
        print(f"[grad parity] {nme}: rows off by > 1e-5 of max: {bad} of {n}; worst {float(row_err.max()):.3e}")
        assert bad <= 2, nme


@pytest.mark.parametrize("n", [900, 129])
def test_fused_colour_kernels_against_autograd(n):
    """The layer-fused colour-net kernels (csrc/tfuse.hip: mp_tf_col_fwd / mp_tf_col_bwd, split-bf16 products inside) against
    torch autograd on the oracle's formula: rgb, every parameter gradient, d XA and d feat (ReLU-mask flips: see above)."""
    from multiply_amd import train as T
    m, _ = seeded_networks(1, 0)
    m = m.cuda()
    ren = m.foreground_rendering_network_list[0]
    assert T.fused_col_supported(ren)
    torch.manual_seed(2)
    XA = torch.randn(n, 6, device="cuda")
    feat = (torch.randn(n + 7, 256, device="cuda") * 0.3).contiguous()      # more rows than points, like the SDF net's batch
    cond = torch.randn(69, device="cuda") * 0.1
    rt = T.RenderTrainFused(ren, XA, feat, n, cond)
    sd = {k: v for k, v in m.named_parameters()}
    XAg, featg = XA.clone().requires_grad_(True), feat[:n].clone().requires_grad_(True)
    want_rgb = O.rendering_forward_pose_no_view(sd, "foreground_rendering_network_list.0.", XAg[:, :3], XAg[:, 3:], cond, featg)
    assert rel("fused rgb", rt.rgb, want_rgb.detach()) < 3e-5
    a = torch.randn(n, 3, device="cuda")
    plist = [p for nme, p in m.named_parameters() if nme.startswith("foreground_rendering_network_list.0.")]
    names = [nme for nme, p in m.named_parameters() if nme.startswith("foreground_rendering_network_list.0.")]
    want = torch.autograd.grad((want_rgb * a).sum(), plist + [XAg, featg])
    dXA = torch.empty(n, 6, device="cuda")
    dfeat = torch.full((n + 7, 256), 7.0, device="cuda")
    rt.backward(a, dXA, dfeat)
    assert bool((dfeat[n:] == 7.0).all()), "rows past the n points are not touched"
    got = dict(zip([id(p) for p in rt.params()], rt.param_grads()))
    assert len(got) == len(plist)
    # per-sample data gradients first: a ReLU mask that flips (a pre-activation within the products' 2^-16 of zero -- these
    # inputs are dense around zero) touches exactly the sample it belongs to, so all but a few ROWS must agree tightly ...
    for nme, g, w in (("d XA", dXA, want[-2]), ("d feat", dfeat[:n], want[-1])):
        row_err = (g - w).abs().amax(1) / w.abs().max()
        bad = int((row_err > 5e-5).sum())
        print(f"[grad parity] fused {nme}: rows off by > 5e-5 of max: {bad} of {n}; worst {float(row_err.max()):.3e}")
        assert bad <= 4, nme
    # ... and one flipped sample moves an element of a 900-sample parameter gradient by up to ~2e-2 of the tensor's maximum
    for nme, p, ww in zip(names, plist, want[:len(plist)]):
        assert rel(nme, got[id(p)].reshape(ww.shape), ww) < 4e-2, nme


def test_background_branch_backward():
    """bg ImplicitNet (no weight norm, 4-D input, L=10) + bg RenderingNet (nerf_frame_encoding) + inverse-sphere
    compositing: forward and every parameter / frame-code gradient vs torch autograd on the oracle's formulas."""
    from multiply_amd import train as T
    m, _ = seeded_networks(1, 0)
    m = m.cuda()
    L = T.hip.lib()
    torch.manual_seed(3)
    R, NB = 96, 32
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda") * 0.2 + torch.tensor([0, 0, 1.0], device="cuda"), dim=1)
    cam = torch.tensor([0.05, -0.1, -2.5], device="cuda")
    code = torch.randn(32, device="cuda") * 0.3
    zb = (torch.linspace(0, 1, NB, device="cuda")[None] + torch.rand(R, NB, device="cuda") * 0.01).clamp(0, 1) / 3.0
    zbg = torch.flip(zb, dims=[-1]).contiguous()
    pts = torch.empty(R * NB, 4, device="cuda")
    T._chk(L.mp_tr_bg_points(T._p(dirs), T._p(cam), T._p(zbg), R, NB, C.c_float(3.0), T._p(pts), T.hip.stream()), "bg_points")
    want_pts = O.depth2pts_outside(cam[None, None].expand(R, NB, 3), dirs[:, None].expand(R, NB, 3), zbg, 3.0).reshape(-1, 4)
    assert rel("bg points", pts, want_pts) < 1e-5
    bit = T.ImplicitTrain(m.bg_implicit_network, pts, code, fwd=False)
    drep = dirs[:, None, :].expand(R, NB, 3).reshape(-1, 3).contiguous()
    XAb = torch.empty(R * NB, 27, device="cuda")
    T._chk(L.mp_tr_pe(T._p(drep), 3, R * NB, 4, 0, C.c_float(1.0), T._p(XAb), 27, 0, T.hip.stream()), "pe")
    brt = T.RenderTrain(m.bg_rendering_network, XAb, T.off(bit.out, 1), 257, R * NB, code)
    sdfb = bit.out[:, 0].contiguous()
    out = torch.empty(R, 3, device="cuda")
    T._chk(L.mp_tr_bg_comp_fwd(T._p(sdfb), T._p(brt.rgb), T._p(zbg), R, NB, T._p(out), T.hip.stream()), "bg_comp")
    # torch
    sd = {k: v for k, v in m.named_parameters()}
    codeg = code.clone().requires_grad_(True)
    o = O.implicit_forward(sd, "bg_implicit_network.", want_pts, codeg, multires=10)
    rgb = O.rendering_forward_nerf_frame(sd, "bg_rendering_network.", drep, o[:, 1:], codeg)
    dens = o[:, :1].abs().reshape(R, NB)
    dists = torch.cat([zbg[:, :-1] - zbg[:, 1:], 1e10 * torch.ones(R, 1, device="cuda")], -1)
    free = dists * dens
    shifted = torch.cat([torch.zeros(R, 1, device="cuda"), free[:, :-1]], -1)
    w = (1 - torch.exp(-free)) * torch.exp(-torch.cumsum(shifted, -1))
    want_out = (w[:, :, None] * rgb.reshape(R, NB, 3)).sum(1)
    assert rel("bg sdf+feat", bit.out, o.detach()) < 1e-5
    assert rel("bg rgb", out, want_out.detach()) < 1e-5
    a = torch.randn(R, 3, device="cuda")
    names = [n for n, p in m.named_parameters() if n.startswith("bg_")]
    plist = [p for n, p in m.named_parameters() if n.startswith("bg_")]
    want = torch.autograd.grad((want_out * a).sum(), plist + [codeg])
    rows = R * NB
    dsdfb = torch.empty(rows, device="cuda"); drgbb = torch.empty(rows, 3, device="cuda")
    T._chk(L.mp_tr_bg_comp_bwd(T._p(sdfb), T._p(brt.rgb), T._p(zbg), R, NB, T._p(a), T._p(dsdfb), T._p(drgbb), T.hip.stream()),
           "bg_comp_bwd")
    dZ8b = torch.zeros(rows, 257, device="cuda"); dXAb = torch.empty(rows, 27, device="cuda")
    dcode = brt.backward(drgbb, dXAb, T.off(dZ8b, 1), 257)
    dZ8b[:, 0] = dsdfb
    dcode = dcode + bit.backward(dZ8b)
    got = dict(zip([id(p) for p in bit.params() + brt.params()], bit.param_grads() + brt.param_grads()))
    errs = {n: rel(n, got[id(p)].reshape(ww.shape), ww) for n, p, ww in zip(names, plist, want)}
    errs["d frame code"] = rel("d frame code", dcode, want[-1])
    assert max(errs.values()) < 1e-2, errs


@pytest.mark.parametrize("P", [3000, 129, 5])
def test_fused_background_kernels_against_autograd(P, monkeypatch):
    """The layer-fused kernels of the background ImplicitNet (csrc/tfuse.hip k_tf_bg_fwd / k_tf_bg_bwd: the 84 Fourier features
    in the register operand, the skip connection patched into layer 4's operand, value sweep only; split-bf16 products) against
    torch autograd on the oracle's formula and against the layer-wise path: outputs, every parameter gradient, the frame code's
    adjoint.  Weights perturbed so that hidden units sit in the softplus transition."""
    from multiply_amd import train as T
    monkeypatch.setattr(T, "TRAIN_PRECISION", "bf16x3")       # the fused kernels' arithmetic (their weight gradients included)
    m, _ = seeded_networks(1, 0)
    m = m.cuda()
    net = m.bg_implicit_network
    torch.manual_seed(13)
    with torch.no_grad():
        for prm in net.parameters():
            prm.add_(torch.randn_like(prm) * 0.05 * prm.abs().mean().clamp_min(1e-2))
    assert T.fused_bg_supported(net)
    x = torch.cat([torch.nn.functional.normalize(torch.randn(P, 3, device="cuda"), dim=1), torch.rand(P, 1, device="cuda") / 3], 1).contiguous()
    code = torch.randn(32, device="cuda") * 0.3
    fus = T.ImplicitTrainFusedBG(net, x, code)
    ref = T.ImplicitTrain(net, x, code, fwd=False)
    sd = {k: v for k, v in m.named_parameters()}
    codeg = code.clone().requires_grad_(True)
    out = O.implicit_forward(sd, "bg_implicit_network.", x, codeg, multires=10)
    got_out = torch.cat([fus.sdf[:P, None], fus.feat[:P]], 1)
    assert rel("fused bg sdf+feat vs autograd", got_out, out.detach()) < 3e-5
    assert rel("fused bg sdf+feat vs layer-wise", got_out, ref.out) < 3e-5
    a_out = torch.randn(P, 257, device="cuda")
    names = [n for n, p in m.named_parameters() if n.startswith("bg_implicit_network.")]
    plist = [p for n, p in m.named_parameters() if n.startswith("bg_implicit_network.")]
    want = torch.autograd.grad((out * a_out).sum(), plist + [codeg])
    dc_f = fus.backward(a_out[:, 1:].contiguous(), a_out[:, 0].contiguous())
    dc_r = ref.backward(a_out.clone())
    assert rel("d frame code, fused vs layer-wise", dc_f, dc_r) < 2e-4
    assert rel("d frame code vs autograd", dc_f, want[-1]) < 5e-4
    got = dict(zip([id(p) for p in fus.params()], fus.param_grads()))
    gref = dict(zip([id(p) for p in ref.params()], ref.param_grads()))
    assert len(got) == len(names)
    for n, p, ww in zip(names, plist, want):
        assert rel(n + " vs layer-wise", got[id(p)].reshape(ww.shape), gref[id(p)].reshape(ww.shape)) < 2e-4, n
        assert rel(n + " vs autograd", got[id(p)].reshape(ww.shape), ww) < 5e-4, n


def test_composite_backward():
    """mp_composite / mp_tr_composite_bwd vs the oracle's packed compositing under autograd (two persons, rays hit by
    both, one or none; includes the exclusive background transmittance and d beta)."""
    from multiply_amd import train as T
    L = T.hip.lib()
    torch.manual_seed(4)
    R, NZ = 70, 98
    S = NZ - 1
    hit = [torch.arange(0, 50), torch.arange(30, 65)]
    z = [torch.sort(torch.rand(len(h), NZ) * 2.0 + 1.0, dim=1)[0] for h in hit]
    sdf = [torch.randn(len(h), S) * 0.05 for h in hit]
    rgb = [torch.rand(len(h), S, 3) for h in hit]
    nrm = [torch.randn(len(h), S, 3) for h in hit]
    bg = torch.rand(R, 3)
    beta_p = torch.tensor(0.02)
    cu = lambda t: t.cuda().contiguous()
    inv = []
    for h in hit:
        iv = torch.full((R,), -1, dtype=torch.int32)
        iv[h] = torch.arange(len(h), dtype=torch.int32)
        inv.append(cu(iv))
    zc, sc, rc, nc, bgc = [cu(t) for t in z], [cu(t.reshape(-1)) for t in sdf], [cu(t.reshape(-1, 3)) for t in rgb], \
        [cu(t.reshape(-1, 3)) for t in nrm], cu(bg)
    beta = cu(beta_p.reshape(1))
    tabs = [T._table(ts, "cuda") for ts in (inv, zc, sc, rc, nc)]
    f = lambda *s: torch.empty(*s, device="cuda")
    rgb_v, fg_v, nrm_v, acc, accp, bgT = f(R, 3), f(R, 3), f(R, 3), f(R), f(R, 2), f(R)
    T._chk(L.mp_composite(R, 2, NZ, *[T._p(t) for t in tabs], T._p(beta), T._p(bgc), T._p(rgb_v), T._p(fg_v), T._p(nrm_v),
                          T._p(acc), T._p(accp), T._p(bgT), T.hip.stream()), "composite")
    # oracle
    sdf_g = [t.clone().requires_grad_(True) for t in sdf]
    rgb_g = [t.clone().requires_grad_(True) for t in rgb]
    bg_g, beta_g = bg.clone().requires_grad_(True), beta_p.clone().requires_grad_(True)
    fg_o, nrm_o, acc_o, accp_o, bgT_o = O.packed_composite(R, hit, [t[:, :-1] for t in z], [t[:, -1] for t in z], sdf_g, rgb_g,
                                                           nrm, beta_g, [0, 1])
    rgb_o = fg_o + bgT_o[:, None] * bg_g
    assert rel("rgb_values", rgb_v, rgb_o.detach()) < 1e-5
    assert rel("acc", acc, acc_o.detach()) < 1e-5 and rel("acc_person", accp, accp_o.detach()) < 1e-5
    assert rel("bg_T", bgT, bgT_o.detach()) < 1e-5 and rel("normal_values", nrm_v, nrm_o.detach()) < 1e-5
    a_rgb, a_acc, a_accp = torch.randn(R, 3), torch.randn(R), torch.randn(R, 2)
    want = torch.autograd.grad((rgb_o * a_rgb).sum() + (acc_o * a_acc).sum() + (accp_o * a_accp).sum(),
                               sdf_g + rgb_g + [bg_g, beta_g])
    dsdf = [torch.zeros_like(t) for t in sc]; drgb = [torch.zeros_like(t) for t in rc]
    dbg, dbeta = torch.zeros(R, 3, device="cuda"), torch.zeros(1, device="cuda")
    td, tr = T._table(dsdf, "cuda"), T._table(drgb, "cuda")
    g_rgb, g_acc, g_accp = cu(a_rgb), cu(a_acc), cu(a_accp)      # keep the device copies alive across the launch
    T._chk(L.mp_tr_composite_bwd(R, 2, NZ, T._p(tabs[0]), T._p(tabs[1]), T._p(tabs[2]), T._p(tabs[3]), T._p(beta), T._p(bgc),
                                 T._p(g_rgb), T._p(g_acc), T._p(g_accp), T._p(td), T._p(tr), T._p(dbg), T._p(dbeta),
                                 T.hip.stream()), "composite_bwd")
    for p in range(2):
        assert rel(f"d sdf[{p}]", dsdf[p].reshape(-1, S), want[p]) < 1e-4
        assert rel(f"d rgb[{p}]", drgb[p].reshape(-1, S, 3), want[2 + p]) < 1e-5
    assert rel("d bg_rgb", dbg, want[4]) < 1e-5
    assert rel("d beta", dbeta.reshape(()), want[5]) < 1e-4
