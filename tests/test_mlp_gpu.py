"""GPU parity of the fused MLP kernels (through the C ABI) against the CPU oracle on the same seeded inputs.

Arithmetic: f16 MFMA operands (fp32 accumulation, half-precision activations); the reference is fp32 end to end, so the
tolerances are the stated half-precision tolerances of tests/tolerances.py: <= 5x the errors measured on MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import multiply_oracle as O
from tests import tolerances as TOL
from tests.util import seeded_networks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nets_gpu():
    m, opt = seeded_networks(2, 0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.cuda(), sd


def report(name, got, want):
    err = (got.double().cpu() - want.double()).abs()
    print(f"[parity] {name}: max {err.max().item():.3e} mean {err.mean().item():.3e} (ref max {want.abs().max().item():.3e})")
    return err.max().item()


def test_implicit_full_and_sdf(nets_gpu):
    from multiply_amd import hip
    m, sd = nets_gpu
    rng = np.random.RandomState(3)
    x = torch.tensor(rng.uniform(-0.9, 0.9, (3000, 3)), dtype=torch.float32)
    cond = torch.tensor(rng.normal(0, 0.1, 69), dtype=torch.float32)
    want = O.implicit_forward(sd, "foreground_implicit_network_list.1.", x, cond, multires=6)
    net = m.foreground_implicit_network_list[1]
    got = net(x.cuda(), {"smpl": cond.cuda()[None]})[0]
    torch.cuda.synchronize()
    assert report("fg implicit sdf", got[:, 0], want[:, 0]) < TOL.MLP["fg_sdf"]
    assert report("fg implicit feat", got[:, 1:], want[:, 1:]) < TOL.MLP["fg_feat"]
    sdf = hip.implicit_sdf(net, x.cuda(), cond.cuda())
    assert report("fg sdf-only kernel vs full kernel", sdf, got[:, 0].cpu()) < 1e-6
    # background network: 4-D input, 10 octaves, frame conditioning
    x4 = torch.tensor(rng.uniform(-1, 1, (1500, 4)), dtype=torch.float32)
    code = sd["frame_latent_encoder.weight"][7]
    want = O.implicit_forward(sd, "bg_implicit_network.", x4, code, multires=10)
    got = m.bg_implicit_network(x4.cuda(), {"frame": code.cuda()[None]})[0]
    assert report("bg implicit sdf", got[:, 0], want[:, 0]) < TOL.MLP["bg_sdf"]
    assert report("bg implicit feat", got[:, 1:], want[:, 1:]) < TOL.MLP["bg_feat"]


def test_precise_sdf_value_kernel(nets_gpu):
    """mp_tf_sdf_val (csrc/tfuse.hip): the sampler's queries at the training path's arithmetic -- split-bfloat16 products, fp32
    activations -- through mp_mlp_sdf's worklist interface: against the fp32 oracle (two orders of magnitude below the f16 kernel's
    1e-3) and, for the worklist / device-count handling, untouched entries stay untouched"""
    from multiply_amd import hip, train as T
    m, sd = nets_gpu
    rng = np.random.RandomState(11)
    n = 5000                                                     # not a multiple of the 128-point tile
    x = torch.tensor(rng.uniform(-0.9, 0.9, (n, 3)), dtype=torch.float32)
    cond = torch.tensor(rng.normal(0, 0.1, 69), dtype=torch.float32)
    want = O.implicit_forward(sd, "foreground_implicit_network_list.1.", x, cond, multires=6)[:, 0]
    net = m.foreground_implicit_network_list[1]
    fs = T.fused_sdf_state(net).refresh(cond.cuda())
    L = hip.lib()
    xd = x.cuda()
    out = torch.full((n,), -7.0, device="cuda")
    hip.check(L.mp_tf_sdf_val(hip.ptr(fs.wpack), hip.ptr(fs.bias_all), hip.ptr(xd), None, None, n, hip.ptr(out), hip.stream()), "val")
    torch.cuda.synchronize()
    assert report("precise sdf kernel (all points)", out, want) < 2e-5
    f16 = hip.implicit_sdf(net, xd, cond.cuda())
    assert report("f16 sdf kernel, same points", f16, want) > 10 * float((out.cpu() - want).abs().max())
    # worklist of every third point in shuffled order, device-side count smaller than the launch's upper bound
    ids = torch.tensor(rng.permutation(np.arange(0, n, 3)), dtype=torch.int32).cuda()
    k = 1000
    work = torch.cat([ids[:k], torch.zeros(300, dtype=torch.int32, device="cuda")])
    cnt = torch.tensor([k], dtype=torch.int32, device="cuda")
    out2 = torch.full((n,), -7.0, device="cuda")
    hip.check(L.mp_tf_sdf_val(hip.ptr(fs.wpack), hip.ptr(fs.bias_all), hip.ptr(xd), hip.ptr(work), hip.ptr(cnt), k + 300, hip.ptr(out2),
                              hip.stream()), "val")
    torch.cuda.synchronize()
    sel = ids[:k].long()
    assert torch.equal(out2[sel], out[sel])
    rest = torch.ones(n, dtype=torch.bool, device="cuda")
    rest[sel] = False
    assert bool((out2[rest] == -7.0).all())


def test_split_activation_sdf_kernel(nets_gpu):
    """mp_mlp_sdf_x2 (csrc/mlp.hip k_mlp_sdf_x2, round 6): the sampler's default queries -- the half-precision kernel's packed weights,
    activations as two halves, fp32 softplus -- against the fp32 oracle: an order of magnitude below the f16 kernel's error on the
    same points; worklist / device-count handling like mp_mlp_sdf (untouched entries stay untouched)"""
    from multiply_amd import hip
    m, sd = nets_gpu
    rng = np.random.RandomState(12)
    n = 5003                                                     # not a multiple of the 128-point tile
    x = torch.tensor(rng.uniform(-0.9, 0.9, (n, 3)), dtype=torch.float32)
    cond = torch.tensor(rng.normal(0, 0.1, 69), dtype=torch.float32)
    want = O.implicit_forward(sd, "foreground_implicit_network_list.1.", x, cond, multires=6)[:, 0]
    net = m.foreground_implicit_network_list[1]
    xd = x.cuda()
    out = hip.implicit_sdf(net, xd, cond.cuda(), mode="f16x2")
    f16 = hip.implicit_sdf(net, xd, cond.cuda())
    torch.cuda.synchronize()
    e2, e1 = report("split-activation sdf kernel", out, want), report("f16 sdf kernel, same points", f16, want)
    assert e2 < TOL.MLP["fg_sdf_x2"] and e1 > 2.5 * e2
    L, pk = hip.lib(), hip.packed(net, "sdf", 2)
    ids = torch.tensor(rng.permutation(np.arange(0, n, 3)), dtype=torch.int32).cuda()
    k = 1000
    work = torch.cat([ids[:k], torch.zeros(300, dtype=torch.int32, device="cuda")])
    cnt = torch.tensor([k], dtype=torch.int32, device="cuda")
    out2 = torch.full((n,), -7.0, device="cuda")
    hip.check(L.mp_mlp_sdf_x2(C.byref(pk.net), hip.ptr(pk.wpack), hip.ptr(pk.bias), hip.ptr(xd), hip.ptr(work), hip.ptr(cnt), k + 300,
                              hip.ptr(out2), hip.stream()), "mp_mlp_sdf_x2")
    torch.cuda.synchronize()
    sel = ids[:k].long()
    assert torch.equal(out2[sel], out[sel])
    rest = torch.ones(n, dtype=torch.bool, device="cuda")
    rest[sel] = False
    assert bool((out2[rest] == -7.0).all())


def test_rendering_net_standalone_forward(nets_gpu):
    """RenderingNet.forward (networks.py:263-312, mode pose_no_view) called like the reference does -- points, normals, view
    dirs, body pose, fp32 feature vectors -- through the fragment packer + mp_mlp_color"""
    m, sd = nets_gpu
    rng = np.random.RandomState(6)
    n = 777                                                      # not a multiple of any tile size
    x = torch.tensor(rng.uniform(-0.9, 0.9, (n, 3)), dtype=torch.float32)
    nrm = torch.nn.functional.normalize(torch.tensor(rng.normal(0, 1, (n, 3)), dtype=torch.float32), dim=1)
    cond = torch.tensor(rng.normal(0, 0.1, 69), dtype=torch.float32)
    feat = torch.tensor(rng.normal(0, 0.5, (n, 256)), dtype=torch.float32)
    want = O.rendering_forward_pose_no_view(sd, "foreground_rendering_network_list.1.", x, nrm, cond, feat)
    net = m.foreground_rendering_network_list[1]
    got = net(x.cuda(), nrm.cuda(), None, cond.cuda()[None], feat.cuda())
    assert got.shape == (n, 3)
    assert report("standalone RenderingNet rgb", got, want[:, :3]) < 3e-5      # f16-rounded external features: measured 4.9e-6
    with pytest.raises(NotImplementedError):
        m.bg_rendering_network(None, None, x.cuda(), None, feat.cuda(), torch.zeros(1, 32).cuda())


def test_shade_points(nets_gpu):
    from multiply_amd import hip
    m, sd = nets_gpu
    rng = np.random.RandomState(4)
    n = 2000
    x = torch.tensor(rng.uniform(-0.9, 0.9, (n, 3)), dtype=torch.float32)
    cond = torch.tensor(rng.normal(0, 0.1, 69), dtype=torch.float32)
    A = torch.tensor(rng.normal(0, 0.3, (n, 3, 3)), dtype=torch.float32) + torch.eye(3)
    jinv = torch.linalg.inv(A)
    xg = x.clone().requires_grad_(True)
    out = O.implicit_forward(sd, "foreground_implicit_network_list.0.", xg, cond, multires=6)
    g = torch.autograd.grad(out[:, :1], xg, torch.ones(n, 1))[0]
    nrm = torch.nn.functional.normalize(torch.einsum("bi,bij->bj", g, jinv), dim=1)
    rgb = O.rendering_forward_pose_no_view(sd, "foreground_rendering_network_list.0.", x, nrm, cond, out[:, 1:].detach())
    # both implementations of the value + input-gradient pass: reverse mode (default: forward sweep + transposed reverse
    # sweep) and forward mode (value + three tangent columns); tolerances = 5-10x the errors measured with f16 operands
    res = {}
    for mode in ("reverse", "forward"):
        sdf_g, nrm_g, rgb_g = hip.shade_points(m.foreground_implicit_network_list[0], m.foreground_rendering_network_list[0],
                                               x.cuda(), jinv.cuda(), cond.cuda(), mode=mode)
        torch.cuda.synchronize()
        assert report(f"shade sdf ({mode})", sdf_g, out[:, 0].detach()) < TOL.MLP["shade_sdf"]
        assert report(f"shade normal ({mode})", nrm_g, nrm) < TOL.MLP["shade_normal"]
        assert report(f"shade rgb ({mode})", rgb_g, rgb.detach()) < TOL.MLP["shade_rgb"]
        res[mode] = (sdf_g, nrm_g, rgb_g)
    assert torch.equal(res["reverse"][0], res["forward"][0])          # the value column is the same arithmetic
    assert report("normals reverse vs forward", res["reverse"][1], res["forward"][1].cpu()) < TOL.MLP["normal_rev_vs_fwd"]


def test_background(nets_gpu, smpl_tables):
    from multiply_amd import hip
    m, sd = nets_gpu
    model = O.MultiplyOracle(sd, smpl_tables, np.zeros((2, 10), np.float32))
    rng = np.random.RandomState(5)
    R = 300
    d = torch.nn.functional.normalize(torch.tensor(rng.normal(0, 1, (R, 3)), dtype=torch.float32) + torch.tensor([0, 0, 2.0]), dim=1)
    cam = torch.tensor([0.1, -0.2, -2.5])
    code = sd["frame_latent_encoder.weight"][11]
    want = model.background(d, cam[None].expand(R, -1), code)
    z = torch.flip(O.bg_depths(model.cfg, 1), dims=[-1])[0]
    got = hip.background(m.bg_implicit_network, m.bg_rendering_network, d.cuda(), cam.cuda(), z.cuda(), code.cuda())
    torch.cuda.synchronize()
    assert report("background rgb", got, want) < TOL.MLP["bg_rgb"]
