"""In / off-surface flags (multiply.py:153-167): mp_mesh_signed_distance / mp_mesh_ray_flags vs the float64 oracle."""
import numpy as np
import pytest
import torch

from oracle import multiply_oracle as O

pytestmark = pytest.mark.gpu


def icosphere(n=3, radius=0.5):
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float)
    f = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])
    for _ in range(n):
        cache, vs, nf = {}, list(v), []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = (vs[a] + vs[b]) / 2
                cache[k] = len(vs)
                vs.append(m / np.linalg.norm(m))
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(vs), np.array(nf)
    return torch.tensor(v * radius, dtype=torch.float32), torch.tensor(f, dtype=torch.int64)


def test_signed_distance_and_flags_match_oracle():
    from multiply_amd import hip
    L = hip.lib()
    v, f = icosphere(3)
    # squash the sphere so that it is not symmetric and add an offset
    v = v * torch.tensor([1.0, 1.4, 0.7]) + torch.tensor([0.03, -0.02, 0.05])
    fv = v[f].contiguous()                                       # (F,3,3) like mesh_face_vertices_list[p][0]
    g = torch.Generator().manual_seed(0)
    n_rays, n_s = 300, 17
    pts = (torch.rand(n_rays * n_s, 3, generator=g) - 0.5) * 2.0
    pts[:n_s * 40] *= 0.2                                        # some rays entirely inside
    pts[n_s * 40:n_s * 80] = pts[n_s * 40:n_s * 80] * 0.1 + torch.tensor([0.9, 0.9, 0.9])   # some entirely far outside
    want_off, want_in, want_sd = O.off_in_surface_flags(pts, n_s, fv, 0.05)
    dp, dfv = pts.cuda(), fv.reshape(-1, 9).cuda()
    sd = torch.empty(pts.shape[0], device="cuda")
    off = torch.empty(n_rays, dtype=torch.uint8, device="cuda"); inn = torch.empty(n_rays, dtype=torch.uint8, device="cuda")
    hip.check(L.mp_mesh_signed_distance(hip.ptr(dp), pts.shape[0], hip.ptr(dfv), fv.shape[0], hip.ptr(sd), hip.stream()), "sd")
    hip.check(L.mp_mesh_ray_flags(hip.ptr(sd), n_rays, n_s, 0.05, hip.ptr(off), hip.ptr(inn), hip.stream()), "flags")
    torch.cuda.synchronize()
    err = (sd.cpu().double() - want_sd.reshape(-1)).abs()
    print(f"[parity] mesh signed distance: max {err.max().item():.3e}; inside fraction {(want_sd < 0).double().mean().item():.3f}")
    assert err.max() < 1e-5                                      # fp32 vs float64 geometry; also proves every sign agrees
    # flags are exact except for rays whose minimum sits within fp32 round-off of a threshold
    m = want_sd.min(1)[0]
    decided = ((m - 0.05).abs() > 1e-5) & (m.abs() > 1e-5)
    assert decided.sum() > 0.95 * n_rays
    assert torch.equal(off.cpu().bool()[decided], want_off[decided]) and torch.equal(inn.cpu().bool()[decided], want_in[decided])
    assert want_off.sum() > 10 and want_in.sum() > 10 and (~want_off & ~want_in).sum() > 10


def test_training_forward_epoch_below_250_returns_flags():
    from tests.test_train_step_gpu import _train_setup
    model, oracle, inp, gin, gt, loss_fn, train = _train_setup(epoch=101)
    R = inp["uv"].shape[1]
    # the trainer replaces the canonical meshes every 20 epochs (multiply_model.py:504-506): install a closed one
    v, f = icosphere(3, radius=0.45)
    for p in range(2):
        model.mesh_v_cano_list[p] = v[None].cuda()
        model.mesh_f_cano_list[p] = f.cuda()
        model.mesh_face_vertices_list[p] = v[f][None].cuda()
    hit = [torch.arange(R), torch.arange(R)]
    out = model({**gin, "hit_index": hit})
    graph = model._last_train
    assert out["index_off_surface"].dtype == torch.bool and out["index_off_surface"].shape == (R,)
    off_l, in_l = [], []
    for p in range(2):
        x_c = graph.fg[p]["X"][:graph.fg[p]["npts"]].cpu()
        o, i, sd = O.off_in_surface_flags(x_c, 97, v[f], 0.05)
        m = sd.min(1)[0]
        assert (((m - 0.05).abs() > 1e-5) & (m.abs() > 1e-5)).all(), "test scene has a ray on a threshold"
        off_l.append(o); in_l.append(i)
    want_off = torch.stack(off_l, 1).all(1)
    want_in = torch.stack(in_l, 1).any(1)
    assert torch.equal(out["index_off_surface"].cpu(), want_off) and torch.equal(out["index_in_surface"].cpu(), want_in)
    print("[info] off-surface rays", int(want_off.sum()), "in-surface rays", int(want_in.sum()), "of", R)
    lo = loss_fn(out, gt)
    assert float(lo["in_shape_loss"]) > 0 and torch.isfinite(lo["loss"])
    lo["loss"].backward()
