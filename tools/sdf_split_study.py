"""Which operand roundings of the sampler's SDF queries matter?  (round-6 study, CPU only; no GPU needed)

Emulates, in float64 arithmetic on rounded operands, the fused SDF-value kernels' MFMA arithmetic layer by layer and prints the
error of the sdf against the float64 network on canonical points within the 0.1 outlier radius of the synthetic body:
    f16      weights and activations rounded to IEEE half (k_mlp_sdf), activation function exact
    b3       split-bfloat16 x3  (hi.hi + hi.lo + lo.hi; k_tf_sdf_val)
    h3       split-half x3      (hi.hi + hi.lo + lo.hi on IEEE halves)
    w2       weights exact to two halves, activations ONE half (two MFMAs)
    x2       weights ONE half, activations two halves
per-layer mixes: a string of 9 codes, one per linear layer.
    python tools/sdf_split_study.py [n_points]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from oracle import multiply_oracle as O         # noqa: E402


def r_half(a):
    return a.to(torch.float32).to(torch.float16).to(torch.float64)


def r_bf16(a):
    return a.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def split(a, rnd):
    hi = rnd(a)
    lo = rnd(a - hi)
    return hi, lo


def product(w, x, code):
    """x (N, K) @ w (O, K)^T with the operand rounding of `code`; accumulation exact (fp32 accumulation error is 1e-7 class)"""
    if code == "e":
        return x @ w.T
    if code == "f16":
        return r_half(x) @ r_half(w).T
    if code in ("b3", "h3"):
        rnd = r_bf16 if code == "b3" else r_half
        wh, wl = split(w, rnd)
        xh, xl = split(x, rnd)
        return xh @ wh.T + xl @ wh.T + xh @ wl.T
    if code == "w2":
        wh, wl = split(w, r_half)
        xh = r_half(x)
        return xh @ wh.T + xh @ wl.T
    if code == "x2":
        wh = r_half(w)
        xh, xl = split(x, r_half)
        return xh @ wh.T + xl @ wh.T
    raise ValueError(code)


def net(sd, prefix, x, cond, codes, act32=True):
    emb = O.fourier_embed(x.float(), 6).double() if act32 else O.fourier_embed(x, 6)
    h = emb
    for l in range(9):
        w, b = O.linear_params(sd, prefix, l)
        w, b = w.double(), b.double()
        if l == 0:
            h = torch.cat([h, cond.double().view(1, -1).expand(h.shape[0], -1)], -1)
        if l == 4:
            h = torch.cat([h, emb], 1) / np.sqrt(2)
        h = product(w, h, codes[l]) + b
        if l < 8:
            h = torch.nn.functional.softplus(h, beta=100.0, threshold=20.0)
            if act32:
                h = h.float().double()          # the activation leaves the VALU as an fp32 number
    return h[:, 0]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    torch.manual_seed(0)
    from multiply_amd.config import load_config
    from multiply_amd.networks import ImplicitNet
    from multiply_amd.synthetic import make_scene, make_smpl_tables
    tables, sc = make_smpl_tables(0), make_scene(2, seed=0, H=512, W=512)
    netm = ImplicitNet(load_config().implicit_network)       # the shipped geometric initialisation (what bench.py renders)
    prefix = "n."
    sd = {prefix + k: v.detach() for k, v in netm.state_dict().items()}
    betas = torch.tensor(sc["smpl_params"][0, 0, 76:], dtype=torch.float32)
    server = O.SMPLServerOracle(tables, betas)
    verts_c = server.verts_c.reshape(-1, 3) if hasattr(server, "verts_c") else None
    if verts_c is None:
        out = server.forward(torch.ones(1), torch.zeros(1, 3), O.canonical_thetas().view(1, -1), betas.view(1, -1))
        verts_c = out["smpl_verts"].reshape(-1, 3)
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, verts_c.shape[0], (n,), generator=g)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True) * (torch.rand(n, 1, generator=g) * 0.1)
    x = (verts_c[idx] + d).double()
    cond = torch.tensor(sc["smpl_params"][0, 0, 7:76], dtype=torch.float32) / np.pi
    ref = net(sd, prefix, x, cond, ["e"] * 9, act32=False)
    print(f"{n} canonical points within 0.1 of the body; sdf range [{float(ref.min()):.3f}, {float(ref.max()):.3f}]")
    fp32 = O.implicit_forward(sd, prefix, x.float(), cond, 6)[:, 0].double()
    print(f"{'fp32 torch (the oracle)':34s} max {float((fp32 - ref).abs().max()):.2e} mean {float((fp32 - ref).abs().mean()):.2e}")

    def show(name, codes):
        e = (net(sd, prefix, x, cond, codes) - ref).abs()
        print(f"{name:34s} max {float(e.max()):.2e} mean {float(e.mean()):.2e} p99.9 {float(e.quantile(0.999)):.2e}")

    for c in ("f16", "b3", "h3", "w2", "x2"):
        show("all layers " + c, [c] * 9)
    for l in range(9):
        cs = ["b3"] * 9
        cs[l] = "f16"
        show(f"b3, layer {l} f16", cs)
    for l in range(9):
        cs = ["f16"] * 9
        cs[l] = "b3"
        show(f"f16, layer {l} b3", cs)
    for l in range(9):
        cs = ["x2"] * 9
        cs[l] = "h3"
        show(f"x2, layer {l} h3", cs)
    for name, cs in (("h3 on 0,4; x2 else", ["h3", "x2", "x2", "x2", "h3", "x2", "x2", "x2", "x2"]),
                     ("h3 on 0,4,8; x2 else", ["h3", "x2", "x2", "x2", "h3", "x2", "x2", "x2", "h3"]),
                     ("h3 on 0..4; x2 else", ["h3"] * 5 + ["x2"] * 4),
                     ("h3 on 4..8; x2 else", ["x2"] * 4 + ["h3"] * 5),
                     ("h3 on 5..8; x2 else", ["x2"] * 5 + ["h3"] * 4),
                     ("b3 on 0,4,8; f16 else", ["b3", "f16", "f16", "f16", "b3", "f16", "f16", "f16", "b3"]),
                     ("b3 on 0,4,8; w2 else", ["b3", "w2", "w2", "w2", "b3", "w2", "w2", "w2", "b3"]),
                     ("h3 on 0,4,8; w2 else", ["h3", "w2", "w2", "w2", "h3", "w2", "w2", "w2", "h3"]),
                     ("b3 on 0..3; f16 else", ["b3"] * 4 + ["f16"] * 5),
                     ("b3 on 4..8; f16 else", ["f16"] * 4 + ["b3"] * 5),
                     ("b3 on 6..8; f16 else", ["f16"] * 6 + ["b3"] * 3),
                     ("b3 on 7,8; f16 else", ["f16"] * 7 + ["b3"] * 2)):
        show(name, cs)


if __name__ == "__main__":
    main()
