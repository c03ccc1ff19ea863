#!/usr/bin/env python
"""GPU box: how the compositing weights of the shaded points of the headline frame are distributed (round 6).  Renders bench.py's
frame, recomputes every person's own weights  w = T alpha  from the rendered sdf and depths (ignoring the other person: an upper bound
of the merged weight) and counts the points on the network work list by weight."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                     # noqa: E402


def main():
    model, inp, tables, sc = bench.build_model(128)
    with torch.no_grad():
        model(bench.to_dev(inp))
        torch.cuda.synchronize()
    L = model._last
    beta = float(model.density.get_beta())
    print(f"beta {beta:.5f}")
    for n, p in enumerate(L["persons"]):
        pp = L["per"][p]
        Rp = int(model.last_stats["n_hit"][n])
        z = pp["zfinal"][:Rp].double()
        S = z.shape[1] - 1
        sdf = pp["sdf"].view(-1, S)[:Rp].double()
        dt = z[:, 1:] - z[:, :-1]
        sig = (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))
        fe = sig * dt
        alpha = 1 - torch.exp(-fe)
        T = torch.exp(-(torch.cumsum(fe, 1) - fe))
        w = (T * alpha).float()
        nw = int(pp["wc2"][0])
        ids = pp["work2"][:nw].long()
        listed = torch.zeros(w.numel(), dtype=torch.bool, device=w.device)
        listed[ids[ids < w.numel()]] = True
        wl = w.reshape(-1)[listed]
        acc = w.sum(1)
        print(f"person {p}: {Rp} rays in the box, {S} samples each, {nw} points on the work list = {nw / (Rp * S):.3f} of all")
        for eps in (0.0, 1e-12, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4):
            print(f"   listed points with w <= {eps:7.0e}: {float((wl <= eps).float().mean()):.4f}")
        for a in (1e-6, 1e-4, 1e-2, 0.5):
            print(f"   rays with opacity < {a:6.0e}: {float((acc < a).float().mean()):.4f}")
        # what a per-sample threshold costs: the dropped weight per ray
        for eps in (1e-8, 1e-7, 1e-6):
            lost = torch.where(w <= eps, w, torch.zeros_like(w)).sum(1)
            print(f"   threshold {eps:.0e}: dropped weight per ray max {float(lost.max()):.3e} mean {float(lost.mean()):.3e}")


if __name__ == "__main__":
    main()
