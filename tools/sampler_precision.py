"""Sampler precision experiment (round-4 review, item 4): what do near-fp32 SDF queries in the SAMPLER buy?

The product's sampler takes its depths from half-precision (f16 MFMA) network queries; a shifted sample can cross the
reference's `dist > 0.1 => sdf = 4` discontinuity (multiply.py:142-143) and a grazing ray's opacity moves by up to 0.1.  This
tool renders the always-on 4 096-ray slice of the headline frame (tests/test_headline_slow_gpu.py: 8 convergence groups spread
over the 512x512 two-person frame, N_samples 128) once per sampler arithmetic --
    f16     the fused kernel k_mlp_sdf (product)
    bf16x3  the same worklists through mp_tf_sdf_val: the value sweep of the training path's layer-fused kernel (split-bfloat16
            products, fp32 activations; Multiply.sampler_sdf_mode); `bf16x3-layerwise`: the same arithmetic layer by layer
-- and compares depths and pixels with the fp32 oracle on the same hit sets.  Shading stays on the product's f16 kernels in
both runs, so the difference between the rows is the sampler's arithmetic alone.
    python tools/sampler_precision.py [n_groups] [beta]  ->  gpurun_out/sampler_precision[_beta<b>].txt
beta (default: the initial 0.1 of confs/model.yaml): the Laplace density's scale.  At the initial 0.1 the density at the outlier
boundary (distance 0.1 from the body, where the reference switches to sdf = 4, multiply.py:142-143) is still 1.8 per unit length, so
a sample the f16 sampler shifts across that boundary moves a grazing ray's opacity; a trained model's beta (~0.01) makes the
density there e^-10 of its surface value.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from oracle import multiply_oracle as O         # noqa: E402


def main():
    n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    beta = float(sys.argv[2]) if len(sys.argv) > 2 else None
    model, inp, tables, sc = bench.build_model(128)
    if beta is not None:
        with torch.no_grad():
            model.density.beta.fill_(beta)
    model.convergence_group = 512
    R = inp["uv"].shape[1]
    gin = bench.to_dev(inp)
    model(gin)                                   # the frame's own cull: hit sets of every group
    torch.cuda.synchronize()
    hit = [model._last["per"][p]["hit_index"][:n].long().cpu() for p, n in zip(model._last["persons"], model.last_stats["n_hit"])]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    oracle = O.MultiplyOracle(sd, tables, sc["smpl_params"][0, :, 76:], O.SamplerCfg(N_samples=128, N_samples_eval=128))
    groups = np.unique(np.linspace(0, R // 512 - 1, n_groups).round().astype(int))
    keys = ("rgb_values", "acc_map", "normal_values", "fg_rgb_values")
    want, subs = [], []
    t0 = time.time()
    for g in groups:
        c0 = int(g) * 512
        sub = dict(inp)
        sub["uv"] = inp["uv"][:, c0:c0 + 512]
        hg = [h[(h >= c0) & (h < c0 + 512)] - c0 for h in hit]
        hg = [h if len(h) else torch.zeros(1, dtype=torch.long) for h in hg]
        want.append(oracle.forward_eval(sub, hg))
        subs.append((sub, hg))
    t_or = time.time() - t0
    lines = [f"sampler precision: {512 * len(groups)} rays = {len(groups)} convergence groups of the headline frame (2 persons, 512x512, "
             f"N_samples 128), density beta {float(model.density.beta):.3g}, fp32 oracle {t_or:.0f} s; shading on the product's f16 kernels in every row",
             f"{'sampler sdf':10s} {'z max':>9s} {'z mean':>9s} | {'acc max':>9s} {'acc>1e-2':>8s} {'acc>3e-3':>8s} | {'nrm max':>9s} {'nrm>1e-2':>8s} | "
             f"{'rgb max':>9s} | sampler-sdf ms (these rays)"]
    for mode in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("f16", "f16x2", "bf16x3")):
        model.sampler_sdf_mode = mode
        zerr, err = [], {k: [] for k in keys}
        model.profile = True
        model.phase_events = {}
        for (sub, hg), w in zip(subs, want):
            got = model({**bench.to_dev(sub), "hit_index": hg})
            torch.cuda.synchronize()
            for p in range(len(hg)):
                zo = torch.cat([w["z_vals"][p], w["z_max"][p][:, None]], 1)
                zerr.append((model._last["per"][p]["zfinal"][:zo.shape[0]].cpu() - zo).abs().reshape(-1))
            for k in keys:
                e = (got[k].cpu().double() - w[k].double()).abs().nan_to_num()
                err[k].append(e.reshape(e.shape[0], -1).max(1).values)
        ph = model.phase_times_ms()
        model.profile = False
        z = torch.cat(zerr)
        e = {k: torch.cat(v) for k, v in err.items()}
        lines.append(f"{mode[:10]:10s} {float(z.max()):9.2e} {float(z.mean()):9.2e} | {float(e['acc_map'].max()):9.2e} "
                     f"{int((e['acc_map'] > 1e-2).sum()):8d} {int((e['acc_map'] > 3e-3).sum()):8d} | {float(e['normal_values'].max()):9.2e} "
                     f"{int((e['normal_values'] > 1e-2).sum()):8d} | {float(e['rgb_values'].max()):9.2e} | {ph.get('sampler_mlp_sdf', (0, 0.0))[1]:.2f}")
    model.sampler_sdf_mode = "f16"
    lines.append("(f16x2: split activations on the half-precision weights, mp_mlp_sdf_x2; bf16x3: mp_tf_sdf_val; bf16x3-layerwise: the same "
                 "arithmetic through the layer-wise GEMMs with a host read per iteration)")
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/sampler_precision.txt" if beta is None else f"gpurun_out/sampler_precision_beta{beta:g}.txt", "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
