"""Stated tolerances of the GPU parity tests, in ONE place: (max over elements, mean over elements) of |HIP - oracle|.

The MLPs run with f16 MFMA operands and fp32 accumulation, everything else is fp32; the oracle is fp32 throughout.  Each
bound is <= 5x the largest value MEASURED on MI355X over the test scenes (the measured figures are quoted beside it and in
DESIGN.md §4), so that an order-of-magnitude regression of a kernel turns the suite red."""

def Dist(mean, bulk, frac, hard, p999, floor=None):
    """distribution bound of a transmittance-like quantity: element mean < `mean`; at most the fraction `frac` of the rays
    (never fewer than `floor` rays; default SMALL_SAMPLE_RAYS) may have an element off by more than `bulk`; the 99.9th percentile
    of the elements below `p999`; no element off by more than `hard`"""
    return dict(mean=mean, bulk=bulk, frac=frac, hard=hard, p999=p999, floor=floor)


# Transmittance-like quantities (opacities, normals, the foreground colour over a white background): a ray that GRAZES a
# surface has one sample whose alpha flips between ~0 and ~1 with the f16 rounding of its sdf, so single rays can be off by
# several 1e-2 while everything else agrees to 1e-3.  A worst-case bound alone would have to be ~0.1 and assert nothing:
# these are bounded by their DISTRIBUTION instead.  Measured on MI355X (headline 1 024-ray scene / the 16 384-ray headline run,
# profiles/r03_parity_16k.txt): 0.22-0.4 % of the rays above 1e-2 (99th percentile 6.6e-4, 99.9th 1.2e-2 ... 3.2e-2), worst
# single ray 9.0e-2 (16k rays) / 1.1e-1 (the always-on 4 096-ray slice, profiles/r04_parity_4k.txt).  Round 4 (review): the fraction at 2.5x the measured one (was 9x), the worst case at 1.33x, and a bound on
# the 99.9th percentile per quantity (2-2.5x the 16k-ray figure; on a 1 024-ray scene it is the second-worst element).
# Round 5 (advisor): `hard` had 9 % of margin over the worst single ray ever measured (1.1e-1) -- one grazing ray moving with a
# kernel's worklist / atomics order would have made the suite flaky while saying nothing about quality.  The statistical power
# is in `frac` and `p999`; `hard` is the gross-failure bound (a wrong pixel is O(1)), with real margin.
_GRAZE = dict(bulk=1e-2, frac=0.01, hard=0.2)
SMALL_SAMPLE_RAYS = 6                     # floor of the allowed number of rays above `bulk` (scenes of < 600 rays)
P999_MIN_RAYS = 2000                      # the p99.9 bound applies to samples of at least this many rays (see within)
EVAL_F16 = {                               # eval-mode Multiply.forward outputs, per pixel, with the HALF-PRECISION sampler (sampler_sdf_mode = 'f16': the opt-out)
    # largest measured over the test scenes   (max, mean)
    "rgb_values": (8e-3, 3e-5),             # 1.7e-3, 1.3e-5   (headline N = 128 scene; 16k rays: 3.0e-3, 4.9e-6)
    "fg_rgb_values": Dist(1e-3, p999=0.04, **_GRAZE),  # mean 2.4e-4; 3 of 1024 rays above 1e-2; p99.9 1.6e-2; fg + T_bg: the transmittance error, undamped
    "acc_map": Dist(1e-3, p999=0.08, **_GRAZE),        # mean 4.8e-4; 4 of 1024 rays above 1e-2, worst 5.3e-2; 16k: p99.9 3.2e-2, worst 9.0e-2
    "acc_person_list": Dist(1e-3, p999=0.05, **_GRAZE),   # mean 2.7e-4; 4 of 1024; 16k: p99.9 1.6e-2
    "bg_transmittance": Dist(1e-3, p999=0.08, **_GRAZE),  # mean 1.2e-4 (= 1 - acc of the last sample's exclusive transmittance)
    "normal_values": Dist(8e-4, p999=0.04, **_GRAZE),  # mean 3.2e-4; 3 of 1024 rays above 1e-2, worst 3.9e-2; 16k: p99.9 1.2e-2
    "bg_rgb": (1.5e-4, 2.5e-5),             # 2.7e-5, 5.4e-6
}
# Round 6: the DEFAULT sampler is the near-fp32 one (sampler_sdf_mode 'auto' -> 'bf16x3'); the grazing-ray tail the distribution bounds
# above exist for is gone with it (profiles/r05_parity_16k.txt: 36 rays of 16 384 above 1e-2 -> 1), so the default path is bounded
# by maxima again: EVAL.  The bounds are <= 5x the maxima measured over the test scenes in round 6 (profiles/r06_gpu_suite.txt).
# Largest values measured over every scene of the suite (profiles/r06_gpu_suite.txt; 80 ... 4 096 rays, 1 ... 8 persons, trained and
# initial weights): acc_map / acc_person / bg transmittance max 1.14e-3, mean 1.25e-4; normals 1.33e-3, 1.7e-4; fg colour 5.9e-4, 5.1e-5;
# pixels 4.4e-5, 8.4e-6; NOT ONE ray of any scene above 3e-3.  `hard` covers the one ray in 16 384 the opt-in 16k run finds at 1.1e-2
# (profiles/r05_parity_16k.txt): <= 2.5x that; the tightness is in `bulk` / `frac` (at most max(1, 5e-4 x rays) rays above 3e-3),
# `p999` and `mean`.
_TIGHT = dict(bulk=3e-3, frac=5e-4, hard=2.5e-2, floor=1)
EVAL = {
    "rgb_values": (1e-3, 3e-5),             # 4.4e-5, 8.4e-6 over the suite's scenes; 3.5e-4 on the opt-in 16 384-ray run (profiles/r06_parity_16k.txt)
    "fg_rgb_values": Dist(2e-4, p999=2.5e-3, **_TIGHT),
    "acc_map": Dist(5e-4, p999=5e-3, **_TIGHT),
    "acc_person_list": Dist(3e-4, p999=5e-3, **_TIGHT),
    "bg_transmittance": Dist(5e-4, p999=5e-3, **_TIGHT),
    "normal_values": Dist(6e-4, p999=5e-3, **_TIGHT),
    "bg_rgb": (1.5e-4, 2.5e-5),
}
# eval outputs with the sampler's network queries at near-fp32 precision (Multiply.sampler_sdf_mode = 'bf16x3'; shading still f16).
# Measured (profiles/r05_parity_16k.txt, 16 384 rays of the headline frame): acc_map 1 ray of 16 384 above 3e-3 (1.1e-2; the f16
# sampler: 36 rays above 1e-2, worst 9.0e-2), normals 3 elements above 3e-3 (worst 9.7e-3), pixels 3.5e-4; on the always-on 4 096- and
# 1 024-ray slices nothing above 1.2e-3.  Bounds: `hard` <= 5x the 16k maxima, and at most max(1, frac x rays) rays with an element
# above `bulk` -- 5e-4 of the rays where the f16 sampler's distribution bound allows 1e-2 of them above 1e-2.
EVAL_PRECISE = {"rgb_values": dict(hard=1.5e-3, bulk=3e-3, frac=5e-4), "acc_map": dict(hard=5e-2, bulk=3e-3, frac=5e-4),
                "acc_person_list": dict(hard=5e-2, bulk=3e-3, frac=5e-4), "normal_values": dict(hard=5e-2, bulk=3e-3, frac=5e-4),
                "fg_rgb_values": dict(hard=2.5e-2, bulk=3e-3, frac=5e-4)}


def within_precise(err, tol):
    """err: |got - want| per element, rows = rays (NaNs already zeroed); tol: an EVAL_PRECISE entry"""
    import torch
    e = torch.as_tensor(err).double()
    ray = e.reshape(e.shape[0], -1).max(dim=1).values if e.dim() > 1 else e
    return float(e.max()) < tol["hard"] and int((ray > tol["bulk"]).sum()) <= max(1, int(tol["frac"] * ray.numel()))


# depths in that mode: mean, and the fraction of the RAYS with a depth off by more than 3e-3 -- not a maximum: where the CDF is flat (no
# weight) an inverse-CDF depth moves by centimetres with the last bit of an sdf (measured on the 1 024-ray headline scene: mean
# 3.9e-7 / 2.6e-5, one ray of 479 at 4.0e-2; with the f16 sampler 4.8e-5 / 3.0e-4, 12 / 101 rays above 3e-3, worst 0.22)
Z_VALS_PRECISE = dict(mean=1.3e-4, bulk=3e-3, frac=0.01)
Z_VALS_F16 = (5e-2, 3e-4)                   # sampler depths with the f16 sampler (inverse CDF of f16 sdf queries); measured 1.4e-2, 6.7e-5
Z_VALS = (5e-3, 1e-5)                       # the default (near-fp32) sampler's depths on the small scenes; measured 4.5e-4, 2.1e-6
TRAIN_Z_VALS = (6e-3, 3e-5)                 # training-mode depths (stratified / random draws), default sampler; measured 1.2e-3, 6.5e-6
TRAIN_Z_VALS_PRECISE = TRAIN_Z_VALS
# the device's training-mode sampler against the ORACLE'S OWN sampler on the same draws (bench.py parity_sampler_depth_*): rays with a depth
# off by more than 3e-3 (flat stretches of a CDF).  The device's depths are the same bits on every run (profiles/r06_determinism.txt); the
# oracle's move with the host it runs on: measured 5, 5, 6 and 9 of 787 rays on four boxes (512-ray iteration), 2 and 4 of ~190 (128 rays).
TRAIN_DEPTH_RAYS = dict(frac=0.03, floor=6)
TRAIN_Z_VALS_F16 = (0.15, 1e-3)             # the same with sampler_sdf_mode = 'f16'; measured 3e-2, 2e-4
MLP = {                                     # the fused MLP kernels on random points vs the fp32 oracle: max |err|
    "fg_sdf": 5e-3, "fg_feat": 7e-3,        # 9.8e-4, 1.4e-3
    "fg_sdf_x2": 1.5e-3,                    # split-activation kernel (mp_mlp_sdf_x2) on points of the whole cube: 3.2e-4 (f16 kernel: 1.0e-3)
    "bg_sdf": 1e-4, "bg_feat": 2.5e-4,      # 2.0e-5, 5.3e-5
    "shade_sdf": 2.5e-3, "shade_normal": 1.5e-2, "shade_rgb": 3e-5,   # 5.4e-4, 3.0e-3, 6.0e-6
    "normal_rev_vs_fwd": 8e-3,              # 1.7e-3
    "bg_rgb": 1.6e-4,                       # 3.3e-5
}
# training-mode outputs vs the fp32 oracle, max |err|.  The differentiable path runs its GEMMs either on the exact-fp32 matrix
# instruction (MP_TRAIN_PRECISION=f32: fp32 on both sides, summation order only; measured 7e-7 ... 1.8e-6) or -- the default --
# on split-bfloat16 products (three 16-bit MFMAs per product, ~2^-16 relative each; measured rgb 1.4e-6, acc 6e-6 (121 rays) /
# acc_person 5.2e-5 (the 512-ray bench workload after ten Adam steps), grad_theta 1.2e-5)
TRAIN_FWD_BY_PRECISION = {
    # acc_person_list, round 6: two persons' samples at (nearly) equal depth may be merged in either order -- the per-person opacities then
    # move by the product of two alphas while their sum does not, and the composited normal by that product times the difference of the two
    # normals (measured 4.0e-5 / 4.1e-5 on one ray of 121, acc_map 2.0e-6 on the same run)
    "f32": {"rgb_values": 8e-6, "acc_map": 8e-6, "acc_person_list": 2e-4, "grad_theta": 5e-6, "normal_values": 2e-4},
    "bf16x3": {"rgb_values": 8e-6, "acc_map": 5e-5, "acc_person_list": 2e-4, "grad_theta": 5e-5, "normal_values": 2e-4},
}


def train_fwd():
    from multiply_amd import train
    return TRAIN_FWD_BY_PRECISION[train.TRAIN_PRECISION]


TRAIN_FWD = TRAIN_FWD_BY_PRECISION["bf16x3"]
TRAIN_GRAD_REL = 5e-3                       # per-tensor relative L2 error of a parameter gradient; measured 4.8e-4
TRAIN_GRAD_REL_RENDERING = 2e-2             # colour nets: single ReLU masks flip between summation orders


class Stats(tuple):
    """(max, mean) over the ELEMENTS of |got - want|, with the error of every element (`.err`) and the worst element of every
    ray / row (`.ray`) attached"""

    def __new__(cls, err, ray=None):
        st = tuple.__new__(cls, (float(err.max()) if err.numel() else 0.0, float(err.mean()) if err.numel() else 0.0))
        st.err = err
        st.ray = err if ray is None else ray
        return st


def torch_quantile(e, q):
    """torch.quantile without its 16 M element limit (sorts; the k-th order statistic, linear interpolation)"""
    import torch
    v, _ = torch.sort(e)
    pos = q * (v.numel() - 1)
    lo = int(pos)
    hi = min(lo + 1, v.numel() - 1)
    return v[lo] + (v[hi] - v[lo]) * (pos - lo)


def within(stats, tol):
    """tol = (max, mean) over elements, or a Dist {mean, bulk, frac, hard}: element mean below `mean`, at most the fraction
    `frac` of the RAYS with an element above `bulk`, and nothing above `hard`."""
    if isinstance(tol, dict):
        ray = stats.ray
        n = max(ray.numel(), 1)
        # the fraction is a RATE: on the few hundred rays of the small scenes `frac * n` is 2-3 rays and a single grazing ray more
        # or less would decide the test (measured: 4 of 256 in the 2-rank scene) -- never fewer than SMALL_SAMPLE_RAYS are allowed
        allowed = max(int(-(-tol["frac"] * n // 1)), SMALL_SAMPLE_RAYS if tol.get("floor") is None else tol["floor"])
        e = stats.err.reshape(-1).float()
        # the 99.9th percentile is a statement about a DISTRIBUTION: below ~2 000 rays it is the maximum under another name (one
        # grazing ray of the 144-ray pipeline scene decides it: measured 4.1e-2 there against 4.0e-2), so it is asserted on the
        # samples large enough to have a 0.1 % tail (the 4 096-ray headline slice, the full-size frame); `hard` bounds the rest
        p999 = float(torch_quantile(e, 0.999)) if (e.numel() and n >= P999_MIN_RAYS) else 0.0
        return stats[1] < tol["mean"] and int((ray > tol["bulk"]).sum()) <= allowed and stats[0] < tol["hard"] and p999 <= tol["p999"]
    return stats[0] < tol[0] and stats[1] < tol[1]
