"""Density modules with the reference's names (code/lib/model/density.py). Only `beta` is a learnable parameter; the
per-sample density evaluation of the hot path happens inside the sampler / compositing kernels (csrc/common.hpp)."""
import torch
import torch.nn as nn


class Density(nn.Module):
    def __init__(self, params_init={}):
        super().__init__()
        for p in params_init:
            setattr(self, p, nn.Parameter(torch.tensor(params_init[p])))

    def forward(self, sdf, beta=None):
        return self.density_func(sdf, beta=beta)


class LaplaceDensity(Density):
    def __init__(self, params_init={}, beta_min=0.0001):
        super().__init__(params_init=params_init)
        self.beta_min = float(beta_min)

    def get_beta(self):
        return self.beta.abs() + self.beta_min

    def density_func(self, sdf, beta=None):
        # convenience for callers outside the renderer; the renderer itself never calls this
        if beta is None:
            beta = self.get_beta()
        return (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


class AbsDensity(Density):
    def density_func(self, sdf, beta=None):
        return torch.abs(sdf)
