"""Point samplers for the eikonal term (reference code/lib/model/sampler.py:84-108)."""
import torch


class PointInSpace:
    def __init__(self, global_sigma=0.5, local_sigma=0.01):
        self.global_sigma = global_sigma
        self.local_sigma = local_sigma

    def get_points(self, pc_input=None, local_sigma=None, global_ratio=0.125):
        """One Gaussian-jittered point per input point plus `global_ratio` uniform points in [-sigma_g, sigma_g]^3."""
        b, n, d = pc_input.shape
        sigma = self.local_sigma if local_sigma is None else local_sigma
        local = pc_input + torch.randn_like(pc_input) * sigma
        glob = torch.rand(b, int(n * global_ratio), d, device=pc_input.device) * (2 * self.global_sigma) - self.global_sigma
        return torch.cat([local, glob], dim=1)
