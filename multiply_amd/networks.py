"""Parameter containers for the SDF / colour MLPs with the reference's module and state-dict names.

Mirrors the constructor contract of ImplicitNet (reference code/lib/model/networks.py:7-116) and RenderingNet
(networks.py:223-261) for the configurations reachable from the shipped YAMLs:
  ImplicitNet : cond in {'smpl', 'frame', 'none'}, Fourier embedding, skip connection, geometric init, weight-norm
  RenderingNet: mode in {'pose_no_view', 'nerf_frame_encoding'}
so that checkpoints (`lin{l}.weight_g/weight_v/bias`, `lin_pose.*`) load unchanged and so that, under the same
torch seed, construction consumes the RNG exactly like the reference (nn.Linear default init first, then the
geometric overrides) and yields bit-identical initial weights (tests/test_oracle_golden.py proves this with the
checksum stored by tests/golden/make_golden.py).

The arithmetic itself does NOT live here: `forward` hands the effective weights to the HIP kernels
(multiply_amd/csrc) through multiply_amd.hip; there is no PyTorch fallback.
The triplane / person-encoder variants (networks.py:32-40, 86-116, 243-252) are out of scope.
"""
import math

import numpy as np
import torch
import torch.nn as nn


def embed_dim(d_in, multires):
    return d_in + d_in * 2 * multires if multires > 0 else d_in


def _weight_norm(lin):
    # old-style parametrisation on purpose: it creates the `weight_g` / `weight_v` names of the checkpoints
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return nn.utils.weight_norm(lin)


def effective_weight(lin):
    """g * v / ||v|| (row-wise) for weight-normed layers, the plain weight otherwise; fp32, differentiable."""
    if hasattr(lin, "weight_g"):
        v, g = lin.weight_v, lin.weight_g
        return v * (g / v.norm(dim=1, keepdim=True))
    return lin.weight


class ImplicitNet(nn.Module):
    def __init__(self, opt, betas=None):
        super().__init__()
        if opt.get("offset_head", False) or opt.get("beta_encoding", False) or opt.cond in ("smpl_id", "smpl_tri"):
            raise NotImplementedError("person-encoder / triplane / offset-head variants are outside the hot-path scope")
        self.opt = opt
        self.cond = opt.cond
        self.cond_dim = {"smpl": 69, "frame": 32, "none": 0}[self.cond]
        self.cond_layer = [0]
        self.skip_in = list(opt.skip_in)
        self.multires = int(opt.multires)
        self.d_in = int(opt.d_in)
        self.embed_dim = embed_dim(self.d_in, self.multires)
        widths = [self.embed_dim] + list(opt.dims) + [opt.d_out + opt.feature_vector_size]
        self.dims = widths
        self.num_layers = len(widths)
        for l in range(self.num_layers - 1):
            n_out = widths[l + 1] - widths[0] if (l + 1) in self.skip_in else widths[l + 1]
            n_in = widths[l] + (self.cond_dim if (self.cond != "none" and l in self.cond_layer) else 0)
            lin = nn.Linear(n_in, n_out)
            if opt.init == "geometry":
                self._geometric_init(lin, l, widths, n_out, opt)
            elif opt.init == "zero" and l == self.num_layers - 2:
                nn.init.constant_(lin.bias, 0.0)
                nn.init.uniform_(lin.weight, -1e-5, 1e-5)
            if opt.weight_norm:
                lin = _weight_norm(lin)
            setattr(self, f"lin{l}", lin)

    def _geometric_init(self, lin, l, widths, n_out, opt):
        """SAL/IGR geometric initialisation: the network starts as the SDF of a sphere of radius `bias`."""
        last = l == self.num_layers - 2
        if last:
            nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(widths[l]), std=0.0001)
            nn.init.constant_(lin.bias, -opt.bias)
            return
        nn.init.constant_(lin.bias, 0.0)
        std = np.sqrt(2) / np.sqrt(n_out)
        if self.multires > 0 and l == 0:
            nn.init.constant_(lin.weight[:, 3:], 0.0)
            nn.init.normal_(lin.weight[:, :3], 0.0, std)
        elif self.multires > 0 and l in self.skip_in:
            nn.init.normal_(lin.weight, 0.0, std)
            nn.init.constant_(lin.weight[:, -(widths[0] - 3):], 0.0)
        else:
            nn.init.normal_(lin.weight, 0.0, std)

    def layers(self):
        return [getattr(self, f"lin{l}") for l in range(self.num_layers - 1)]

    def forward(self, input, cond, current_epoch=None, person_id=-1):
        """(N, d_in) or (1, N, d_in) points -> (1, N, 1 + feature_vector_size), like the reference (networks.py:126-208).

        Used by callers outside the fused renderer (e.g. the mesh-extraction query of multiply_model.py:941-945)."""
        from . import hip
        if input.ndim == 2:
            input = input.unsqueeze(0)
        nb, npnt, nd = input.shape
        if nb * npnt == 0:
            return input
        cvec = None if self.cond == "none" else cond[self.cond].reshape(-1)
        out = hip.implicit_forward(self, input.reshape(-1, nd), cvec)
        return out.reshape(nb, npnt, -1)


class RenderingNet(nn.Module):
    def __init__(self, opt, triplane=None):
        super().__init__()
        self.mode = opt.mode
        if self.mode not in ("pose_no_view", "nerf_frame_encoding"):
            raise NotImplementedError(f"rendering mode {self.mode} is outside the hot-path scope")
        widths = [opt.d_in + opt.feature_vector_size] + list(opt.dims) + [opt.d_out]
        self.multires_view = int(opt.multires_view)
        if self.multires_view > 0:
            widths[0] += embed_dim(3, self.multires_view) - 3
        if self.mode == "nerf_frame_encoding":
            widths[0] += 32
        if self.mode == "pose_no_view":
            self.dim_cond_embed = 8
            self.cond_dim = 69
            self.lin_pose = nn.Linear(self.cond_dim, self.dim_cond_embed)
        self.dims = widths
        self.num_layers = len(widths)
        for l in range(self.num_layers - 1):
            lin = nn.Linear(widths[l], widths[l + 1])
            if opt.weight_norm:
                lin = _weight_norm(lin)
            setattr(self, f"lin{l}", lin)

    def layers(self):
        return [getattr(self, f"lin{l}") for l in range(self.num_layers - 1)]

    def forward(self, points, normals, view_dirs, body_pose, feature_vectors, frame_latent_code=None,
                id_latent_code=None, person_id=-1, tri_feat=None):
        from . import hip
        return hip.rendering_forward(self, points, normals, view_dirs, body_pose, feature_vectors, frame_latent_code)
