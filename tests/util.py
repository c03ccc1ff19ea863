"""Shared helpers for the parity tests."""
import warnings

import numpy as np
import torch


def state_checksum(sd):
    tot = 0.0
    for k in sorted(sd):
        tot += float(sd[k].double().abs().sum()) + 3.0 * float(sd[k].double().sum())
    return tot


def seeded_networks(num_person=2, seed=0):
    """The scene networks built under torch.manual_seed(seed) in the reference's construction order
    (multiply.py:53-66); returns a plain nn.Module whose state_dict has the reference's key names."""
    warnings.filterwarnings("ignore")
    from multiply_amd.config import load_config
    from multiply_amd.networks import ImplicitNet, RenderingNet
    opt = load_config()
    torch.manual_seed(seed)
    m = torch.nn.Module()
    m.foreground_implicit_network_list = torch.nn.ModuleList()
    m.foreground_rendering_network_list = torch.nn.ModuleList()
    for _ in range(num_person):
        m.foreground_implicit_network_list.append(ImplicitNet(opt.implicit_network))
        m.foreground_rendering_network_list.append(RenderingNet(opt.rendering_network))
    m.bg_implicit_network = ImplicitNet(opt.bg_implicit_network)
    m.bg_rendering_network = RenderingNet(opt.bg_rendering_network)
    m.frame_latent_encoder = torch.nn.Embedding(opt.num_training_frames, opt.dim_frame_encoding)
    m.density = torch.nn.Module()
    m.density.beta = torch.nn.Parameter(torch.tensor(opt.density.params_init.beta))
    return m, opt


def t32(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)
