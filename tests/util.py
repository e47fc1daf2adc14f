"""shared helpers of the parity tests"""
import numpy as np
import torch

from oracle import config as OC, model as OM


def rel_err(out: torch.Tensor, ref: torch.Tensor) -> float:
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    return ((out - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def max_abs(out, ref) -> float:
    return (out.detach().float().cpu() - ref.detach().float().cpu()).abs().max().item()


def seeded(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def to_pkg_cfg(pkg, ocfg):
    return pkg.UNetConfig(ocfg.adm_in_channels, ocfg.model_channels, list(ocfg.channel_mults), ocfg.n_head_channels,
                          list(ocfg.transformer_depths), ocfg.context_dim, ocfg.in_channels, ocfg.out_channels,
                          ocfg.is_refiner)


def to_pkg_vcfg(pkg, v):
    return pkg.VAEConfig(list(v.enc_channels), list(v.dec_channels), v.n_group, v.enc_out_channels, v.scale_factor)


def unet_weights(ocfg, seed=0):
    return OM.to_torch(OC.synth_weights(OC.unet_param_specs(ocfg), seed))
