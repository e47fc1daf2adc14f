"""Sampler / latent decoder restatement -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows /root/reference/src/model/stablediffusion/mod.rs.  The reference draws noise from an *unseeded*
libtorch generator (gen_noise :378-388), so parity is defined on explicit noise inputs: every function that
consumes randomness takes the noise tensors as arguments.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from .config import UNetConfig, VAEConfig
from .model import NUM, decode_latent_vae, encode_image_vae, unet_forward

Tensor = torch.Tensor


@dataclass
class Conditioning:
    """stablediffusion/mod.rs:544-555 (rank-reduced unconditional tensors exactly as the reference holds them)."""
    unconditional_context_full: Tensor        # [77, ctx_full]
    unconditional_context_open_clip: Tensor   # [77, 1280]
    context_full: Tensor                      # [B, 77, ctx_full]
    context_open_clip: Tensor                 # [B, 77, 1280]
    unconditional_channel_context: Tensor     # [adm]
    unconditional_channel_context_refiner: Tensor
    channel_context: Tensor                   # [B, adm]
    channel_context_refiner: Tensor
    resolution: tuple                         # (height, width)


def step_schedule(n_steps: int, step_start: int = 0, n_train: int = 1000) -> List[int]:
    """(0..n_train-step_start).rev().step_by(n_train / n_steps)  (:400-406).  n_steps=30 -> 31 values."""
    step_size = n_train // n_steps
    return list(range(n_train - step_start - 1, -1, -step_size))


class Diffuser:
    """stablediffusion/mod.rs:308-542."""

    def __init__(self, cfg: UNetConfig, W, alphas_cumprod: np.ndarray, n_steps: int = 1000):
        self.cfg, self.W = cfg, W
        self.alphas = np.asarray(alphas_cumprod)
        self.n_steps = n_steps
        self.is_refiner = cfg.is_refiner

    def get_alpha(self, i: int) -> float:       # :485-492  (scalar read back as f64)
        return float(self.alphas[i])

    def forward_diffuser(self, latent, t: int, cond: Conditioning, cfg_scale: float) -> Tensor:
        """:494-541 -- two separate UNet forwards, refiner returns the conditional branch only."""
        n = latent.shape[0]
        ts = torch.full((n,), t, dtype=torch.int64)
        if not self.is_refiner:
            uctx, ctx, uy, y = (cond.unconditional_context_full, cond.context_full,
                                cond.unconditional_channel_context, cond.channel_context)
        else:
            uctx, ctx, uy, y = (cond.unconditional_context_open_clip, cond.context_open_clip,
                                cond.unconditional_channel_context_refiner, cond.channel_context_refiner)
        c = unet_forward(self.cfg, self.W, latent, ts, ctx, y)
        if self.is_refiner:
            return c
        u = unet_forward(self.cfg, self.W, latent, ts, uctx[None].repeat(n, 1, 1), uy[None].repeat(n, 1))
        return NUM.r(u + NUM.r(NUM.r(c - u) * cfg_scale))      # (f16ref: three ops on f16 tensors, :538-540)

    def diffuse_latent(self, latent, cond, step_start, n_steps, cfg_scale, trace: Optional[list] = None):
        """:390-432 (sigma = 0, so the per-step gen_noise()*sigma term vanishes)."""
        step_size = self.n_steps // n_steps
        for t in step_schedule(n_steps, step_start, self.n_steps):
            a_t = self.get_alpha(t)
            a_prev = self.get_alpha(t - step_size) if t >= step_size else 1.0
            sqrt_noise = (1.0 - a_t) ** 0.5
            eps = self.forward_diffuser(latent, t, cond, cfg_scale)
            r = NUM.r      # identity in fp32; f16ref: the Diffuser's backend is LibTorch<f16> too (sample/main.rs:122), every op output an f16 tensor
            predx0 = r(r(latent - r(eps * sqrt_noise)) / (a_t ** 0.5))
            latent = r(r(predx0 * (a_prev ** 0.5)) + r(eps * ((1.0 - a_prev) ** 0.5)))
            if trace is not None:
                trace.append(latent.clone())
        return latent

    def sample_latent(self, cond, cfg_scale, n_steps, noise0, trace=None):
        """:317-332; noise0 plays gen_noise()."""
        return self.diffuse_latent(noise0, cond, 0, n_steps, cfg_scale, trace)

    def refine_latent(self, latent, cond, cfg_scale, step_start, n_steps, noise, trace=None):
        """:355-376."""
        a = self.get_alpha(self.n_steps - step_start)
        noised = latent * (a ** 0.5) + noise * ((1.0 - a) ** 0.5)
        return self.diffuse_latent(noised, cond, step_start, n_steps, cfg_scale, trace)

    def sample_latent_with_inpainting(self, cond, cfg_scale, n_steps, reference, mask, noise0, step_noise,
                                      trace=None):
        """:334-353 + :434-483.  mask True -> keep the generated latent, False -> re-noised reference
        (mask_where(mask, latent), :465).  step_noise[i] plays the per-step gen_noise() of :463."""
        latent = noise0
        step_size = self.n_steps // n_steps
        for i, t in enumerate(step_schedule(n_steps, 0, self.n_steps)):
            a_t = self.get_alpha(t)
            a_prev = self.get_alpha(t - step_size) if t >= step_size else 1.0
            sqrt_noise = (1.0 - a_t) ** 0.5
            noised_ref = reference * (a_t ** 0.5) + step_noise[i] * sqrt_noise
            latent = torch.where(mask, latent, noised_ref)
            eps = self.forward_diffuser(latent, t, cond, cfg_scale)
            predx0 = (latent - eps * sqrt_noise) / (a_t ** 0.5)
            latent = predx0 * (a_prev ** 0.5) + eps * ((1.0 - a_prev) ** 0.5)
            if trace is not None:
                trace.append(latent.clone())
        return latent


class LatentDecoder:
    """stablediffusion/mod.rs:193-267."""

    def __init__(self, cfg: VAEConfig, W):
        self.cfg, self.W = cfg, W

    def decode_latent(self, x: Tensor) -> Tensor:          # :263-266
        return decode_latent_vae(self.cfg, self.W, x * (1.0 / self.cfg.scale_factor))

    def encode_image(self, x: Tensor) -> Tensor:            # :257-261
        return encode_image_vae(self.cfg, self.W, x) * self.cfg.scale_factor

    def latent_to_image(self, latent: Tensor) -> np.ndarray:
        """:200-237 -> u8 [B,H,W,3]: ((img+1)/2)*255, clamp to [0,255] in f64, truncating cast."""
        img = self.decode_latent(latent)
        img = ((img + 1.0) / 2.0).permute(0, 2, 3, 1) * 255.0
        return img.double().clamp(0.0, 255.0).to(torch.uint8).numpy()

    def image_to_latent(self, images_u8: np.ndarray) -> Tensor:
        """:239-255: u8 [B,H,W,3] -> /255 -> NCHW -> *2-1 -> encode."""
        x = torch.from_numpy(images_u8.astype(np.float32)) / 255.0
        x = x.permute(0, 3, 1, 2) * 2.0 - 1.0
        return self.encode_image(x)
