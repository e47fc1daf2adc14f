"""Generates the round-3 fixtures tests/golden/fullsize_{refiner1024,refine1024,encode1024,inpaint1024,unet1024_f16w}.npz:
the oracle at BASELINE configs[3] (refiner) and configs[4] (inpainting).  TEST INFRASTRUCTURE.

    python -m oracle.make_golden_r3 [refiner1024] [refine1024] [encode1024] [inpaint1024] [unet1024_f16w]   (default: all)

Like oracle/make_golden_fullsize.py these are outputs of the ORACLE (fp32 torch-CPU restatement of the reference graph; the
reference itself -- Rust + burn + libtorch -- cannot be built here: parity unpinned) on the full SDXL architectures with the
seeded synthetic weights (seed 0) the HIP fill kernel reproduces bit for bit.

  fullsize_refiner1024.npz     one refiner UNet::forward at 1024x1024 (latent 128x128): 4 levels, 384/768/1536/1536 channels,
                               transformer depth 4, context 1280, adm 2560 (python/unet.py:163-200,233-270;
                               stablediffusion/mod.rs:528-530)
  fullsize_refine1024.npz      Diffuser::refine_latent at full size, 2 refiner iterations: step_start 800, n_steps 10 ->
                               (0..200).rev().step_by(100) = t 199, 99 (stablediffusion/mod.rs:355-376,400-406); per-step latents
  fullsize_encode1024.npz      LatentDecoder::image_to_latent of a 1024x1024 u8 image: /255*2-1 -> Encoder::forward (PaddedConv2d
                               downsamples) -> channels 0..4 * 0.13025 (stablediffusion/mod.rs:239-261; autoencoder/mod.rs:59-65,
                               131-144,384-407)
  fullsize_inpaint1024.npz     Diffuser::sample_latent_with_inpainting at 1024x1024, n_steps=4 (t = 999, 749, 499, 249), CFG 7.5,
                               mask = latent rows 0..25 keep the generated content (the 200 px crop of BASELINE configs[4] / 8),
                               reference = the oracle's encode above, per-step re-noise tensors (stablediffusion/mod.rs:434-483)
  fullsize_unet1024_f16w.npz   the base UNet::forward of fullsize_unet1024 with every parameter rounded to IEEE f16 first -- what
                               a real record holds (HalfPrecisionSettings, src/bin/sample/main.rs:37): with such weights the
                               engine's f16 weight rounding is exact and the remaining error is activation rounding alone
"""
import os
import sys
import time

import numpy as np
import torch

from . import config as OC, model as OM, pipeline as OP
from .make_golden_fullsize import OUT, base_weights, checksum, seeded, unet1024_inputs


def refiner_weights():
    t0 = time.time()
    cfg = OC.sdxl_refiner_config()
    W = {}
    for p in OC.unet_param_specs(cfg):
        W[p.name] = torch.from_numpy(OC.synth_values(p.name, p.numel, p.scale, p.mean, 0).reshape(p.shape))
    print(f"[golden] SDXL-refiner synthetic weights: {time.time() - t0:.1f} s", flush=True)
    return cfg, W


def refiner1024_inputs(cfg):
    return dict(x=seeded(1, 4, 128, 128, seed=141), t=torch.tensor([150]), ctx=seeded(1, 77, cfg.context_dim, seed=142),
                y=seeded(1, cfg.adm_in_channels, seed=143))


def refine1024_inputs(cfg):
    return dict(latent=seeded(1, 4, 128, 128, seed=151), noise=seeded(1, 4, 128, 128, seed=152),
                ctx=seeded(1, 77, cfg.context_dim, seed=153), uctx=seeded(77, cfg.context_dim, seed=154),
                y=seeded(1, cfg.adm_in_channels, seed=155), uy=seeded(cfg.adm_in_channels, seed=156))


REFINE_START, REFINE_STEPS = 800, 10        # -> t = 199, 99: two refiner iterations


def encode1024_image():
    """a smooth-plus-noise u8 image (platform independent: torch CPU generator)"""
    g = torch.Generator().manual_seed(161)
    yy, xx = torch.meshgrid(torch.arange(1024, dtype=torch.float32), torch.arange(1024, dtype=torch.float32), indexing="ij")
    base = torch.stack([torch.sin(xx / 37.0) * torch.cos(yy / 53.0), torch.sin((xx + yy) / 91.0), torch.cos(xx / 17.0 - yy / 29.0)], -1)
    img = (base * 0.35 + 0.5 + 0.08 * torch.randn(1024, 1024, 3, generator=g)).clamp(0, 1) * 255.0
    return img.to(torch.uint8).numpy()[None]          # [1, 1024, 1024, 3]


def inpaint1024_inputs(cfg):
    return dict(noise=seeded(1, 4, 128, 128, seed=171), ctx=seeded(1, 77, cfg.context_dim, seed=172),
                uctx=seeded(77, cfg.context_dim, seed=173), y=seeded(1, cfg.adm_in_channels, seed=174),
                uy=seeded(cfg.adm_in_channels, seed=175), step_noise=seeded(4, 1, 4, 128, 128, seed=176))


def inpaint_mask():
    m = torch.zeros(1, 4, 128, 128, dtype=torch.bool)
    m[:, :, 0:25, :] = True                  # 200 px / 8: generated content kept in latent rows 0..24
    return m


def run_refiner1024(cfg, W):
    i = refiner1024_inputs(cfg)
    t0 = time.time()
    out = OM.unet_forward(cfg, W, i["x"], i["t"], i["ctx"], i["y"])
    dt = time.time() - t0
    print(f"[golden] refiner UNet::forward 1024^2: {dt:.1f} s, |out|max {out.abs().max():.3f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_refiner1024.npz"), out=out.numpy(),
                        in_checksum=checksum(i["x"], i["ctx"], i["y"]), oracle_seconds=np.array([dt]))


def run_refine1024(cfg, W):
    i = refine1024_inputs(cfg)
    cond = OP.Conditioning(None, i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], (1024, 1024))
    trace = []
    t0 = time.time()
    out = OP.Diffuser(cfg, W, OC.alphas_cumprod()).refine_latent(i["latent"], cond, 7.5, REFINE_START, REFINE_STEPS, i["noise"], trace)
    dt = time.time() - t0
    print(f"[golden] refine_latent 1024^2 ({len(trace)} iterations): {dt:.1f} s, |latent|max {out.abs().max():.3f}", flush=True)
    assert len(trace) == 2
    np.savez_compressed(os.path.join(OUT, "fullsize_refine1024.npz"), traj=np.stack([t.numpy() for t in trace]), latent=out.numpy(),
                        in_checksum=checksum(*i.values()), oracle_seconds=np.array([dt]))


def run_encode1024():
    v = OC.sdxl_vae_config()
    Wv = OM.to_torch(OC.synth_weights(OC.vae_encoder_param_specs(v), 0))
    img = encode1024_image()
    ld = OP.LatentDecoder(v, Wv)
    t0 = time.time()
    lat = ld.image_to_latent(img)
    dt = time.time() - t0
    print(f"[golden] image_to_latent 1024^2: {dt:.1f} s, |latent|max {lat.abs().max():.4f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_encode1024.npz"), latent=lat.numpy(),
                        in_checksum=np.array([float(img.astype(np.float64).sum()), float((img.astype(np.float64) ** 2).sum())]),
                        oracle_seconds=np.array([dt]))
    return lat


def run_inpaint1024(cfg, W, ref_latent):
    i = inpaint1024_inputs(cfg)
    cond = OP.Conditioning(i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], None, (1024, 1024))
    trace = []
    t0 = time.time()
    out = OP.Diffuser(cfg, W, OC.alphas_cumprod()).sample_latent_with_inpainting(
        cond, 7.5, 4, ref_latent, inpaint_mask(), i["noise"], [i["step_noise"][k] for k in range(4)], trace)
    dt = time.time() - t0
    print(f"[golden] inpainting 1024^2 (4 CFG pairs): {dt:.1f} s, |latent|max per step "
          f"{[round(float(t.abs().max()), 2) for t in trace]}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_inpaint1024.npz"), traj=np.stack([t.numpy() for t in trace]), latent=out.numpy(),
                        reference=ref_latent.numpy(), in_checksum=checksum(*i.values()), oracle_seconds=np.array([dt]))


def run_unet1024_f16w(cfg, W):
    i = unet1024_inputs(cfg)
    # (the per-norm eps is a module constant, not a record entry: it stays 1e-5 exactly)
    W16 = {k: (v if k.endswith(".eps") else v.half().float()) for k, v in W.items()}
    t0 = time.time()
    out = OM.unet_forward(cfg, W16, i["x"], i["t"], i["ctx"], i["y"])
    dt = time.time() - t0
    print(f"[golden] UNet::forward 1024^2 with f16-representable weights: {dt:.1f} s, |out|max {out.abs().max():.3f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_unet1024_f16w.npz"), out=out.numpy(),
                        in_checksum=checksum(i["x"], i["ctx"], i["y"]), oracle_seconds=np.array([dt]))


def run_config2_f16w(cfg, W):
    """BASELINE configs[1] (the benchmarked trajectory of make_golden_fullsize.run_config2, same seeded inputs) with every parameter
    rounded to IEEE f16 first -- the weights a real SDXL record holds (HalfPrecisionSettings, src/bin/sample/main.rs:37).  Not part of
    the default set (about 25 minutes of oracle time): `python -m oracle.make_golden_r3 config2_f16w`."""
    from oracle.make_golden_fullsize import config2_inputs
    i = config2_inputs(cfg)
    W16 = {k: (v if k.endswith(".eps") else v.half().float()) for k, v in W.items()}
    cond = OP.Conditioning(i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], None, (1024, 1024))
    trace = []
    t0 = time.time()
    lat = OP.Diffuser(cfg, W16, OC.alphas_cumprod()).sample_latent(cond, 7.5, 30, i["noise"], trace)
    dt = time.time() - t0
    keep = (0, 15, 30)
    print(f"[golden] config 2 with f16-representable weights: 31-step sample_latent {dt:.1f} s, |latent|max {float(lat.abs().max()):.2f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_config2_f16w.npz"), steps=np.array(keep), traj=np.stack([trace[k].numpy() for k in keep]),
                        latent=lat.numpy(), in_checksum=checksum(*i.values()), oracle_seconds=np.array([dt]),
                        oracle_threads=np.array([torch.get_num_threads()]))


def main():
    what = set(sys.argv[1:]) or {"refiner1024", "refine1024", "encode1024", "inpaint1024", "unet1024_f16w"}
    os.makedirs(OUT, exist_ok=True)
    ref_latent = None
    if what & {"encode1024", "inpaint1024"}:
        gp = os.path.join(OUT, "fullsize_encode1024.npz")
        if "encode1024" in what or not os.path.exists(gp):
            ref_latent = run_encode1024()
        else:
            ref_latent = torch.from_numpy(np.load(gp)["latent"])
    if what & {"refiner1024", "refine1024"}:
        cfg, W = refiner_weights()
        if "refiner1024" in what:
            run_refiner1024(cfg, W)
        if "refine1024" in what:
            run_refine1024(cfg, W)
        del W
    if what & {"inpaint1024", "unet1024_f16w", "config2_f16w"}:
        cfg, W = base_weights()
        if "config2_f16w" in what:
            run_config2_f16w(cfg, W)
        if "unet1024_f16w" in what:
            run_unet1024_f16w(cfg, W)
        if "inpaint1024" in what:
            run_inpaint1024(cfg, W, ref_latent)


if __name__ == "__main__":
    main()
