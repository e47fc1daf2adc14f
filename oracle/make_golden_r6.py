"""Round-6 fixtures.  TEST INFRASTRUCTURE (outputs of the ORACLE: fp32 torch-CPU restatement of the reference graph; parity unpinned --
the reference itself cannot be built here).

    python -m oracle.make_golden_r6 [config5_f16w] [config5] [refiner1024_f16w] [refine1024_f16w] [decode1024_f16w]

  fullsize_config5_f16w.npz      BASELINE configs[4] AT ITS OWN STEP COUNT: Diffuser::sample_latent_with_inpainting at 1024x1024, n_steps = 100
                                 (t = 999, 989, ... 9: 100 CFG-7.5 pairs = 200 UNet forwards), mask = latent rows 0..24 generated (the 200 px crop),
                                 reference latent = the oracle's encode (fullsize_inpaint1024.npz), per-step re-noise tensors seeded(100, ..., seed=186),
                                 on f16-representable UNet weights (what the reference's records hold, src/bin/sample/main.rs:37).  Every 10th latent
                                 + the final one are kept.  The 4-step fixtures of rounds 3 / 5 (fullsize_inpaint1024*.npz) take 250-step jumps, which
                                 amplify one forward's error ~10x more than the 10-step jumps of the configuration BASELINE names; this is that configuration.
  fullsize_config5.npz           the same on the synthetic fp32 weights
  fullsize_decode1024_f16w.npz   LatentDecoder::latent_to_image at 1024^2 (latent of fullsize_decode1024.npz) with the VAE decoder's parameters rounded to f16 -- the
                                 reference's decoder record is HalfPrecisionSettings too (src/bin/sample/main.rs:37-51); with such weights the split-operand VAE runs
                                 two MFMAs per product instead of three (bench.py --weights f16)
  fullsize_refiner1024_f16w.npz  one refiner UNet::forward at 1024^2 (inputs of fullsize_refiner1024.npz) on f16-representable weights
  fullsize_refine1024_f16w.npz   Diffuser::refine_latent, 2 refiner iterations (inputs of fullsize_refine1024.npz) on f16-representable weights
"""
import os
import sys
import time

import numpy as np
import torch

from . import config as OC, model as OM, pipeline as OP
from .make_golden_fullsize import OUT, base_weights, checksum, seeded
from .make_golden_r3 import (REFINE_START, REFINE_STEPS, inpaint_mask, refine1024_inputs, refiner1024_inputs, refiner_weights)

CONFIG5_STEPS = 100
CONFIG5_KEEP = tuple(range(9, 100, 10))          # iterations whose latent is kept (the last one is the result)


def f16w(W):
    # (the per-norm eps is a module constant, not a record entry: it stays 1e-5 exactly)
    return {k: (v if k.endswith(".eps") else v.half().float()) for k, v in W.items()}


def config5_inputs(cfg):
    return dict(noise=seeded(1, 4, 128, 128, seed=181), ctx=seeded(1, 77, cfg.context_dim, seed=182), uctx=seeded(77, cfg.context_dim, seed=183),
                y=seeded(1, cfg.adm_in_channels, seed=184), uy=seeded(cfg.adm_in_channels, seed=185),
                step_noise=seeded(CONFIG5_STEPS, 1, 4, 128, 128, seed=186))


def run_config5(cfg, W, name):
    i = config5_inputs(cfg)
    ref_latent = torch.from_numpy(np.load(os.path.join(OUT, "fullsize_inpaint1024.npz"))["reference"])
    cond = OP.Conditioning(i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], None, (1024, 1024))

    class Trace(list):            # progress line per kept step (the run takes over an hour)
        def append(self, t):
            super().append(t)
            if (len(self) - 1) in CONFIG5_KEEP:
                print(f"[golden r6] {name}: iteration {len(self)} / {CONFIG5_STEPS}, {time.time() - t0:.0f} s, |latent|max {float(t.abs().max()):.2f}", flush=True)
    trace = Trace()
    t0 = time.time()
    out = OP.Diffuser(cfg, W, OC.alphas_cumprod()).sample_latent_with_inpainting(
        cond, 7.5, CONFIG5_STEPS, ref_latent, inpaint_mask(), i["noise"], i["step_noise"], trace)
    dt = time.time() - t0
    assert len(trace) == CONFIG5_STEPS
    np.savez_compressed(os.path.join(OUT, name + ".npz"), steps=np.array(CONFIG5_KEEP), traj=np.stack([trace[k].numpy() for k in CONFIG5_KEEP]),
                        latent=out.numpy(), in_checksum=checksum(*i.values()), oracle_seconds=np.array([dt]),
                        oracle_threads=np.array([torch.get_num_threads()]))
    print(f"[golden r6] {name}: {dt:.0f} s, |latent|max {float(out.abs().max()):.2f}", flush=True)


def run_refiner1024_f16w(cfg, W16):
    i = refiner1024_inputs(cfg)
    t0 = time.time()
    out = OM.unet_forward(cfg, W16, i["x"], i["t"], i["ctx"], i["y"])
    dt = time.time() - t0
    print(f"[golden r6] refiner UNet::forward 1024^2, f16-representable weights: {dt:.1f} s, |out|max {out.abs().max():.3f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_refiner1024_f16w.npz"), out=out.numpy(), in_checksum=checksum(i["x"], i["ctx"], i["y"]),
                        oracle_seconds=np.array([dt]))


def run_refine1024_f16w(cfg, W16):
    i = refine1024_inputs(cfg)
    cond = OP.Conditioning(None, i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], (1024, 1024))
    trace = []
    t0 = time.time()
    out = OP.Diffuser(cfg, W16, OC.alphas_cumprod()).refine_latent(i["latent"], cond, 7.5, REFINE_START, REFINE_STEPS, i["noise"], trace)
    dt = time.time() - t0
    assert len(trace) == 2
    print(f"[golden r6] refine_latent 1024^2, f16-representable weights: {dt:.1f} s, |latent|max {out.abs().max():.3f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_refine1024_f16w.npz"), traj=np.stack([t.numpy() for t in trace]), latent=out.numpy(),
                        in_checksum=checksum(*i.values()), oracle_seconds=np.array([dt]))


def run_decode1024_f16w():
    from .make_golden_fullsize import SUB, decode1024_inputs, vae_weights
    v, Wv = vae_weights()
    i = decode1024_inputs()
    ld = OP.LatentDecoder(v, f16w(Wv))
    t0 = time.time()
    img = ld.decode_latent(i["latent"])
    u8 = ld.latent_to_image(i["latent"])
    print(f"[golden r6] decode 1024^2 (x2), f16-representable VAE weights: {time.time() - t0:.1f} s, |img|max {img.abs().max():.3f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_decode1024_f16w.npz"), image_sub=img[:, :, ::SUB, ::SUB].numpy().copy(), u8_sub=u8[:, ::SUB, ::SUB].copy(),
                        in_checksum=checksum(i["latent"]))


def main():
    what = set(sys.argv[1:]) or {"refiner1024_f16w", "refine1024_f16w"}
    if "--threads" in sys.argv:
        torch.set_num_threads(int(sys.argv[sys.argv.index("--threads") + 1]))
    os.makedirs(OUT, exist_ok=True)
    if "decode1024_f16w" in what:
        run_decode1024_f16w()
    if what & {"refiner1024_f16w", "refine1024_f16w"}:
        cfg, W = refiner_weights()
        W16 = f16w(W)
        del W
        if "refiner1024_f16w" in what:
            run_refiner1024_f16w(cfg, W16)
        if "refine1024_f16w" in what:
            run_refine1024_f16w(cfg, W16)
        del W16
    if what & {"config5_f16w", "config5"}:
        cfg, W = base_weights()
        if "config5_f16w" in what:
            run_config5(cfg, f16w(W), "fullsize_config5_f16w")
        if "config5" in what:
            run_config5(cfg, W, "fullsize_config5")


if __name__ == "__main__":
    main()
