"""Generates tests/golden/fullsize_*.npz: the oracle at the BASELINE configurations.  TEST INFRASTRUCTURE.

    python -m oracle.make_golden_fullsize [config1] [unet1024] [decode1024]      (default: all three)

The reference (Rust + burn + libtorch) cannot be executed here, so -- like oracle/make_golden.py -- these fixtures are
outputs of the ORACLE (the fp32 torch-CPU restatement of the reference graph), produced in the CPU build container on the
full SDXL-base / SDXL-VAE architectures with the seeded synthetic weights (seed 0) that the HIP fill kernel reproduces bit
for bit on the device.  They give the -m gpu tests an oracle target at the sizes bench.py times, without paying the
oracle's 1-2 minutes (and 10 GB of fp32 weights) on the GPU box:

  fullsize_config1.npz   BASELINE configs[0]: SDXL-base, 512x512 (latent 64x64), n_steps=4 (t = 999, 749, 499, 249),
                         CFG 1.0 (both branches evaluated, stablediffusion/mod.rs:523-540), per-step latents + final
                         latent + decode_latent + u8 image (src/bin/sample/main.rs:239-278)
  fullsize_config2.npz   BASELINE configs[1] (the configuration bench.py times): 1024x1024, n_steps=30 -> 31 iterations,
                         CFG 7.5; final latent + 9 of the 31 per-step latents (~25 minutes of oracle time on 8 cores;
                         not part of the default set: `python -m oracle.make_golden_fullsize config2`)
  fullsize_unet1024.npz  one UNet::forward at 1024x1024 (latent 128x128), unet/mod.rs:450-492
  fullsize_decode1024.npz one LatentDecoder::latent_to_image at 1024x1024: stride-5 subsample of the fp32 image and of
                         the u8 image (the full fp32 image is 12 MB) + four dense 64x64 crops

Inputs come from torch's CPU generator with the seeds below (platform independent); a checksum of every input is stored so
the test notices a generator change instead of reporting a parity failure.
"""
import os
import sys
import time

import numpy as np
import torch

from . import config as OC, model as OM, pipeline as OP

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SUB = 5          # image subsample stride: co-prime with the 8x upsampling, so every phase of the up-convs is hit
CROPS = ((0, 0), (0, 960), (480, 480), (960, 0))   # top-left corners of the dense 64x64 crops (1024^2 image)


def seeded(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def checksum(*tensors) -> np.ndarray:
    return np.array([float(t.double().sum()) for t in tensors] + [float(t.double().abs().sum()) for t in tensors])


def config1_inputs(cfg):
    """inputs of the config-1 run (seeds shared with tests/test_gpu_baseline_parity.py and bench.py --config 1)"""
    return dict(noise=seeded(1, 4, 64, 64, seed=101), ctx=seeded(1, 77, cfg.context_dim, seed=102),
                uctx=seeded(77, cfg.context_dim, seed=103), y=seeded(1, cfg.adm_in_channels, seed=104),
                uy=seeded(cfg.adm_in_channels, seed=105))


def config2_inputs(cfg):
    """BASELINE configs[1] (the benchmarked one): 1024x1024, n_steps=30 -> 31 iterations, CFG 7.5 (seeds shared with the tests)"""
    return dict(noise=seeded(1, 4, 128, 128, seed=131), ctx=seeded(1, 77, cfg.context_dim, seed=132),
                uctx=seeded(77, cfg.context_dim, seed=133), y=seeded(1, cfg.adm_in_channels, seed=134),
                uy=seeded(cfg.adm_in_channels, seed=135))


CONFIG2_KEEP = (0, 1, 2, 5, 10, 15, 20, 25, 30)   # trajectory steps stored in the fixture (each 256 KB)


def run_config2(cfg, W):
    i = config2_inputs(cfg)
    cond = OP.Conditioning(i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], None, (1024, 1024))
    trace = []
    t0 = time.time()
    lat = OP.Diffuser(cfg, W, OC.alphas_cumprod()).sample_latent(cond, 7.5, 30, i["noise"], trace)
    dt = time.time() - t0
    amax = np.array([float(t.abs().max()) for t in trace])
    print(f"[golden] config 2: 31-step sample_latent {dt:.1f} s, |latent|max per step {np.array2string(amax, precision=2)}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_config2.npz"), steps=np.array(CONFIG2_KEEP),
                        traj=np.stack([trace[k].numpy() for k in CONFIG2_KEEP]), latent=lat.numpy(), absmax=amax,
                        in_checksum=checksum(*i.values()), oracle_seconds=np.array([dt]),
                        oracle_threads=np.array([torch.get_num_threads()]))


def unet1024_inputs(cfg):
    return dict(x=seeded(1, 4, 128, 128, seed=111), t=torch.tensor([500]), ctx=seeded(1, 77, cfg.context_dim, seed=112),
                y=seeded(1, cfg.adm_in_channels, seed=113))


def decode1024_inputs():
    return dict(latent=seeded(1, 4, 128, 128, seed=121))


def base_weights():
    t0 = time.time()
    cfg = OC.sdxl_base_config()
    W = {}
    for p in OC.unet_param_specs(cfg):      # one tensor at a time: the fp32 set is 10.3 GB
        W[p.name] = torch.from_numpy(OC.synth_values(p.name, p.numel, p.scale, p.mean, 0).reshape(p.shape))
    print(f"[golden] SDXL-base synthetic weights: {time.time() - t0:.1f} s", flush=True)
    return cfg, W


def vae_weights():
    v = OC.sdxl_vae_config()
    return v, OM.to_torch(OC.synth_weights(OC.vae_decoder_param_specs(v), 0))


def run_config1(cfg, W, v, Wv):
    i = config1_inputs(cfg)
    cond = OP.Conditioning(i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], None, (512, 512))
    trace = []
    t0 = time.time()
    lat = OP.Diffuser(cfg, W, OC.alphas_cumprod()).sample_latent(cond, 1.0, 4, i["noise"], trace)
    t1 = time.time()
    ld = OP.LatentDecoder(v, Wv)
    img = ld.decode_latent(lat)
    u8 = ld.latent_to_image(lat)
    t2 = time.time()
    print(f"[golden] config 1: sample_latent {t1 - t0:.1f} s, decode {t2 - t1:.1f} s, |latent|max {lat.abs().max():.3f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_config1.npz"), traj=np.stack([t.numpy() for t in trace]),
                        latent=lat.numpy(), image_sub=img[:, :, ::SUB, ::SUB].numpy().copy(), u8=u8,
                        in_checksum=checksum(*i.values()), oracle_seconds=np.array([t1 - t0, t2 - t1]),
                        oracle_threads=np.array([torch.get_num_threads()]))


def run_unet1024(cfg, W):
    i = unet1024_inputs(cfg)
    t0 = time.time()
    out = OM.unet_forward(cfg, W, i["x"], i["t"], i["ctx"], i["y"])
    print(f"[golden] UNet::forward 1024^2: {time.time() - t0:.1f} s, |out|max {out.abs().max():.3f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_unet1024.npz"), out=out.numpy(),
                        in_checksum=checksum(i["x"], i["ctx"], i["y"]), oracle_seconds=np.array([time.time() - t0]))


def run_decode1024(v, Wv):
    i = decode1024_inputs()
    ld = OP.LatentDecoder(v, Wv)
    t0 = time.time()
    img = ld.decode_latent(i["latent"])
    u8 = ld.latent_to_image(i["latent"])
    print(f"[golden] decode 1024^2 (x2): {time.time() - t0:.1f} s, |img|max {img.abs().max():.3f}", flush=True)
    crops = np.stack([img[0, :, r:r + 64, c:c + 64].numpy() for r, c in CROPS])
    crops8 = np.stack([u8[0, r:r + 64, c:c + 64] for r, c in CROPS])
    np.savez_compressed(os.path.join(OUT, "fullsize_decode1024.npz"), image_sub=img[:, :, ::SUB, ::SUB].numpy().copy(),
                        u8_sub=u8[:, ::SUB, ::SUB].copy(), crops=crops, crops_u8=crops8, in_checksum=checksum(i["latent"]),
                        image_stats=np.array([float(img.mean()), float(img.std()), float(img.min()), float(img.max())]),
                        u8_hist=np.bincount(u8.reshape(-1), minlength=256))


def main():
    what = set(sys.argv[1:]) or {"config1", "unet1024", "decode1024"}
    os.makedirs(OUT, exist_ok=True)
    v, Wv = vae_weights()
    if "decode1024" in what:
        run_decode1024(v, Wv)
    if what & {"config1", "unet1024", "config2"}:
        cfg, W = base_weights()
        if "unet1024" in what:
            run_unet1024(cfg, W)
        if "config1" in what:
            run_config1(cfg, W, v, Wv)
        if "config2" in what:
            run_config2(cfg, W)


if __name__ == "__main__":
    main()
