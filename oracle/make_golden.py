"""Generates tests/golden/*.npz from the oracle on the reference's deterministic probe inputs
(arb_tensor = sin(arange), /root/reference/src/bin/test/main.rs:51-54,128-162).  TEST INFRASTRUCTURE.

    python -m oracle.make_golden

The reference itself cannot be executed here (no Rust toolchain), so these fixtures pin the ORACLE (they make silent
drift of the restatement visible and give the GPU tests a box-independent target); they are not reference outputs.
"""
import os

import numpy as np
import torch

from . import config as OC, model as OM, pipeline as OP

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    cfg = OC.tiny_config()
    W = OM.to_torch(OC.synth_weights(OC.unet_param_specs(cfg)))
    x = torch.from_numpy(OC.arb_tensor(1, 4, 8, 8))
    ctx = torch.from_numpy(OC.arb_tensor(1, 1, cfg.context_dim))
    y = torch.from_numpy(OC.arb_tensor(1, cfg.adm_in_channels))
    unet_out = OM.unet_forward(cfg, W, x, torch.tensor([1]), ctx, y)
    v = OC.tiny_vae_config()
    Wv = OM.to_torch(OC.synth_weights(OC.vae_decoder_param_specs(v) + OC.vae_encoder_param_specs(v)))
    lat = torch.from_numpy(OC.arb_tensor(1, 4, 4, 4))
    dec = OM.vae_decoder_forward(v, Wv, lat)
    enc = OM.vae_encoder_forward(v, Wv, torch.from_numpy(OC.arb_tensor(1, 3, 16, 16)))
    # 4-step CFG trajectory (BASELINE config-1 flavour: 4 steps, cfg 1.0) on the tiny arch
    cond = OP.Conditioning(ctx[0], None, ctx, None, y[0] * 0.5, None, y, None, (64, 64))
    trace = []
    OP.Diffuser(cfg, W, OC.alphas_cumprod()).sample_latent(cond, 1.0, 4, torch.from_numpy(OC.arb_tensor(1, 4, 8, 8)), trace)
    np.savez_compressed(os.path.join(OUT, "tiny_unet_arb.npz"), unet_out=unet_out.numpy(), vae_dec=dec.numpy(),
                        vae_enc=enc.numpy(), traj=np.stack([t.numpy() for t in trace]))
    print("wrote", os.path.join(OUT, "tiny_unet_arb.npz"))


if __name__ == "__main__":
    main()
