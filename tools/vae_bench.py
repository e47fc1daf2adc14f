"""VAE legs alone at 1024x1024: latent_to_image per precision (ms, error vs the committed oracle fixture, u8 diffs) + a per-launch profile.

    python tools/vae_bench.py [dtype ...]        dtypes: f32 f32_split f16
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
ctx = pkg.Context(0)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize_decode1024.npz"))
latent = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(121)).cuda()
names = sys.argv[1:] or ["f32", "f32_split", "f16"]
dts = {"f32": pkg.DTYPE_F32, "f32_split": pkg.DTYPE_F32_SPLIT, "f16": pkg.DTYPE_F16}
for name in names:
    ld = pkg.LatentDecoder(ctx, None, dts[name], seed=0)
    img = ld.decode_latent(latent)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        u8 = ld.latent_to_image(latent)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    sub = img.cpu()[:, :, ::5, ::5]
    ref = torch.from_numpy(g["image_sub"])
    err = float((sub - ref).abs().max())
    d8 = np.abs(u8.buffer.cpu().numpy()[:, ::5, ::5].astype(np.int32) - g["u8_sub"].astype(np.int32))
    print(f"latent_to_image 1024^2 {name}: {ms:.2f} ms  ({10.47 / ms:.1f} TFLOP/s algorithmic); image max-abs err vs oracle {err:.3e} (|ref| {float(ref.abs().max()):.2f}); "
          f"u8 max diff {d8.max()} in {float((d8 > 0).mean()):.2e} of bytes", flush=True)
    del ld
