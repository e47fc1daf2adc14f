"""Scratch: N VAE decodes of a 128x128 latent (1024^2 image) at the bench's decode precision, for rocprofv3 --kernel-trace --stats (per-kernel times of LatentDecoder::latent_to_image).
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vd -o v -- python tools/profile_decode.py [f32_split|f16|f32] [fp32|f16 weights]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f32_split"
dt = {"f32_split": pkg.DTYPE_F32_SPLIT, "f16": pkg.DTYPE_F16, "f32": pkg.DTYPE_F32}[prec]
seed = pkg.SEED_F16_WEIGHTS if (len(sys.argv) > 2 and sys.argv[2] == "f16") else 0
dec = pkg.LatentDecoder(ctx, None, dt, seed=seed)
lat = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(6):
    img = dec.latent_to_image(lat)
torch.cuda.synchronize()
print("ok", tuple(img.buffer.shape) if hasattr(img, "buffer") else type(img))
