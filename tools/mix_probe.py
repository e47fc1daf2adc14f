"""Mixed mode (SDXL_DTYPE_F32_SPLIT_MIX) against the split engine over sizes: tiny nets at token counts down to 4 (where a wrong f16 / HL16 hand-over shows
as NaN: the round-5 bug of the GEGLU condition) and SDXL-base from 512^2 down to 64^2 images.  `SDXL_NAN_CHECK=1` names the first GEMM with a non-finite output."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge
from oracle import config as OC
from util import to_pkg_cfg, seeded
pkg = ge.load_package(); ctx = pkg.Context(0)
for which in ("tiny", "tiny_refiner"):
    ocfg = OC.tiny_config() if which == "tiny" else OC.tiny_refiner_config()
    for mixc in (3, 1, 2):
        for xs in (1,):
            pkg.debug_set("mix_classes", mixc); pkg.debug_set("attn_xsplit", xs)
            for (B, H, W) in ((1, 8, 8), (2, 8, 12), (4, 8, 12), (2, 16, 16), (1, 8, 12)):
                u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), 4, seed=0)
                u3 = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), 3, seed=0)
                x, c, y = seeded(B, 4, H, W, seed=1), seeded(B, 5, ocfg.context_dim, seed=2), seeded(B, ocfg.adm_in_channels, seed=3)
                t = torch.tensor([500] * B, dtype=torch.int32)
                o = u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
                o3 = u3.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
                fin = bool(torch.isfinite(o).all())
                print(f"{which} mix_classes={mixc} xsplit={xs} B={B} {H}x{W}: finite {fin} rel-to-split {float((o - o3).abs().max() / o3.abs().max()):.3e}", flush=True)
                del u, u3
pkg.debug_set("mix_classes", -1); pkg.debug_set("attn_xsplit", 1)
cfg = pkg.sdxl_base_config()
u, u3 = pkg.UNet(ctx, cfg, 4, seed=0), pkg.UNet(ctx, cfg, 3, seed=0)
for (B, H, W) in ((2, 64, 64), (2, 32, 32), (1, 32, 32), (2, 16, 16), (1, 8, 8)):
    x, c, y = seeded(B, 4, H, W, seed=1), seeded(B, 77, cfg.context_dim, seed=2), seeded(B, cfg.adm_in_channels, seed=3)
    t = torch.tensor([500] * B, dtype=torch.int32)
    o = u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu(); o3 = u3.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
    print(f"SDXL-base mix B={B} latent {H}x{W}: finite {bool(torch.isfinite(o).all())} rel-to-split {float((o - o3).abs().max() / o3.abs().max()):.3e}", flush=True)
