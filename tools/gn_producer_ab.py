"""A/B of sdxl_unet_set_gn_from_producer on the full SDXL-base UNet at 1024^2: batch 2 (the CFG pair) and batch 1 (refiner-style
single forwards, split-CFG chains), hipGraph replay, ms per forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
cfg = pkg.sdxl_base_config()
g = torch.Generator(device="cuda").manual_seed(3)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
for B in (2, 1):
    x, c, y = r(B, 4, 128, 128), r(B, 77, cfg.context_dim), r(B, cfg.adm_in_channels)
    t = torch.full((B,), 500, dtype=torch.int32, device="cuda")
    for on in (True, False, True, False):
        u = pkg.UNet(ctx, cfg, pkg.DTYPE_F16, seed=0)
        u.set_gn_from_producer(on)
        for _ in range(3):
            u.forward(x, t, c, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            u.forward(x, t, c, y)
        torch.cuda.synchronize()
        print(f"B={B} gn_from_producer={on}: {(time.perf_counter() - t0) * 100:.3f} ms per forward", flush=True)
        del u
