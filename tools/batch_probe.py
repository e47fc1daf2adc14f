"""Throughput with n prompts per sample_latent call ('multi-image batches per GPU'): UNet batch = 2n."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
cfg = pkg.sdxl_base_config()
g = torch.Generator(device="cuda").manual_seed(7)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F16, seed=0)
STEPS = 10
it = pkg.step_count(STEPS)
for n in (1, 2, 4):
    cond = pkg.Conditioning(context_full=r(n, 77, cfg.context_dim), channel_context=r(n, cfg.adm_in_channels),
                            unconditional_context_full=r(77, cfg.context_dim), unconditional_channel_context=r(cfg.adm_in_channels),
                            resolution=(1024, 1024))
    noise = r(n, 4, 128, 128)
    d.sample_latent(cond, 7.5, STEPS, noise)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d.sample_latent(cond, 7.5, STEPS, noise); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"n={n} prompts per call (UNet batch {2 * n}): {best * 1e3 / it:.2f} ms per step = {best * 1e3 / it / n:.2f} ms per image-step", flush=True)
