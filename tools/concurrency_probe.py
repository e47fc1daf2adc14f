"""Do two independent UNet chains overlap on one GPU?  One CFG trajectory (batch-2 steps) against two concurrent
cond-only trajectories (batch-1 steps, two Diffusers on two streams): same UNet FLOPs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
cfg = pkg.sdxl_base_config()
g = torch.Generator(device="cuda").manual_seed(7)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
cond = pkg.Conditioning(context_full=r(1, 77, cfg.context_dim), channel_context=r(1, cfg.adm_in_channels),
                        unconditional_context_full=r(77, cfg.context_dim), unconditional_channel_context=r(cfg.adm_in_channels),
                        resolution=(1024, 1024))
noise = r(1, 4, 128, 128)
STEPS = 10


def timed(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F16, seed=0)
d.sample_latent(cond, 7.5, STEPS, noise)
t_cfg = timed(lambda: d.sample_latent(cond, 7.5, STEPS, noise))
iters = pkg.step_count(STEPS)
print(f"CFG pair, batch 2: {t_cfg:.1f} ms for {iters} iterations = {t_cfg / iters:.2f} ms per step pair", flush=True)
pkg.debug_set("no_cfg", 1)
d.sample_latent(cond, 7.5, STEPS, noise)
t_one = timed(lambda: d.sample_latent(cond, 7.5, STEPS, noise))
print(f"cond only, batch 1, one stream: {t_one:.1f} ms = {t_one / iters:.2f} ms per single step (x2 = {2 * t_one / iters:.2f})", flush=True)
d2 = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F16, seed=0)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(sa):
        d.sample_latent(cond, 7.5, STEPS, noise)
    with torch.cuda.stream(sb):
        d2.sample_latent(cond, 7.5, STEPS, noise)
both()
t_two = timed(both)
print(f"two concurrent cond-only chains: {t_two:.1f} ms = {t_two / iters:.2f} ms per step pair  (vs CFG batch-2 {t_cfg / iters:.2f})", flush=True)
