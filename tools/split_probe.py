"""split-CFG mode (two concurrent batch-1 chains inside UNet::forward) against the batched CFG pair: bit-exact outputs and
step time, full-size SDXL-base at 1024^2."""
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
d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F16, seed=0)
res = {}
OFFS = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]
for mode, off in [(0, 0)] + [(1, o) for o in OFFS]:
    d.diffusion.set_split_cfg(bool(mode), off)
    out = d.sample_latent(cond, 7.5, STEPS, noise)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = d.sample_latent(cond, 7.5, STEPS, noise); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    it = pkg.step_count(STEPS)
    print(f"split_cfg={mode} offset={off}: {best * 1e3:.1f} ms / {it} iterations = {best * 1e3 / it:.2f} ms per step pair", flush=True)
    if mode == 0:
        ref = out.clone()
    else:
        assert torch.equal(ref, out), 'split-CFG output differs from the batched pair'
print('all split outputs bit-exact with the batched pair')
