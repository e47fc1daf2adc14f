"""Scratch: one eager, per-launch-profiled UNet CFG step at the bench configuration (writes SDXL_PROFILE_DUMP csv)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
for kv in filter(None, os.environ.get("SDXL_DEBUG_SET", "").split(",")):      # A/B knobs, e.g. wreg_xcd2d=1
    k_, v_ = kv.split("="); pkg.debug_set(k_, int(v_))
ctx = pkg.Context(0)
cfg = pkg.sdxl_base_config()
d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F16, seed=0)
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
cond = pkg.Conditioning(context_full=r(1, 77, 2048), channel_context=r(1, 2816), unconditional_context_full=r(77, 2048),
                        unconditional_channel_context=r(2816), resolution=(1024, 1024))
d.diffusion.set_graph(False)
lat = d.sample_latent(cond, 7.5, 2, r(1, 4, 128, 128))       # n_steps = 2 -> step 500 -> 2 iterations (t = 999, 499): warm-up + plan
torch.cuda.synchronize()
prof = d.diffusion.profile(2, 128, 128)
print({k: (round(v[0], 3), v[1]) for k, v in prof.items()})
