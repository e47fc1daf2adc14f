import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
cfg = pkg.sdxl_base_config()
name = sys.argv[1] if len(sys.argv) > 1 else "f32"
d = pkg.Diffuser(ctx, cfg, {"f32": pkg.DTYPE_F32, "f16": pkg.DTYPE_F16, "f16_f32res": pkg.DTYPE_F16_F32RES, "f32_split": pkg.DTYPE_F32_SPLIT}[name], seed=(pkg.SEED_F16_WEIGHTS if os.environ.get("F16W") == "1" else 0))
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
cond = pkg.Conditioning(context_full=r(1, 77, cfg.context_dim), channel_context=r(1, cfg.adm_in_channels), unconditional_context_full=r(77, cfg.context_dim),
                        unconditional_channel_context=r(cfg.adm_in_channels), resolution=(1024, 1024))
d.enable_step_timing(True)
d.sample_latent(cond, 7.5, 4, r(1, 4, 128, 128))
print(name, "step ms", d.step_times_ms())
print({k: (round(v[0], 2), v[1]) for k, v in d.diffusion.profile(2, 128, 128).items()})
