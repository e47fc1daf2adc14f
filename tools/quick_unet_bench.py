"""Scratch timing of the full-size UNet forward / one trajectory (not the contract bench; see bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
dt = {"f16": pkg.DTYPE_F16, "f32": pkg.DTYPE_F32, "mixed": pkg.DTYPE_F16_F32RES}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = pkg.Context(0)
cfg = pkg.sdxl_base_config()
t0 = time.time()
d = pkg.Diffuser(ctx, cfg, dt, seed=0)
torch.cuda.synchronize()
print(f"create: {time.time()-t0:.1f}s, weights {d.diffusion.weight_arena()[1]/1e9:.2f} GB", flush=True)
g = torch.Generator(device="cuda").manual_seed(0)
n = 1
cond = pkg.Conditioning(context_full=torch.randn(n, 77, 2048, device="cuda", generator=g),
                        channel_context=torch.randn(n, 2816, device="cuda", generator=g),
                        unconditional_context_full=torch.randn(77, 2048, device="cuda", generator=g),
                        unconditional_channel_context=torch.randn(2816, device="cuda", generator=g),
                        resolution=(res, res))
noise = torch.randn(n, 4, res // 8, res // 8, device="cuda", generator=g)
d.enable_step_timing(True)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    lat = d.sample_latent(cond, 7.5, steps, noise)
    torch.cuda.synchronize(); dtm = time.time() - t0
    ms = d.step_times_ms()
    print(f"run {it}: {dtm*1e3:.1f} ms total, steps(ms): {[round(x,2) for x in ms]}  finite={bool(torch.isfinite(lat).all())} absmax={lat.abs().max().item():.3f}", flush=True)
ld = pkg.LatentDecoder(ctx, None, dt, seed=0)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    img = ld.latent_to_image(lat)
    torch.cuda.synchronize()
    print(f"vae decode {it}: {(time.time()-t0)*1e3:.1f} ms, img mean {img.buffer.float().mean().item():.2f}", flush=True)
