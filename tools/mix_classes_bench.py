"""SDXL_DTYPE_F32_SPLIT_MIX with further classes on f16 (sdxl_debug_set "mix_classes": 1 self-attention, 2 GEGLU, 4 QKV, 8 FF-out, 16 self-attention out-projection) on the config-2 trajectory:
final-latent error against the oracle fixture and UNet step p50, on the synthetic fp32 weights and on f16-representable ones (the reference's records).
    python tools/mix_classes_bench.py > gpurun_out/r05_mix_classes.txt"""
import os, statistics, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0); cfg = pkg.sdxl_base_config()
def seeded(*s, seed): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
i = dict(noise=seeded(1, 4, 128, 128, seed=131), ctx=seeded(1, 77, cfg.context_dim, seed=132), uctx=seeded(77, cfg.context_dim, seed=133),
         y=seeded(1, cfg.adm_in_channels, seed=134), uy=seeded(cfg.adm_in_channels, seed=135))
def cond(): return pkg.Conditioning(context_full=i["ctx"].cuda(), channel_context=i["y"].cuda(), unconditional_context_full=i["uctx"].cuda(),
                                    unconditional_channel_context=i["uy"].cuda(), resolution=(1024, 1024))
refs = {"fp32 weights": (0, torch.from_numpy(np.load(os.path.join(ROOT, "tests/golden/fullsize_config2.npz"))["latent"])),
        "f16-representable weights": (pkg.SEED_F16_WEIGHTS, torch.from_numpy(np.load(os.path.join(ROOT, "tests/golden/fullsize_config2_f16w.npz"))["latent"]))}
names = {1: "attn", 2: "geglu", 4: "qkv", 8: "ff", 16: "out1", 32: "out2", 64: "xattn"}
print("classes on f16 | weights | final latent max-abs (bound) | UNet step p50 ms | img/s at 31 steps + 39 ms decode + 8 ms")
for mask in [int(a) for a in sys.argv[1:]] or (3, 7, 11, 15, 31):
    pkg.debug_set("mix_classes", mask)
    for wname, (seed, ref) in refs.items():
        d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F32_SPLIT_MIX, seed=seed)
        d.enable_step_timing(True)
        d.sample_latent(cond(), 7.5, 2, i["noise"].cuda())
        lat = d.sample_latent(cond(), 7.5, 30, i["noise"].cuda())
        torch.cuda.synchronize()
        step = statistics.median(d.step_times_ms())
        err = float((lat.cpu() - ref).abs().max()); bound = 1e-3 * max(1.0, float(ref.abs().max()) / 4)
        print(f"{'+'.join(n for b, n in names.items() if mask & b):18s} | {wname:26s} | {err:.4e} ({bound:.4e}, {'inside' if err <= bound else 'OUTSIDE'}) | {step:6.2f} | {1e3 / (31 * step + 47):.3f}", flush=True)
        del d
pkg.debug_set("mix_classes", -1)
