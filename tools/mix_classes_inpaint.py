"""SDXL_DTYPE_F32_SPLIT_MIX class maps (sdxl_debug_set "mix_classes") on the 4-step INPAINTING fixture with f16-representable weights -- the
configuration whose 250-step jumps amplify a forward's error most: per-step error as a fraction of the scaled bound, for every subset of
{GEGLU, QKV, FF-out, out1, out2} on top of the f16 self-attention.
    python tools/mix_classes_inpaint.py [masks...] > gpurun_out/r06_mix_classes_inpaint.txt"""
import os, statistics, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0); cfg = pkg.sdxl_base_config()
GOLD = os.path.join(ROOT, "tests", "golden")
def seeded(*s, seed): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
i = dict(noise=seeded(1, 4, 128, 128, seed=171), ctx=seeded(1, 77, cfg.context_dim, seed=172), uctx=seeded(77, cfg.context_dim, seed=173),
         y=seeded(1, cfg.adm_in_channels, seed=174), uy=seeded(cfg.adm_in_channels, seed=175), step_noise=seeded(4, 1, 4, 128, 128, seed=176))
def cond(): return pkg.Conditioning(context_full=i["ctx"].cuda(), channel_context=i["y"].cuda(), unconditional_context_full=i["uctx"].cuda(),
                                    unconditional_channel_context=i["uy"].cuda(), resolution=(1024, 1024))
g = np.load(os.path.join(GOLD, "fullsize_inpaint1024_f16w.npz"))
reference = torch.from_numpy(np.load(os.path.join(GOLD, "fullsize_inpaint1024.npz"))["reference"])
mask = torch.zeros(1, 4, 128, 128, dtype=torch.bool); mask[:, :, 0:25, :] = True
ref_traj = torch.from_numpy(g["traj"]).clone()
alphas = pkg.default_alphas_cumprod(); ts = [999, 749, 499, 249]
for k in range(3):       # (the engine's trace holds the blend for the next iteration: tests/test_gpu_baseline_parity.py)
    a_n = float(alphas[ts[k + 1]])
    ref_traj[k] = torch.where(mask, ref_traj[k], reference * (a_n ** 0.5) + i["step_noise"][k + 1] * ((1.0 - a_n) ** 0.5))
bounds = [1e-3 * max(1.0, float(ref_traj[k].abs().max()) / 4) for k in range(4)]
names = {1: "attn", 2: "geglu", 4: "qkv", 8: "ff", 16: "out1", 32: "out2", 64: "xattn", 128: "q2"}
masks = [int(a) for a in sys.argv[1:]] or [1 | (m << 1) for m in range(32)]
print("classes on f16 | per-step error / bound (4 steps) | worst | UNet step p50 ms")
for m in masks:
    pkg.debug_set("mix_classes", m)
    d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F32_SPLIT_MIX, seed=pkg.SEED_F16_WEIGHTS)
    d.enable_step_timing(True)
    trace = torch.zeros(4, 1, 4, 128, 128, device="cuda")
    d.set_trace(trace)
    d.sample_latent_with_inpainting(cond(), 7.5, 4, reference.cuda(), mask.cuda(), i["noise"].cuda(), i["step_noise"].cuda())
    torch.cuda.synchronize()
    d.set_trace(None)
    r = [float((trace[k].cpu() - ref_traj[k]).abs().max()) / bounds[k] for k in range(4)]
    step = statistics.median(d.step_times_ms())
    print(f"{m:3d} {'+'.join(n for b, n in names.items() if m & b):28s} | " + " ".join(f"{x:5.2f}" for x in r) + f" | {max(r):5.2f} | {step:6.2f}", flush=True)
    del d
pkg.debug_set("mix_classes", -1)
