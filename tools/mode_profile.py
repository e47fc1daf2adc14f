"""Per-class time of one UNet step pair of an engine mode on BASELINE configs[1] (1024^2, CFG pair): the engine's per-launch profile (events around every
launch, SDXL_PROFILE_DUMP) summed by the class tag the UNet driver attaches to each GEMM / attention launch, event overhead removed per launch so that the
classes sum to the graph-replayed step.
    python tools/mode_profile.py [dtype code, default 5 = F32_SPLIT_MIX_F16W] [f16w|fp32] > gpurun_out/r05_mode_profile.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import __graft_entry__ as ge
from precision_frontier import class_profile, seeded

pkg = ge.load_package(); ctx = pkg.Context(0); cfg = pkg.sdxl_base_config()
dt = int(sys.argv[1]) if len(sys.argv) > 1 else pkg.DTYPE_F32_SPLIT_MIX_F16W
f16w = (sys.argv[2] if len(sys.argv) > 2 else "f16w") == "f16w"
i = dict(noise=seeded(1, 4, 128, 128, seed=131), ctx=seeded(1, 77, cfg.context_dim, seed=132), uctx=seeded(77, cfg.context_dim, seed=133),
         y=seeded(1, cfg.adm_in_channels, seed=134), uy=seeded(cfg.adm_in_channels, seed=135))
cond = pkg.Conditioning(context_full=i["ctx"].cuda(), channel_context=i["y"].cuda(), unconditional_context_full=i["uctx"].cuda(),
                        unconditional_channel_context=i["uy"].cuda(), resolution=(1024, 1024))
r = class_profile(pkg, ctx, cfg, dt, cond, i["noise"].cuda(), seed=pkg.SEED_F16_WEIGHTS if f16w else 0)
print(f"dtype {dt}, {'f16-representable' if f16w else 'fp32 synthetic'} weights: step p50 {r['step_ms_p50']} ms, eager sum {r['eager_sum_ms']} ms, "
      f"event overhead {r['event_overhead_us_per_launch']} us per launch")
for c, ms in sorted(r["class_ms"].items(), key=lambda kv: -kv[1]):
    print(f"  {c:12s} {ms:7.3f} ms  {r['launches'][c]:4d} launches")
