"""Per-shape launch times of SDXL_DTYPE_F32_SPLIT_MIX class maps side by side (sdxl_debug_set "mix_classes"), f16-representable weights, 1024^2 CFG pair:
the engine's per-launch profile (hipEvents around every launch of an eager forward, SDXL_PROFILE_DUMP) grouped by (class, M, N, K); raw event times --
every launch carries the same ~3 us of event overhead in both columns.
    python tools/shape_profile.py 63 319 [447 ...] > gpurun_out/r06_shape_profile.txt"""
import csv, os, sys, tempfile, statistics
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import __graft_entry__ as ge
from precision_frontier import label_rows, seeded
pkg = ge.load_package(); ctx = pkg.Context(0); cfg = pkg.sdxl_base_config()
i = dict(noise=seeded(1, 4, 128, 128, seed=131), ctx=seeded(1, 77, cfg.context_dim, seed=132), uctx=seeded(77, cfg.context_dim, seed=133),
         y=seeded(1, cfg.adm_in_channels, seed=134), uy=seeded(cfg.adm_in_channels, seed=135))
cond = pkg.Conditioning(context_full=i["ctx"].cuda(), channel_context=i["y"].cuda(), unconditional_context_full=i["uctx"].cuda(),
                        unconditional_channel_context=i["uy"].cuda(), resolution=(1024, 1024))
masks = [int(a) for a in sys.argv[1:]] or [63, 319]
tables, steps = [], []
for m in masks:
    if m >= 0: pkg.debug_set("mix_classes", m)
    d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F32_SPLIT_MIX if m >= 0 else pkg.DTYPE_F16, seed=pkg.SEED_F16_WEIGHTS)
    d.sample_latent(cond, 7.5, 2, i["noise"].cuda())
    with tempfile.NamedTemporaryFile(suffix=".csv", delete=False) as f: path = f.name
    os.environ["SDXL_PROFILE_DUMP"] = path
    d.diffusion.profile(2, 128, 128)
    del os.environ["SDXL_PROFILE_DUMP"]
    rows = list(csv.DictReader(open(path))); os.unlink(path)
    t = OrderedDict()
    for r, k in zip(rows, label_rows(rows)):
        key = (k, int(r["M"]), int(r["N"]), int(r["K"]), int(r["ksize"]))
        t.setdefault(key, []).append(float(r["ms"]))
    tables.append(t)
    d.enable_step_timing(True); d.sample_latent(cond, 7.5, 8, i["noise"].cuda())
    steps.append(statistics.median(d.step_times_ms()))
    del d
pkg.debug_set("mix_classes", -1)
keys = []
for t in tables:
    for k in t:
        if k not in keys: keys.append(k)
print("mask:", " | ".join(f"{m:>18d}" for m in masks))
print("graph step p50 ms:", " | ".join(f"{s:18.2f}" for s in steps))
print(f"{'class':12s} {'M':>6s} {'N':>6s} {'K':>6s} ks | " + " | ".join("  n    us   tot ms" for _ in masks))
tot = [0.0] * len(masks); nl = [0] * len(masks)
for k in sorted(keys):
    cells = []
    for j, t in enumerate(tables):
        v = t.get(k, [])
        tot[j] += sum(v); nl[j] += len(v)
        cells.append(f"{len(v):3d} {1e3 * sum(v) / max(len(v), 1):6.1f} {sum(v):8.3f}")
    print(f"{k[0]:12s} {k[1]:6d} {k[2]:6d} {k[3]:6d} {k[4]:2d} | " + " | ".join(cells))
print("eager sum ms / launches:", " | ".join(f"{a:10.2f} {b:6d}" for a, b in zip(tot, nl)))
