"""Per-shape launch times of ONE refiner UNet forward at 1024^2, batch 1 (what Diffuser::refine_latent runs: 10 of them in BASELINE configs[3]), f16 engine:
the engine's per-launch profile (hipEvents around every launch of an eager forward) grouped by (class, M, N, K); raw event times.
    python tools/refiner_shape_profile.py > gpurun_out/r06_refiner_shape_profile.txt"""
import csv, os, sys, tempfile
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import __graft_entry__ as ge
from precision_frontier import label_rows, seeded
pkg = ge.load_package(); ctx = pkg.Context(0); cfg = pkg.sdxl_refiner_config()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
u = pkg.UNet(ctx, cfg, pkg.DTYPE_F16, seed=0)
x = seeded(B, 4, 128, 128, seed=1).cuda(); c = seeded(B, 77, cfg.context_dim, seed=2).cuda(); y = seeded(B, cfg.adm_in_channels, seed=3).cuda()
t = torch.full((B,), 500, dtype=torch.int32).cuda()
for _ in range(3): u.forward(x, t, c, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): u.forward(x, t, c, y)
e1.record(); torch.cuda.synchronize()
print(f"refiner forward, batch {B}, graph replay: {e0.elapsed_time(e1) / 5:.3f} ms")
with tempfile.NamedTemporaryFile(suffix=".csv", delete=False) as f: path = f.name
os.environ["SDXL_PROFILE_DUMP"] = path
u.profile(B, 128, 128)
del os.environ["SDXL_PROFILE_DUMP"]
rows = list(csv.DictReader(open(path))); os.unlink(path)
tab = OrderedDict()
for r, k in zip(rows, label_rows(rows)):
    tab.setdefault((k, int(r["M"]), int(r["N"]), int(r["K"]), int(r["ksize"])), []).append(float(r["ms"]))
print(f"{'class':12s} {'M':>6s} {'N':>6s} {'K':>6s} ks |   n     us   tot ms   TFLOP/s")
tot = 0.0
for k in sorted(tab):
    v = tab[k]; tot += sum(v)
    fl = 2.0 * k[1] * k[2] * k[3] if k[0] not in ("norm", "other", "attn") else 0.0
    print(f"{k[0]:12s} {k[1]:6d} {k[2]:6d} {k[3]:6d} {k[4]:2d} | {len(v):3d} {1e3 * sum(v) / len(v):6.1f} {sum(v):8.3f} {fl / (1e9 * sum(v) / len(v)) if fl else 0:8.1f}")
print(f"eager sum {tot:.2f} ms, {sum(len(v) for v in tab.values())} launches")
