"""Soak: N eager + N replayed full-size UNet forwards (B = 2, 1024x1024 latents) on identical inputs, every output bit-compared
with the first; and the 4-step 1024x1024 trajectory twice.  usage: python tools/determinism_soak.py [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = pkg.sdxl_base_config()
def seeded(*s, seed): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
x = seeded(2, 4, 128, 128, seed=50).cuda(); c = seeded(2, 77, cfg.context_dim, seed=51).cuda(); y = seeded(2, cfg.adm_in_channels, seed=52).cuda()
t = torch.tensor([999, 333], dtype=torch.int32).cuda()
for dtype, name in ((pkg.DTYPE_F16, "f16"), (pkg.DTYPE_F32_SPLIT, "f32_split")):
    u = pkg.UNet(ctx, cfg, dtype, seed=0)
    n = N if dtype == pkg.DTYPE_F16 else max(4, N // 6)
    u.set_graph(False)
    ref = u.forward(x, t, c, y).clone()
    bad_e = sum(int(not torch.equal(u.forward(x, t, c, y), ref)) for _ in range(n))
    u.set_graph(True)
    bad_g = sum(int(not torch.equal(u.forward(x, t, c, y), ref)) for _ in range(n))
    print(f"{name}: eager {bad_e} / {n} differ, graph {bad_g} / {n} differ", flush=True)
    del u
