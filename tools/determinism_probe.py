import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
for kv in filter(None, os.environ.get("SDXL_DEBUG_SET", "").split(",")):
    k, v = kv.split("="); pkg.debug_set(k, int(v))
cfg = pkg.sdxl_base_config()
def seeded(*s, seed): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
x = seeded(2, 4, 128, 128, seed=50).cuda(); c = seeded(2, 77, cfg.context_dim, seed=51).cuda(); y = seeded(2, cfg.adm_in_channels, seed=52).cuda()
t = torch.tensor([999, 333], dtype=torch.int32).cuda()
u = pkg.UNet(ctx, cfg, pkg.DTYPE_F16, seed=0)
u.set_graph(False)
outs = [u.forward(x, t, c, y).cpu() for _ in range(4)]
print("eager x4 max diffs vs first:", [float((o - outs[0]).abs().max()) for o in outs[1:]])
u.set_graph(True)
outs2 = [u.forward(x, t, c, y).cpu() for _ in range(3)]
print("graph runs vs eager first:", [float((o - outs[0]).abs().max()) for o in outs2])
