import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
for kv in filter(None, os.environ.get("SDXL_DEBUG_SET", "").split(",")):
    k, v = kv.split("="); pkg.debug_set(k, int(v))
def seeded(*s, seed): return torch.randn(*s, generator=torch.Generator().manual_seed(seed)).cuda()
def rep(name, fn, n=4):
    outs = [fn() for _ in range(n)]
    outs = [o[0] if isinstance(o, tuple) else o for o in outs]
    print(f"{name}: max diffs vs first {[float((o - outs[0]).abs().max()) for o in outs[1:]]}", flush=True)
# linears of the 32^2 level
for (M, K, N, geglu) in ((2048, 1280, 1280, False), (2048, 5120, 1280, False), (2048, 1280, 3840, False), (2048, 1280, 10240, True), (8192, 640, 640, False), (8192, 2560, 640, False)):
    x, w, b = seeded(M, K, seed=1), seeded(K, N, seed=2) / math.sqrt(K), seeded(N, seed=3)
    rep(f"linear M{M} K{K} N{N} geglu{int(geglu)}", lambda: pkg.linear(ctx, x, w, b, geglu, pkg.DTYPE_F16))
    g, be = 1 + 0.1 * seeded(K, seed=4), 0.1 * seeded(K, seed=5)
    rep(f"layer_norm_linear M{M} K{K} N{N} geglu{int(geglu)}", lambda: pkg.layer_norm_linear(ctx, x, g, be, w, b, 1e-5, geglu, pkg.DTYPE_F16))
for (B, Cin, H, W, Cout, k, up) in ((2, 1280, 32, 32, 1280, 3, False), (2, 640, 64, 64, 640, 3, False), (2, 320, 128, 128, 320, 3, False), (2, 1280, 32, 32, 1280, 3, True), (2, 960, 128, 128, 320, 1, False)):
    x, w, b = seeded(B, Cin, H, W, seed=6), seeded(Cout, Cin, k, k, seed=7) / math.sqrt(Cin * k * k), seeded(Cout, seed=8)
    rep(f"conv2d {B}x{Cin}x{H}x{W}->{Cout} k{k} up{int(up)}", lambda: pkg.conv2d(ctx, x, w, b, 1, k // 2, up, pkg.DTYPE_F16))
    if k == 3 and not up:
        g, be = 1 + 0.1 * seeded(Cout, seed=9), 0.1 * seeded(Cout, seed=10)
        for fused in (True, False):
            rep(f"conv2d_group_norm fused={fused} {Cin}@{H}", lambda: pkg.conv2d_group_norm(ctx, x, w, b, g, be, 1e-5, 32, True, None, fused))
for (B, Cin, H, W, Cout) in ((2, 1280, 32, 32, 1280), (2, 640, 64, 64, 640), (2, 320, 128, 128, 320)):
    x, w, b = seeded(B, Cin, H, W, seed=6), seeded(Cout, Cin, 3, 3, seed=7) / math.sqrt(Cin * 9), seeded(Cout, seed=8)
    g, be, r = 1 + 0.1 * seeded(Cout, seed=9), 0.1 * seeded(Cout, seed=10), seeded(B, Cout, H, W, seed=11)
    for fused in (True, False):
        rep(f"conv2d_group_norm +residual fused={fused} {Cin}@{H}", lambda: pkg.conv2d_group_norm(ctx, x, w, b, g, be, 1e-5, 32, True, r, fused))
for (B, Nq, C) in ((2, 1024, 1280), (2, 4096, 640)):
    x = seeded(B, Nq, C, seed=12); g, be = 1 + 0.1 * seeded(C, seed=13), 0.1 * seeded(C, seed=14)
    wq = seeded(C, C, seed=15) / math.sqrt(C); k, v = seeded(B, 77, C, seed=16), seeded(B, 77, C, seed=17)
    for fused in (True, False):
        rep(f"ln_query_cross_attention fused={fused} Nq{Nq} C{C}", lambda: pkg.ln_query_cross_attention(ctx, x, g, be, wq, k, v, 1e-5, fused))
for (B, N, C, H) in ((2, 1024, 1280, 20), (2, 4096, 640, 10)):
    q, k, v = seeded(B, N, C, seed=18), seeded(B, N, C, seed=19), seeded(B, N, C, seed=20)
    rep(f"qkv_attention N{N} C{C}", lambda: pkg.qkv_attention(ctx, q, k, v, None, H, pkg.DTYPE_F16))
