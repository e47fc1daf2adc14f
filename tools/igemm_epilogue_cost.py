"""What the folded-LayerNorm prologue / epilogue and the row-statistics epilogue cost on the step's linear shapes (cold weights):
the same GEMM with and without ln_stat (consumer side) / stat_out (producer side).  bench_igemm flag bits: 1 GEGLU, 2 ln_in, 4 stat_out, 8 cold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
S = [("lin32 qkv", 2, 32, 32, 1280, 3840, 0), ("lin32 out/q", 2, 32, 32, 1280, 1280, 0), ("lin32 geglu", 2, 32, 32, 1280, 10240, 1),
     ("lin32 ff", 2, 32, 32, 5120, 1280, 0), ("lin64 qkv", 2, 64, 64, 640, 1920, 0), ("lin64 out/q", 2, 64, 64, 640, 640, 0)]
print("shape          plain us | + ln_stat (LayerNorm folded, consumer) | + stat_out (row statistics, producer)")
for name, B, H, W, K, N, g in S:
    t = {}
    for rep in range(2):
        for tag, fl in (("plain", 0), ("ln", 2), ("st", 4)):
            if tag == "st" and g:
                continue
            ms = pkg.bench_igemm(ctx, B, H, W, K, N, 1, g | fl | 8, 10)
            t[tag] = min(t.get(tag, 1e9), ms * 1e3)
    print(f"{name:14s} {t['plain']:7.1f}  | {t['ln']:7.1f} (+{t['ln'] - t['plain']:.1f}) | " + (f"{t['st']:7.1f} (+{t['st'] - t['plain']:.1f})" if "st" in t else "   -"), flush=True)
