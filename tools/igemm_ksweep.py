"""Fixed-cost vs per-k-tile cost of the implicit-GEMM kernels: sweeps K at fixed (M, N) per variant and fits t = a + b*nk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "11"])]
shapes = [(2, 32, 32, 10240, 0), (2, 32, 32, 10240, 1), (2, 32, 32, 1280, 0), (2, 32, 32, 3840, 0), (2, 64, 64, 5120, 0), (2, 128, 128, 512, 0)]
Ks = [64, 320, 640, 1280, 2560, 5120]
for (B, H, W, N, g) in shapes:
    M = B * H * W
    for v in variants:
        pkg.debug_set("igemm_variant", v)
        ts = [pkg.bench_igemm(ctx, B, H, W, K, N, 1, bool(g), 10) * 1e3 for K in Ks]
        nks = [K // 64 for K in Ks]
        xs, ys = nks[2:], ts[2:]
        n = len(xs); sx = sum(xs); sy = sum(ys); sxx = sum(x * x for x in xs); sxy = sum(x * y for x, y in zip(xs, ys))
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
        print(f"M={M} N={N} geglu={g} v{v}: " + " ".join(f"K{K}:{t:.1f}us" for K, t in zip(Ks, ts)) +
              f" | fit fixed={a:.1f}us per-ktile={b*1e3:.0f}ns -> steady {2.0*M*N*64/(b*1e-6)/1e12:.0f} TF/s", flush=True)
