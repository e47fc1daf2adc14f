"""A/B of the unrolled-ring GEMM kernels on the shapes of a step (auto selection, igemm_unrolled 0 vs 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
shapes = [(2, 32, 32, 1280, 10240, 1, True, 60), (2, 32, 32, 1280, 1280, 1, False, 192), (2, 32, 32, 5120, 1280, 1, False, 60),
          (2, 32, 32, 1280, 3840, 1, False, 60), (2, 64, 64, 640, 5120, 1, True, 10), (2, 64, 64, 640, 640, 1, False, 40),
          (2, 32, 32, 1280, 1280, 3, False, 10), (2, 128, 128, 320, 320, 3, False, 7), (2, 64, 64, 640, 640, 3, False, 6)]
tot = [0.0, 0.0]
for (B, H, W, Cin, Cout, ks, g, cnt) in shapes:
    t = []
    for u in (0, 1):
        pkg.debug_set("igemm_unrolled", u)
        t.append(pkg.bench_igemm(ctx, B, H, W, Cin, Cout, ks, g, 10) * 1e3)
        tot[u] += t[-1] * cnt
    print(f"M={B*H*W} Cin={Cin} Cout={Cout} k={ks} geglu={int(g)}: rolled {t[0]:.1f} us  unrolled {t[1]:.1f} us  ({100*(t[1]/t[0]-1):+.1f} %)", flush=True)
print(f"weighted by launches per step: rolled {tot[0]/1e3:.2f} ms  unrolled {tot[1]/1e3:.2f} ms")
