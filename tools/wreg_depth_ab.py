"""prefetch depth of the weights-in-registers GEMM: L = 4 (production, variant 60) vs 3 (68) vs 2 (69); cold / warm weights, us"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDXL_MEASURE_LIB"] = "1"
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
S = [("out-proj K1280", 2, 32, 32, 1280, 1280), ("ff-out K5120", 2, 32, 32, 5120, 1280), ("skip K2560", 2, 32, 32, 2560, 1280), ("K640", 2, 32, 32, 640, 1280)]
for cold in (1, 0):
    print(f"--- {'cold' if cold else 'warm'} weights")
    for name, B, H, W, Cin, Cout in S:
        row = f"{name:16s}"
        for n, v in (("L4", 60), ("L3", 68), ("L2", 69), ("L4", 60), ("L3", 68), ("L2", 69)):
            pkg.debug_set("igemm_variant", v)
            row += f"  {n} {min(pkg.bench_igemm(ctx, B, H, W, Cin, Cout, 1, 4 | (8 if cold else 0), 30) for _ in range(2)) * 1e3:6.1f}"
        print(row, flush=True)
pkg.debug_set("igemm_variant", 0)
