"""A/B of the weights-in-registers GEMM (igemm_wreg.hip, variants 60 / 61 / 62 = 96 / 128 / 64 rows) against the auto selection
without it, on the plain linear shapes of one SDXL CFG step: cold weights (every launch streams its weights from HBM, as inside
the step) and with the row-statistics epilogue the production launches carry.   python tools/wreg_ab.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S = [  # name, B, H, W, Cin, Cout, count per step
    ("lin32 out-proj K1280", 2, 32, 32, 1280, 1280, 120), ("lin32 ff-out K5120 ", 2, 32, 32, 5120, 1280, 60),
    ("lin64 out-proj K640 ", 2, 64, 64, 640, 640, 20), ("lin64 ff-out K2560 ", 2, 64, 64, 2560, 640, 10),
    ("skip32 2560>1280    ", 2, 32, 32, 2560, 1280, 2), ("skip64 1920>640     ", 2, 64, 64, 1920, 640, 1),
    ("proj32 in/out       ", 2, 32, 32, 1280, 1280, 12), ("lin16 out-proj (512)", 2, 16, 16, 1280, 1280, 0),
]
cols = [("auto-old", 0, 0), ("wreg96", 60, 1), ("wreg128", 61, 1), ("wreg64", 62, 1), ("auto-new", 0, 1)]
print("shape                   GFLOP   " + "  ".join(f"{n:>9}: us  TF/s" for n, _, _ in cols))
tot = [0.0] * len(cols)
for name, B, H, W, Cin, Cout, cnt in S:
    fl = 2.0 * B * H * W * Cin * Cout
    row = f"{name}  {fl/1e9:7.1f}  "
    for ci, (n, v, on) in enumerate(cols):
        pkg.debug_set("igemm_wreg", on)
        pkg.debug_set("igemm_variant", v)
        best = 1e9
        for rep in range(2):
            best = min(best, pkg.bench_igemm(ctx, B, H, W, Cin, Cout, 1, 4 | 8, iters))
        tot[ci] += best * cnt
        row += f"   {best*1e3:8.1f} {fl/best/1e9:5.0f}"
    print(row, flush=True)
pkg.debug_set("igemm_variant", 0); pkg.debug_set("igemm_wreg", 1)
print("weighted ms per step: " + "  ".join(f"{n}: {t:.2f}" for (n, _, _), t in zip(cols, tot)))
