"""Knock-out timing of the weights-in-registers GEMM (measure build: SDXL_MEASURE_LIB=1): which resource bounds its k-loop?
variants: 60 production, 63 no MFMAs, 64 operand pointers frozen (L1 / L2 hits only), 65 no VMEM in the k-loop, 66 no k-loop barriers,
67 production arithmetic with the XCDs owning row tiles instead of weight column tiles; per-stream knock-outs (exact waits): 70 weight stream
only, 71 activation pieces only, 72 activation pieces read as contiguous 1-KiB runs (a k-tile-major activation layout; same bytes);
73 no MFMAs + frozen pointers (fetch stream + LDS reads, L2 hits), 74 the same without the LDS fragment reads, 75 also without barriers,
76 fetch stream alone on real pointers (no MFMAs, no LDS reads), 77 production without the LDS reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
S = [("lin32 out-proj K1280", 2, 32, 32, 1280, 1280), ("lin32 ff-out K5120 ", 2, 32, 32, 5120, 1280), ("lin32 K20480 (probe)", 2, 32, 32, 20480, 1280)]
V = [("prod", 60), ("noMFMA", 63), ("frozen", 64), ("noVMEM", 65), ("nobar", 66), ("xcdrow", 67), ("Wonly", 70), ("Aonly", 71), ("Acontig", 72), ("F+L2", 73), ("F+L2-lds", 74), ("..-bar", 75), ("F-lds", 76), ("prod-lds", 77), ("pipe96", 45)]
for cold in (1, 0):
    print(f"--- {'cold weights (rotating copies)' if cold else 'warm (one weight copy)'}; us per launch")
    print("shape                    " + "".join(f"{n:>9}" for n, _ in V))
    for name, B, H, W, Cin, Cout in S:
        row = f"{name}     "
        for n, v in V:
            print(f"[{name.strip()} {n}]", file=sys.stderr, flush=True)
            pkg.debug_set("igemm_variant", v)
            ms = min(pkg.bench_igemm(ctx, B, H, W, Cin, Cout, 1, 4 | (8 if cold else 0), 20) for _ in range(2))
            row += f"{ms*1e3:9.1f}"
        print(row, flush=True)
pkg.debug_set("igemm_variant", 0)
