"""Knock-out timing of the key-split self-attention kernel at the 32^2 level (measure build: SDXL_MEASURE_LIB=1): which resource bounds its
k-loop?  attn_variant: 6 production (forced key split), 11 no DMA inside the loop, 12 no softmax arithmetic, 13 no per-tile barrier, 14 no MFMAs,
15 no LDS fragment reads, 16 no DMA + no barrier, 17 no softmax + no MFMAs (DMA, barrier and LDS reads alone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
V = [("prod", 6), ("noDMA", 11), ("noSoftmax", 12), ("noBarrier", 13), ("noMFMA", 14), ("noLDSread", 15), ("noDMA+noBar", 16), ("noSoftmax+noMFMA", 17)]
S = [("32^2 CFG pair  B2 H20 N1024", 2, 20, 1024, 1024), ("32^2 one entry B1 H20 N1024", 1, 20, 1024, 1024), ("64^2 CFG pair  B2 H10 N4096", 2, 10, 4096, 4096)]
print("us per launch (min of 3 x 50 launches)")
print(f"{'shape':30}" + "".join(f"{n:>18}" for n, _ in V))
for name, B, H, Nq, Nk in S:
    row = f"{name:30}"
    for n, v in V:
        pkg.debug_set("attn_variant", v)
        us = min(pkg.bench_attention(ctx, B, H, Nq, Nk, 50) for _ in range(3)) * 1e3
        row += f"{us:18.2f}"
    print(row, flush=True)
pkg.debug_set("attn_variant", 0)
