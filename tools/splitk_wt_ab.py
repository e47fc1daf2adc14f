"""A/B of the split-K slab publication (knob splitk_wt): write-through stores against plain stores + agent-scope release.  us per launch of the
long-K 3x3 convolutions of the 32^2 level (three k-slices per 256x128 tile, combined inside the launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
S = [("conv32 1280>1280 K11520", 2, 32, 32, 1280, 1280, 3), ("conv32 2560>1280 K23040", 2, 32, 32, 2560, 1280, 3), ("conv32 1920>1280 K17280", 2, 32, 32, 1920, 1280, 3)]
for rep in range(2):
    for v in (1, 0):
        pkg.debug_set("splitk_wt", v)
        for name, B, H, W, Cin, Cout, ks in S:
            us = min(pkg.bench_igemm(ctx, B, H, W, Cin, Cout, ks, 8, 20) for _ in range(3)) * 1e3
            print(f"splitk_wt={v} {name}: {us:.1f} us", flush=True)
pkg.debug_set("splitk_wt", 1)
