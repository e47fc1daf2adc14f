"""Launches a few implicit-GEMM shapes of the 32^2 / 64^2 levels for a rocprofv3 --pmc pass (tools/igemm_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
# (B, H, W, Cin, Cout, ksize, geglu): M = B*H*W
for shp in ((2, 32, 32, 1280, 1280, 1, 0), (2, 32, 32, 5120, 1280, 1, 0), (2, 32, 32, 1280, 3840, 1, 0), (2, 32, 32, 1280, 10240, 1, 1),
            (2, 64, 64, 640, 640, 1, 0), (2, 64, 64, 640, 640, 3, 0), (2, 32, 32, 1280, 1280, 3, 0)):
    print(shp, round(pkg.bench_igemm(ctx, *shp, 5) * 1e3, 1), "us", flush=True)
