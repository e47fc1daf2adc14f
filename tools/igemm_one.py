"""Runs ONE implicit-GEMM shape with ONE forced kernel variant (for rocprofv3 --pmc passes): variant B H W Cin Cout k geglu iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
v, B, H, W, Cin, Cout, k, g, iters = [int(x) for x in sys.argv[1:10]]
pkg.debug_set("igemm_variant", v)
ms = pkg.bench_igemm(ctx, B, H, W, Cin, Cout, k, bool(g), iters)
fl = 2.0 * B * H * W * Cin * k * k * Cout
print(f"variant {v} shape {B}x{H}x{W} Cin={Cin} Cout={Cout} k={k} geglu={g}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF/s")
