"""DMA-only / contiguous-source measurement modes of the pipelined GEMM (igemm variants 27..32) next to the real kernels:
t(real) vs t(DMA only) vs t(DMA only, 1 KiB contiguous pieces) vs t(full compute, contiguous pieces).  Timing only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
names = {11: "256x128 real", 27: "256x128 dma-only", 28: "256x128 dma-only contig", 29: "256x128 real contig",
         13: "128x128 real", 30: "128x128 dma-only", 31: "128x128 dma-only contig", 32: "128x128 real contig", 18: "256x128 no-dma"}
shapes = [(2, 32, 32, 10240), (2, 32, 32, 3840), (2, 32, 32, 1280), (2, 64, 64, 5120), (2, 64, 64, 640)]
for (B, H, W, N) in shapes:
    M = B * H * W
    for K in (1280, 5120):
        row = []
        for v in (11, 18, 27, 28, 29, 13, 30, 31, 32):
            pkg.debug_set("igemm_variant", v)
            t = pkg.bench_igemm(ctx, B, H, W, K, N, 1, False, 10) * 1e3
            row.append(f"{names[v]}={t:.1f}")
        print(f"M={M} N={N} K={K}: " + "  ".join(row), flush=True)
