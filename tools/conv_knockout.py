"""Knock-out timings of the 3x3 convolutions on the 256x128 pipelined kernel: is a conv's k-loop fetch-bound like the
linears' (DESIGN 10.2), i.e. would an LDS-resident input patch (every input row staged once instead of nine times) pay?

Measure build (SDXL_MEASURE_LIB=1), igemm_pipe_m_kernel<256,128,3> variants: 11 = the real (rolled) k-loop, 17 = DMA sources frozen along k (every
fetch after the first an L2 hit: same instruction stream, no fabric traffic), 18 = no DMA inside the k-loop at all (LDS reads + MFMAs + barriers),
27 = DMA only (no MFMAs, no LDS reads); production auto-selection (variant 0) beside them.  Results of 17 / 18 / 27 are wrong by construction.

    SDXL_MEASURE_LIB=1 python tools/conv_knockout.py > gpurun_out/r05_conv_knockout.txt
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
names = {0: "production", 11: "256x128 real", 17: "frozen ptrs", 18: "no DMA", 27: "DMA only"}
# (B, H, W, Cin, Cout): the 3x3 convolutions of a CFG step by level
shapes = [(2, 128, 128, 320, 320), (2, 128, 128, 640, 320), (2, 64, 64, 640, 640), (2, 64, 64, 1280, 640), (2, 64, 64, 1920, 640),
          (2, 32, 32, 1280, 1280), (2, 32, 32, 2560, 1280)]
print("3x3 conv, CFG pair; us per launch (best of 3 x 10 launches); K = 9 Cin")
for (B, H, W, Ci, Co) in shapes:
    row = []
    for v in (0, 11, 17, 18, 27):
        pkg.debug_set("igemm_variant", v)
        t = min(pkg.bench_igemm(ctx, B, H, W, Ci, Co, 3, False, 10) for _ in range(3)) * 1e3
        row.append(f"{names[v]} {t:7.1f}")
    pkg.debug_set("igemm_variant", 0)
    M, K = B * H * W, 9 * Ci
    fl = 2.0 * M * Co * K
    print(f"M {M:6d} N {Co:5d} K {K:6d}: " + " | ".join(row) + f" | production {fl / 1e6 / float(row[0].split()[-1]):7.1f} TFLOP/s", flush=True)
# the same contraction lengths as LINEAR layers (one fetch per input row instead of nine taps over the same rows): what the patch form could reach at best
print("same M x N x K as a linear layer (no tap walk, every A row fetched once per k-tile):")
for (B, H, W, Ci, Co) in shapes:
    K = 9 * Ci
    pkg.debug_set("igemm_variant", 11)
    t = min(pkg.bench_igemm(ctx, B, H, W, K, Co, 1, False, 10) for _ in range(3)) * 1e3
    pkg.debug_set("igemm_variant", 0)
    print(f"M {B * H * W:6d} N {Co:5d} K {K:6d}: 256x128 real {t:7.1f}", flush=True)
