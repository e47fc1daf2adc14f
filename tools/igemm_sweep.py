"""Sweeps the implicit-GEMM kernel over the shapes of one SDXL CFG step (B=2) and the VAE, per kernel variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["-1", "0", "1", "2", "3"])]
S = [  # name, B, H, W, Cin, Cout, ksize, geglu, count per step
    ("lin32 qkv      ", 2, 32, 32, 1280, 3840, 1, 0, 60), ("lin32 out/q    ", 2, 32, 32, 1280, 1280, 1, 0, 180),
    ("lin32 geglu    ", 2, 32, 32, 1280, 10240, 1, 1, 60), ("lin32 ff       ", 2, 32, 32, 5120, 1280, 1, 0, 60),
    ("lin64 qkv      ", 2, 64, 64, 640, 1920, 1, 0, 10), ("lin64 out/q    ", 2, 64, 64, 640, 640, 1, 0, 30),
    ("lin64 geglu    ", 2, 64, 64, 640, 5120, 1, 1, 10), ("lin64 ff       ", 2, 64, 64, 2560, 640, 1, 0, 10),
    ("conv32 1280    ", 2, 32, 32, 1280, 1280, 3, 0, 9), ("conv32 2560>1280", 2, 32, 32, 2560, 1280, 3, 0, 2),
    ("conv64 640     ", 2, 64, 64, 640, 640, 3, 0, 7), ("conv64 1920>640", 2, 64, 64, 1920, 640, 3, 0, 1),
    ("conv64 1280up  ", 2, 64, 64, 1280, 1280, 3, 0, 1),
    ("conv128 320    ", 2, 128, 128, 320, 320, 3, 0, 9), ("conv128 960>320", 2, 128, 128, 960, 320, 3, 0, 1),
    ("conv128 640up  ", 2, 128, 128, 640, 640, 3, 0, 1),
    ("vae128 512     ", 1, 128, 128, 512, 512, 3, 0, 11), ("vae256 512     ", 1, 256, 256, 512, 512, 3, 0, 8),
    ("vae512 256     ", 1, 512, 512, 256, 256, 3, 0, 6), ("vae1024 128    ", 1, 1024, 1024, 128, 128, 3, 0, 6),
]
tot = {v: 0.0 for v in variants}
print("shape              GFLOP   " + "  ".join(f"v{v:>2}: ms   TF/s" for v in variants))
for name, B, H, W, Cin, Cout, k, g, cnt in S:
    fl = 2.0 * B * H * W * Cin * k * k * Cout
    row = f"{name}  {fl/1e9:7.1f}  "
    for v in variants:
        pkg.debug_set("igemm_variant", v)
        # COLD=1: rotate through > 256 MB of weight copies so every launch streams its weights from HBM, as inside a UNet step
        ms = pkg.bench_igemm(ctx, B, H, W, Cin, Cout, k, int(g) | (8 if os.environ.get("COLD") == "1" else 0), 10)
        tot[v] += ms * cnt
        row += f" {ms:7.3f} {fl/ms/1e9:6.0f}  "
    print(row, flush=True)
print("weighted ms (counts per step / decode): " + "  ".join(f"v{v}: {tot[v]:.1f}" for v in variants))
