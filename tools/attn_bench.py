"""Times the fused attention kernel alone on the shapes of one SDXL CFG step (B=2), per kernel variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2"])]
S = [("self 32^2", 2, 20, 1024, 1024, 60), ("self 64^2", 2, 10, 4096, 4096, 10), ("cross 32^2", 2, 20, 1024, 77, 60), ("cross 64^2", 2, 10, 4096, 77, 10)]
tot = {v: 0.0 for v in variants}
for name, B, H, Nq, Nk, cnt in S:
    fl = 4.0 * B * H * Nq * Nk * 64
    row = f"{name:12s} {fl/1e9:7.1f} GFLOP "
    for v in variants:
        pkg.debug_set("attn_variant", v)
        ms = pkg.bench_attention(ctx, B, H, Nq, Nk, 20)
        tot[v] += ms * cnt
        row += f" v{v}: {ms*1e3:7.1f} us {fl/ms/1e9:6.0f} TF/s "
    print(row, flush=True)
print("ms per step: " + "  ".join(f"v{v}: {tot[v]:.2f}" for v in variants))
