"""Block-count balance of the f16 self-attention launches.

Part 1: variant 2 (128-query blocks) at 64^2 and the key-split variant (64-query blocks) at 32^2 for head counts that give 512, 640
and 768 equal blocks -- two per CU, the CFG pair's 2.5, three per CU.  If the launch time follows the fullest CU, 640 blocks cost
what 768 do.  Part 2: the CFG pair's two shapes under the old automatic choice (variant 9), the mixed block sizes (0 = automatic),
and the single-kernel variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)


def t(v, B, H, Nq, Nk, it=30):
    pkg.debug_set("attn_variant", v)
    try:
        return min(pkg.bench_attention(ctx, B, H, Nq, Nk, it) for _ in range(3)) * 1e3
    finally:
        pkg.debug_set("attn_variant", 0)


print("equal blocks per launch (B = 2):")
for name, v, N, heads in (("64^2 variant 2 (128-query blocks)", 2, 4096, (8, 10, 12)), ("32^2 key split (64-query blocks)", 6, 1024, (16, 20, 24))):
    for H in heads:
        blocks = 2 * H * N // (128 if v == 2 else 64)
        us = t(v, 2, H, N, N)
        print(f"  {name}  H={H:2d}  {blocks:4d} blocks ({blocks/256:.2f} per CU)  {us:7.1f} us   {us/blocks*256:6.1f} us per block-per-CU", flush=True)
print("CFG pair shapes, us per launch:")
tot = {}
for name, H, N, cnt, vs in (("self 64^2", 10, 4096, 10, (9, 0, 2, 6, 7)), ("self 32^2", 20, 1024, 60, (9, 0, 2, 6, 8))):
    row = f"  {name}: "
    for v in vs:
        us = t(v, 2, H, N, N)
        row += f" v{v} {us:6.1f}"
        if v in (9, 0): tot[v] = tot.get(v, 0.0) + us * cnt
    print(row, flush=True)
print(f"self-attention ms per step: old choice {tot[9]/1e3:.3f}   mixed block sizes {tot[0]/1e3:.3f}")
for B in (1, 4):
    for name, H, N in (("self 64^2", 10, 4096), ("self 32^2", 20, 1024)):
        print(f"  B={B} {name}: old {t(9, B, H, N, N):6.1f} us   mixed {t(0, B, H, N, N):6.1f} us", flush=True)
