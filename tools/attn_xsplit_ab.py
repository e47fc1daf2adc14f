import os, sys
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
for rep in range(2):
    for v in (1, 0):
        pkg.debug_set("attn_xsplit", v)
        for (B, H, N) in [(2, 20, 1024), (1, 20, 1024), (4, 20, 1024), (2, 10, 4096), (1, 10, 4096)]:
            us = min(pkg.bench_attention(ctx, B, H, N, N, 50) for _ in range(3)) * 1e3
            print(f"attn_xsplit={v} B{B} H{H} N{N}: {us:.2f} us", flush=True)
pkg.debug_set("attn_xsplit", 1)
