"""us per launch of forced attention variants on one shape: python tools/attn_variant_times.py B H N v1,v2,..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
B, H, N = (int(x) for x in sys.argv[1:4])
for rep in range(2):
    row = f"B={B} H={H} N={N}:"
    for v in (int(x) for x in sys.argv[4].split(",")):
        pkg.debug_set("attn_variant", v)
        row += f"  v{v} {min(pkg.bench_attention(ctx, B, H, N, N, 30) for _ in range(3))*1e3:7.1f}"
    print(row, flush=True)
pkg.debug_set("attn_variant", 0)
