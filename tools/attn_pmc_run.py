"""Launches the f16 self-attention kernels a few times each for a rocprofv3 --pmc pass (tools/attn_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
ctx = pkg.Context(0)
for v, H, N in ((2, 12, 4096), (6, 24, 1024), (0, 10, 4096)):     # 768 equal blocks each (3 per CU); the mixed launch of the CFG pair
    pkg.debug_set("attn_variant", v)
    pkg.bench_attention(ctx, 2, H, N, N, 5)
pkg.debug_set("attn_variant", 0)
