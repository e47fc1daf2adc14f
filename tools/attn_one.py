"""one attention shape for PMC passes: B H Nq Nk iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
B, H, Nq, Nk, it = [int(x) for x in sys.argv[1:6]]
ms = pkg.bench_attention(ctx, B, H, Nq, Nk, it)
print(f"attn B{B} H{H} Nq{Nq} Nk{Nk}: {ms*1e3:.1f} us {4.0*B*H*Nq*Nk*64/ms/1e9:.0f} TF/s")
