"""Does a launch pay for cold instruction memory?  The coarse timeline of the weights-in-registers GEMM (measure build) for
(a) the last of 20 back-to-back launches of the same kernel (its code is in the instruction cache) and (b) ONE launch behind other
kernels (the operator entry sdxl_linear packs the weights and converts the activations first: its GEMM launch finds its code cold in
the instruction cache, warm at most in the L2 / Infinity Cache) -- the situation of every launch inside a UNet step, where seven
different kernels alternate.   SDXL_MEASURE_LIB=1 python tools/icache_probe.py"""
import ctypes, os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDXL_MEASURE_LIB"] = "1"
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0); L = pkg.lib()
names = ["entry->prologue issued", "->tile 0 landed", "k-loop", "k-loop end->partial sums exchanged", "epilogue", "store drain"]
M, K, N = 2048, 1280, 1280
nwg = ((M + 95) // 96) * (N // 128)
def reduce(buf):
    raw = buf.cpu().numpy().astype(np.uint32).reshape(nwg, 8, 16).astype(np.int64)
    t = np.concatenate([raw[:, :4, :6], raw[:, :4, 8:9]], axis=2)
    d = np.diff(t, axis=2) & 0xFFFFFFFF
    life = (t[:, :, 6] - t[:, :, 0]) & 0xFFFFFFFF
    return d.mean(axis=(0, 1)), life.mean()
pkg.debug_set("igemm_variant", 60)
buf = torch.zeros(nwg * 8 * 16, dtype=torch.int32, device="cuda")
L.sdxl_debug_wreg_timeline(ctypes.c_void_p(buf.data_ptr()))
pkg.bench_igemm(ctx, 2, 32, 32, K, N, 1, 0, 20); torch.cuda.synchronize()
warm, lw = reduce(buf)
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(K, N, generator=g) / math.sqrt(K)).cuda(); b = torch.randn(N, generator=g).cuda()
cold = []
for rep in range(6):
    # other big kernels in between, as in a step: an attention launch and a GEGLU-shaped GEMM evict the instruction cache
    pkg.debug_set("igemm_variant", 0)
    pkg.bench_attention(ctx, 2, 20, 1024, 1024, 2); pkg.bench_igemm(ctx, 2, 32, 32, 1280, 10240, 1, 1, 2)
    pkg.debug_set("igemm_variant", 60)
    buf.zero_()
    pkg.linear(ctx, x, w, b, False, 1); torch.cuda.synchronize()
    cold.append(reduce(buf))
L.sdxl_debug_wreg_timeline(None); pkg.debug_set("igemm_variant", 0)
c = np.mean([c_[0] for c_ in cold], axis=0); lc = np.mean([c_[1] for c_ in cold])
print(f"out-projection 2048 x 1280 x 1280 on igemm_wreg_kernel<96, 2>, group-0 waves, cycles (mean over waves):")
print(f"{'phase':40s} {'same kernel back to back':>26s} {'one launch behind other kernels':>34s}")
for i, n in enumerate(names):
    print(f"{n:40s} {warm[i]:26.0f} {c[i]:34.0f}")
print(f"{'lifetime':40s} {lw:26.0f} {lc:34.0f}")
