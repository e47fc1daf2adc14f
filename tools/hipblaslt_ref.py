"""Upper-bound probe: what does the vendor GEMM (hipBLASLt through torch.matmul, f16 in / f32 accumulate) reach on the UNet
step's linear shapes on this box?  Weights rotate through > 256 MB of copies (cold, as inside a step).  Not part of the engine."""
import sys, torch
shapes = [("lin32 qkv", 2048, 1280, 3840), ("lin32 out/q", 2048, 1280, 1280), ("lin32 geglu", 2048, 1280, 10240),
          ("lin32 ff", 2048, 5120, 1280), ("lin64 geglu", 8192, 640, 5120), ("lin64 ff", 8192, 2560, 640),
          ("conv32-as-gemm", 2048, 11520, 1280)]
dev = torch.device("cuda:0")
for name, M, K, N in shapes:
    ncopy = max(2, int(300e6 / (K * N * 2)) + 1)
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    ws = [torch.randn(N, K, device=dev, dtype=torch.float16) for _ in range(ncopy)]
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for i in range(3):
        torch.matmul(a, ws[i % ncopy].t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for i in range(reps):
        torch.matmul(a, ws[(3 + i) % ncopy].t(), out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:16s} M={M} K={K} N={N}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TF/s", flush=True)
