"""Coarse s_memtime timeline of the key-split self-attention kernel (32^2 level of the CFG pair: 40 batch-heads x 1024 queries).
    SDXL_MEASURE_LIB=1 python tools/attn_timeline.py
stamps per wave: 0 entry, 1 Q fragments loaded + first two tiles issued, 2 tile 0 landed (first barrier), 3 k-loop done,
4 key parts merged (key part 0 only), 5 output stores issued; words 6 / 7 = s_memrealtime at entry / exit."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDXL_MEASURE_LIB"] = "1"
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0); L = pkg.lib()
for (B, H, N) in [(2, 20, 1024)]:
    nwg = B * H * (N // 64)
    buf = torch.zeros(nwg * 4 * 8, dtype=torch.int32, device="cuda")
    L.sdxl_debug_attn_timeline(None)
    us0 = pkg.bench_attention(ctx, B, H, N, N, 30) * 1e3
    L.sdxl_debug_attn_timeline(ctypes.c_void_p(buf.data_ptr()))
    us = pkg.bench_attention(ctx, B, H, N, N, 30) * 1e3
    torch.cuda.synchronize()
    L.sdxl_debug_attn_timeline(None)
    t = buf.cpu().numpy().astype(np.uint32).reshape(nwg, 4, 8).astype(np.int64)
    main = t[:, :2, :]                # waves 0, 1 = key part 0: they merge and store
    d = np.diff(main[:, :, :6], axis=2) & 0xFFFFFFFF
    life = (main[:, :, 5] - main[:, :, 0]) & 0xFFFFFFFF
    real = (main[:, :, 7] - main[:, :, 6]) & 0xFFFFFFFF
    ramp = ((t[:, :, 6] - t[:, :, 6].min()) & 0xFFFFFFFF) * 0.01      # s_memrealtime: one 100 MHz counter for the whole chip -> us
    end = ((t[:, :, 7] - t[:, :, 6].min()) & 0xFFFFFFFF) * 0.01
    print(f"self-attention B{B} H{H} N{N}: {nwg} workgroups, unstamped {us0:.1f} us, stamped {us:.1f} us; lifetime of the merging waves mean {life.mean():.0f} max {life.max():.0f} cycles, "
          f"shader clock {(life / np.maximum(real, 1)).mean() * 100:.0f} MHz; entry of a wave after the first one (us): p50 {np.median(ramp):.2f} p90 {np.percentile(ramp, 90):.2f} max {ramp.max():.2f}; exit: p50 {np.median(end):.2f} p90 {np.percentile(end, 90):.2f} max {end.max():.2f}")
    for i, n in enumerate(["entry->Q loaded, first tiles issued", "->tile 0 landed (first barrier)", "k-loop (16 tiles)", "merge of the key parts", "output transpose + stores"]):
        print(f"    {n:38s} mean {d[:, :, i].mean():8.0f}  p90 {np.percentile(d[:, :, i], 90):8.0f}  max {d[:, :, i].max():8.0f} cycles")
