"""Coarse s_memtime timeline of the wide (256x320, GEGLU) kernel: where do its ~17-20 us of fixed cost go?
    SDXL_MEASURE_LIB=1 python tools/wide_timeline.py
stamps per wave: 0 entry, 1 ring fill issued, 2 tile 0 landed (first barrier passed), 3 k-loop done, 4 ring dead (barrier),
5 first epilogue pass issued, 6 second pass issued, 7 stores drained (vmcnt(0))."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDXL_MEASURE_LIB"] = "1"
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0); L = pkg.lib()
names = ["entry->ring fill issued", "ring fill issued->tile 0 landed", "k-loop", "k-loop end->ring dead (barrier)", "epilogue pass 1", "epilogue pass 2", "store drain"]
for (name, B, H, W, Cin, Cout) in [("lin32 geglu K1280", 2, 32, 32, 1280, 10240), ("lin64 geglu K640", 2, 64, 64, 640, 5120)]:
    for cold in (1, 0):
        nwg = ((B * H * W + 255) // 256) * (Cout // 320)
        buf = torch.zeros(nwg * 8 * 16, dtype=torch.int32, device="cuda")
        pkg.debug_set("igemm_variant", 26)
        L.sdxl_debug_wide_timeline(None)
        us0 = pkg.bench_igemm(ctx, B, H, W, Cin, Cout, 1, 1 | 2 | (8 if cold else 0), 20) * 1e3       # GEGLU + folded-LayerNorm prologue, as in the step
        L.sdxl_debug_wide_timeline(ctypes.c_void_p(buf.data_ptr()))
        us = pkg.bench_igemm(ctx, B, H, W, Cin, Cout, 1, 1 | 2 | (8 if cold else 0), 20) * 1e3
        torch.cuda.synchronize()
        L.sdxl_debug_wide_timeline(None)
        raw = buf.cpu().numpy().astype(np.uint32).reshape(nwg, 8, 16).astype(np.int64)
        t = raw[:, :, :8]
        real = (raw[:, :, 9] - raw[:, :, 8]) & 0xFFFFFFFF            # 100 MHz ticks over the wave's lifetime
        d = np.diff(t, axis=2) & 0xFFFFFFFF
        life = (t[:, :, 7] - t[:, :, 0]) & 0xFFFFFFFF
        ramp = (t[:, :, 0] - t[:, :, 0].min()) & 0xFFFFFFFF
        print(f"{name} ({'cold' if cold else 'warm'} weights): {nwg} workgroups, unstamped {us0:.1f} us, stamped {us:.1f} us; wave lifetime mean {life.mean():.0f} max {life.max():.0f} cycles; shader clock {(life / np.maximum(real, 1)).mean() * 100:.0f} MHz (s_memtime / s_memrealtime over each wave's lifetime)")
        for i, n in enumerate(names):
            print(f"    {n:36s} mean {d[:, :, i].mean():8.0f}  p90 {np.percentile(d[:, :, i], 90):8.0f}  max {d[:, :, i].max():8.0f} cycles")
pkg.debug_set("igemm_variant", 0)
