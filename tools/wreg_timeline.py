"""Coarse s_memtime timeline of the weights-in-registers GEMM (96-row tiles): where does the ~7.6 us fixed cost of a launch go?
    SDXL_MEASURE_LIB=1 python tools/wreg_timeline.py
stamps per wave: 0 entry, 1 prologue issued (first L tiles of weights + activation pieces), 2 tile 0 landed (first barrier, first
fragments requested), 3 k-loop done, 4 partial sums of the two k-groups exchanged, 5 epilogue issued (group 0: stores; row statistics
exchanged), 8 stores drained; words 6 / 7 = s_memrealtime at entry / exit."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDXL_MEASURE_LIB"] = "1"
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0); L = pkg.lib()
names = ["entry->prologue issued", "->tile 0 landed", "k-loop", "k-loop end->partial sums exchanged", "epilogue (stores + row statistics)", "store drain"]
for (name, B, H, W, Cin, Cout) in [("out-proj K1280", 2, 32, 32, 1280, 1280), ("ff-out K5120", 2, 32, 32, 5120, 1280)]:
    for cold in (1, 0):
        nwg = ((B * H * W + 95) // 96) * (Cout // 128)
        buf = torch.zeros(nwg * 8 * 16, dtype=torch.int32, device="cuda")
        pkg.debug_set("igemm_variant", 60)
        L.sdxl_debug_wreg_timeline(None)
        us0 = pkg.bench_igemm(ctx, B, H, W, Cin, Cout, 1, 4 | (8 if cold else 0), 20) * 1e3
        L.sdxl_debug_wreg_timeline(ctypes.c_void_p(buf.data_ptr()))
        us = pkg.bench_igemm(ctx, B, H, W, Cin, Cout, 1, 4 | (8 if cold else 0), 20) * 1e3
        torch.cuda.synchronize()
        L.sdxl_debug_wreg_timeline(None)
        raw = buf.cpu().numpy().astype(np.uint32).reshape(nwg, 8, 16).astype(np.int64)
        for gname, sl in (("group 0 (waves 0-3: epilogue)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
            t = np.concatenate([raw[:, sl, :6], raw[:, sl, 8:9]], axis=2)
            d = np.diff(t, axis=2) & 0xFFFFFFFF
            life = (t[:, :, 6] - t[:, :, 0]) & 0xFFFFFFFF
            real = (raw[:, sl, 7] - raw[:, sl, 6]) & 0xFFFFFFFF
            print(f"{name} ({'cold' if cold else 'warm'}), {gname}: unstamped {us0:.1f} us, stamped {us:.1f} us; lifetime mean {life.mean():.0f} max {life.max():.0f} cycles, clock {(life / np.maximum(real, 1)).mean() * 100:.0f} MHz")
            for i, n in enumerate(names):
                print(f"    {n:40s} mean {d[:, :, i].mean():8.0f}  p90 {np.percentile(d[:, :, i], 90):8.0f}  max {d[:, :, i].max():8.0f}")
pkg.debug_set("igemm_variant", 0)
