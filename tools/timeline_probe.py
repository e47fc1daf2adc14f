"""Instruction-level timeline of the production implicit-GEMM schedule (verdict r2 item 4): where the ~45 % s_waitcnt goes.

    SDXL_MEASURE_LIB=1 python tools/timeline_probe.py [out.json]

Runs the s_memtime-stamped twins of the production kernels (igemm_measure.hip: variants 145 = 96x128 / 5 slots, 136 = 128x128 / 4,
135 = 256x128 / 3; same schedule, ~3 scalar round trips per k-tile of overhead) on the step's dominant linear shapes, next to the
unstamped production kernel, and reduces the per-wave stamps to a per-phase cycle table:
  launch -> entry      (first stamp of a workgroup relative to the earliest one in the grid: dispatch ramp)
  prologue issue       T1 - T0: address set-up + the NS-1 tiles of DMA pieces issued
  first tile wait      T2 - T1: ring fill (HBM / L2 latency under the whole grid's burst) + first barrier
  per k-tile           compute (C[kt-1] -> A[kt]: ds_read + MFMA + DMA issue), dma wait (A -> B: own pieces of tile kt+1),
                       barrier wait (B -> C: the other waves' pieces / compute)
  epilogue             T4 - T3: LDS staging, residual loads, stores (to vmcnt(0))
s_memtime ticks are shader cycles (MI355X_MICROARCH.md); 100 ticks = ~42-50 ns at the 2.0-2.4 GHz these kernels clock.
"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDXL_MEASURE_LIB"] = "1"
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
ctx = pkg.Context(0)
L = pkg.lib()
TLW = L.sdxl_debug_timeline_words()

SHAPES = [  # name, B, H, W, Cin, Cout, production variant, timeline variant, (BM, BN, waves)
    ("ff_out 2048x1280 K5120", 2, 32, 32, 5120, 1280, 45, 145, (96, 128, 6)),
    ("attn_out 2048x1280 K1280", 2, 32, 32, 1280, 1280, 45, 145, (96, 128, 6)),
    ("qkv 2048x3840 K1280", 2, 32, 32, 1280, 3840, 35, 135, (256, 128, 8)),
    ("ff_out64 8192x640 K2560", 2, 64, 64, 2560, 640, 35, 135, (256, 128, 8)),
    ("conv64 640 3x3 K5760", 2, 64, 64, 640, 640, 35, 135, (256, 128, 8), 3),
    ("conv128 320 3x3 K2880", 2, 128, 128, 320, 320, 35, 135, (256, 128, 8), 3),
]


def run(name, B, H, W, Cin, Cout, vprod, vtl, tile, ksize=1):
    bm, bn, nw = tile
    M = B * H * W
    nwg = ((M + bm - 1) // bm) * ((Cout + bn - 1) // bn)
    pkg.debug_set("igemm_variant", vprod)
    L.sdxl_debug_timeline(None)
    # COLD=1: rotate through > 256 MB of weight copies, so every launch streams its weights from HBM as it does inside a UNet step
    flags = 8 if os.environ.get("COLD") == "1" else 0
    us_prod = pkg.bench_igemm(ctx, B, H, W, Cin, Cout, ksize, flags, 20) * 1e3
    buf = torch.zeros(nwg * nw * TLW, dtype=torch.int32, device="cuda")
    L.sdxl_debug_timeline(ctypes.c_void_p(buf.data_ptr()))
    pkg.debug_set("igemm_variant", vtl)
    us_tl = pkg.bench_igemm(ctx, B, H, W, Cin, Cout, ksize, flags, 20) * 1e3     # the buffer holds the LAST launch's stamps
    torch.cuda.synchronize()
    L.sdxl_debug_timeline(None)
    t = buf.cpu().numpy().astype(np.uint32).reshape(nwg, nw, TLW).astype(np.int64)
    nk = int(t[0, 0, 5])
    T0, T1, T2, T3, T4 = (t[:, :, i] for i in range(5))
    xcc = t[:, 0, 7]
    d = lambda a, b: ((a - b) & 0xFFFFFFFF).astype(np.float64)     # noqa: E731  (32-bit wrap)
    # s_memtime is per-XCD: compare entry times only inside one XCD
    ramp = np.concatenate([d(T0[xcc == x], T0[xcc == x].min()).ravel() for x in np.unique(xcc)])
    kt = np.arange(nk - 1)
    A, Bm, C = t[:, :, 16 + 3 * kt], t[:, :, 17 + 3 * kt], t[:, :, 18 + 3 * kt]
    dma_wait, bar_wait = d(Bm, A), d(C, Bm)
    compute = d(A[:, :, 1:], C[:, :, :-1])
    first_compute = d(A[:, :, 0], T2)
    tail = d(T3, C[:, :, -1])                       # last k-tile's compute (no barrier behind it)
    total = d(T4, T0)
    loop = d(T3, T2)
    P1, P2, P3, P4, E1 = (t[:, :, i] for i in (8, 9, 10, 11, 12))
    st = lambda x: dict(mean=float(x.mean()), p50=float(np.median(x)), p90=float(np.percentile(x, 90)), max=float(x.max()))   # noqa: E731
    rep = dict(shape=name, M=M, N=Cout, K=Cin * ksize * ksize, tile=f"{bm}x{bn}", workgroups=nwg, waves=nw, k_tiles=nk,
               us_production=us_prod, us_with_stamps=us_tl,
               dispatch_ramp=st(ramp), prologue_issue=st(d(T1, T0)), first_tile_wait=st(d(T2, T1)),
               prologue_parts=dict(kernel_args=st(d(P1, T0)), dma_geometry=st(d(P2, P1)), first_tap_pointers=st(d(P3, P2)),
                                   acc_zero_frag_addr_ln_load=st(d(P4, P3)), dma_issue_ring_fill=st(d(T1, P4))),
               epilogue_parts=dict(barrier_and_issue=st(d(E1, T3)), store_drain=st(d(T4, E1))),
               epilogue_rows=dict(last_barrier=st(d(t[:, :, 13], T3)), residual_request=st(d(t[:, :, 14], t[:, :, 13])),
                                  vectors_request=st(d(t[:, :, 15], t[:, :, 14])), math_swap_stores=st(d(t[:, :, 6], t[:, :, 15])),
                                  tail=st(d(E1, t[:, :, 6]))) if int(t[0, 0, 13]) != 0 else None,
               kernarg_env=os.environ.get("HIP_FORCE_DEV_KERNARG"),
               first_compute=st(first_compute),
               per_ktile=dict(compute=st(compute), dma_wait=st(dma_wait), barrier_wait=st(bar_wait),
                              sum_mean=float(compute.mean() + dma_wait.mean() + bar_wait.mean())),
               last_tile_compute=st(tail), epilogue=st(d(T4, T3)), k_loop=st(loop), kernel_wave_lifetime=st(total),
               dma_wait_by_ktile_mean=[float(x) for x in dma_wait.mean(axis=(0, 1))[:64]],
               barrier_wait_by_ktile_mean=[float(x) for x in bar_wait.mean(axis=(0, 1))[:64]],
               compute_by_wave_mean=[float(x) for x in compute.mean(axis=(0, 2))],
               dma_wait_by_wave_mean=[float(x) for x in dma_wait.mean(axis=(0, 2))],
               barrier_wait_by_wave_mean=[float(x) for x in bar_wait.mean(axis=(0, 2))])
    pk = rep["per_ktile"]
    print(f"{name}: {bm}x{bn} x {nwg} WGs, {nk} k-tiles | production {us_prod:.1f} us, stamped {us_tl:.1f} us")
    print(f"   ramp p50/max {rep['dispatch_ramp']['p50']:.0f}/{rep['dispatch_ramp']['max']:.0f}  prologue {rep['prologue_issue']['mean']:.0f}  "
          f"first-tile wait {rep['first_tile_wait']['mean']:.0f} (max {rep['first_tile_wait']['max']:.0f})  epilogue {rep['epilogue']['mean']:.0f} "
          f"(max {rep['epilogue']['max']:.0f})  lifetime {rep['kernel_wave_lifetime']['mean']:.0f} cycles")
    pp, ep = rep["prologue_parts"], rep["epilogue_parts"]
    print("   prologue: " + "  ".join(f"{k} {v['mean']:.0f}" for k, v in pp.items()) + " | epilogue: " + "  ".join(f"{k} {v['mean']:.0f}" for k, v in ep.items()))
    if rep["epilogue_rows"]:
        print("   direct epilogue: " + "  ".join(f"{k} {v['mean']:.0f}" for k, v in rep["epilogue_rows"].items()))
    print(f"   per k-tile: compute {pk['compute']['mean']:.0f} (p90 {pk['compute']['p90']:.0f})  dma wait {pk['dma_wait']['mean']:.0f} (p90 {pk['dma_wait']['p90']:.0f})  "
          f"barrier wait {pk['barrier_wait']['mean']:.0f} (p90 {pk['barrier_wait']['p90']:.0f})  = {pk['sum_mean']:.0f} cycles")
    return rep


def main():
    if os.environ.get("EPI_STAGED"):
        pkg.debug_set("igemm_epilogue_staged", int(os.environ["EPI_STAGED"]))
    out = [run(*s) for s in SHAPES]
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/timeline_probe.json"
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
