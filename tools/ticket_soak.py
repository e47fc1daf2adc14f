"""Soak of the three cross-workgroup publication protocols of round 4 (VERDICT r4 "weak" 4 / "next" 6): 2 x N launches each on identical
inputs, caches disturbed every 16 launches (the pattern that exposed the section-9.2 hazard, tools/hazard_xa_probe.py), every output
bit-compared with the first.  Zero differing launches = the protocol delivered the same bytes every time whichever workgroup arrived last.

  1. attention key halves merged across workgroups (attention.hip attn_xhalf_merge: write-through image, relaxed ticket, sc1 loads) -- the
     32^2 self-attention of the CFG pair (B 2, 20 heads, 1024 tokens) and a single entry;
  2. split-K slabs published by write-through stores (igemm_pipe_kernel, splitk = 3): the K = 11520 conv of the 32^2 level and a K = 10240 linear;
  3. weight warming (idle workgroups of a weights-in-registers launch read a later launch's weights): one full-size UNet::forward -- the
     recording eager forward (no warmers), then N graph replays with warmers, then N replays with the knob off.

    python tools/ticket_soak.py [launches per run, default 500] > gpurun_out/r05_ticket_soak.txt
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
ctx = pkg.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500


def seeded(*s, seed):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed)).cuda()


def soak(name, fn, runs=2, n=N):
    total_bad = 0
    for run in range(runs):
        ref, bad, worst = None, 0, 0.0
        for r in range(n):
            if r % 16 == 0:        # disturb the caches between bursts: 256 MB of junk through the L2s / Infinity Cache
                junk = torch.full((64 << 20,), float("nan"), device="cuda"); del junk
            o = fn()
            if ref is None:
                ref = o.clone()
            elif not torch.equal(o, ref):
                bad += 1; worst = max(worst, float((o.float() - ref.float()).abs().max()))
        total_bad += bad
        print(f"{name}: run {run}: {bad} of {n} launches differ from the first (max |diff| {worst:.3e}), finite {bool(torch.isfinite(ref.float()).all())}", flush=True)
    return total_bad


bad = 0
# 1. attention key halves across workgroups
for (B, heads, T) in ((2, 20, 1024), (1, 20, 1024), (2, 10, 1024)):
    q, k, v = seeded(B, T, 64 * heads, seed=41), seeded(B, T, 64 * heads, seed=42), seeded(B, T, 64 * heads, seed=43)
    bad += soak(f"attention key halves B{B} H{heads} N{T}", lambda: pkg.qkv_attention(ctx, q, k, v, None, heads, 1))

# 2. split-K slabs
x = seeded(2, 1280, 32, 32, seed=43); w = seeded(1280, 1280, 3, 3, seed=44) / math.sqrt(1280 * 9); b = 0.1 * seeded(1280, seed=45)
bad += soak("split-K conv3x3 K=11520 (2 x 32^2 -> 1280)", lambda: pkg.conv2d(ctx, x, w, b, 1, 1, False, 1), n=max(N // 2, 50))
xl = seeded(1024, 10240, seed=7); wl = seeded(10240, 1280, seed=8) / math.sqrt(10240); bl = 0.1 * seeded(1280, seed=9)
bad += soak("split-K linear 1024 x 10240 -> 1280", lambda: pkg.linear(ctx, xl, wl, bl, False, 1), n=max(N // 2, 50))

# 3. weight warming inside the captured UNet forward
cfg = pkg.sdxl_base_config()
u = pkg.UNet(ctx, cfg, pkg.DTYPE_F16, seed=0)
xi, ts = seeded(2, 4, 128, 128, seed=1), torch.tensor([500, 500], dtype=torch.int32).cuda()
cx, y = seeded(2, 77, cfg.context_dim, seed=2), seeded(2, cfg.adm_in_channels, seed=4)
first = u.forward(xi, ts, cx, y).clone()            # eager: records the schedule, no warmers
bad += soak("UNet::forward 1024^2, captured graph with warming workgroups", lambda: u.forward(xi, ts, cx, y), n=max(N // 2, 50))
warm = u.forward(xi, ts, cx, y).clone()
print(f"recording forward (no warmers) == warmed graph replay: {bool(torch.equal(first, warm))}", flush=True)
bad += 0 if torch.equal(first, warm) else 1
pkg.debug_set("igemm_warm", 0)
u2 = pkg.UNet(ctx, cfg, pkg.DTYPE_F16, seed=0)
u2.forward(xi, ts, cx, y)
bad += soak("UNet::forward 1024^2, knob igemm_warm=0", lambda: u2.forward(xi, ts, cx, y), runs=1, n=max(N // 4, 50))
off = u2.forward(xi, ts, cx, y)
pkg.debug_set("igemm_warm", 1)
print(f"warmed == un-warmed engine: {bool(torch.equal(off, warm))}", flush=True)
bad += 0 if torch.equal(off, warm) else 1
print(f"TOTAL differing launches / comparisons: {bad}", flush=True)
