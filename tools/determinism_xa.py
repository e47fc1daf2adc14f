"""Repeat the fused cross-attention operator on identical inputs and describe where (if anywhere) the runs differ."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
for kv in filter(None, os.environ.get("SDXL_DEBUG_SET", "").split(",")):
    k, v = kv.split("="); pkg.debug_set(k, int(v))
def seeded(*s, seed): return torch.randn(*s, generator=torch.Generator().manual_seed(seed)).cuda()
def ranges(ix):
    out, ix = [], sorted(ix)
    for i in ix:
        if out and out[-1][1] == i - 1: out[-1][1] = i
        else: out.append([i, i])
    return ",".join(f"{a}-{b}" if a != b else f"{a}" for a, b in out)
REP = int(os.environ.get("REP", "6"))
SHAPES = ((2, 4096, 640),) if os.environ.get("ONLY") else ((2, 4096, 640), (1, 4096, 640), (2, 1024, 1280))
for (B, Nq, C) in SHAPES:
    x = seeded(B, Nq, C, seed=12); g, be = 1 + 0.1 * seeded(C, seed=13), 0.1 * seeded(C, seed=14)
    wq = seeded(C, C, seed=15) / math.sqrt(C); k, v = seeded(B, 77, C, seed=16), seeded(B, 77, C, seed=17)
    for fused in ((True,) if os.environ.get("ONLY") else (True, False)):
        outs = []
        for r in range(REP):
            junk = torch.full((64 << 20,), float("nan"), device="cuda"); del junk; torch.cuda.empty_cache()
            outs.append(pkg.ln_query_cross_attention(ctx, x, g, be, wq, k, v, 1e-5, fused)[0].cpu().reshape(-1, C))
        ref = torch.stack(outs).median(0).values
        print(f"B{B} Nq{Nq} C{C} fused={fused}:", flush=True)
        for r, o in enumerate(outs):
            d = (o - ref).abs()
            if float(d.max()) == 0: continue
            rows = (d.amax(1) > 0).nonzero().flatten().tolist(); cols = (d.amax(0) > 0).nonzero().flatten().tolist()
            sub = d[rows][:, cols]
            print(f"  run {r}: max {float(d.max()):.3e} n_bad {int((d > 0).sum())} rows[{len(rows)}] {ranges(rows)} cols[{len(cols)}] {ranges(cols)} "
                  f"bad-per-row min/max {int((sub > 0).sum(1).min())}/{int((sub > 0).sum(1).max())} nan {int(torch.isnan(o).sum())} "
                  f"|ref| there {float(ref[rows][:, cols].abs().mean()):.3f} mean diff {float(sub.mean()):.2e}", flush=True)
