"""Which value replaces the column-sum operand cz[62] on the rows where repeated runs of the fused cross-attention differ:
weights with cz[n] = (n + 1) / 64 exactly (only k = 0 non-zero), gamma = 1, beta = 0, so stale * 64 - 1 names the source column."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
def seeded(*s, seed): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
B, Nq, C, Nk = 2, 4096, 640, 64
x = seeded(B, Nq, C, seed=12) + 1.0
wq = torch.zeros(C, C); wq[0, :] = (torch.arange(C) + 1) / 64.0          # burn layout [K][N]: q[n] = sum_k x[k] W[k][n]
g, be = torch.ones(C), torch.zeros(C)
eye = torch.eye(64).repeat(1, C // 64)
k = (8 * eye).expand(B, Nk, C).contiguous(); v = eye.expand(B, Nk, C).contiguous()
x16 = x.half().double().reshape(-1, C)
mu = x16.mean(1); var = x16.var(1, unbiased=False); a = 1 / torch.sqrt(var + 1e-5); c = -a * mu
cz = ((torch.arange(C) + 1) / 64.0).double()
dev = [t.cuda() for t in (x, g, be, wq, k, v)]
shown = 0
for rep in range(30):
    outs = [pkg.ln_query_cross_attention(ctx, *dev, 1e-5, True)[0].cpu().reshape(-1, C).double() for _ in range(10)]
    ref = torch.stack(outs).median(0).values
    for o in outs:
        d = (o - ref).abs()
        if float(d.max()) == 0: continue
        rows = (d.amax(1) > 0).nonzero().flatten().tolist(); cols = (d.amax(0) > 0).nonzero().flatten().tolist()
        h = cols[0] // 64; n = h * 64 + 62
        est = []
        for r in rows:
            lg, lb = ref[r, h * 64:(h + 1) * 64].clamp_min(1e-300).log(), o[r, h * 64:(h + 1) * 64].clamp_min(1e-300).log()
            dq = lb - lg; dq = dq - dq[:62].mean()
            est.append(float(cz[n] + dq[62] / c[r]))
        e = torch.tensor(est)
        print(f"rows {rows[0]}-{rows[-1]} ({len(rows)}) head {h}: cz[{n}] = {float(cz[n]):.5f}; stale = {float(e.median()):.5f} (min {float(e.min()):.5f} max {float(e.max()):.5f}) -> stale * 64 - 1 = {float(e.median()) * 64 - 1:.2f}", flush=True)
        shown += 1
        if shown >= 12: sys.exit(0)
