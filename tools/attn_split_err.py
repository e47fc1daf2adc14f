"""Forensics for the split-operand attention: one-hot K / V expose the reconstructed q (hi + lo) per element, one-hot Q exposes K."""
import os, sys, torch, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge
from util import seeded
pkg = ge.load_package(); ctx = pkg.Context(0)
B, Nq, C, heads = 2, 300, 640, 10
eye = torch.eye(64).repeat(1, heads)                    # [key][h*64 + d] = (key == d)
q = seeded(B, Nq, C, seed=16)
k = (8 * eye).expand(B, 64, C).contiguous(); v = eye.expand(B, 64, C).contiguous()     # S[d] = q[d] * 8 * 0.125 = q[d]
out = pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None, heads, 3).cpu().double()
lg = out.clamp_min(1e-300).log()                         # = q - logsumexp
qd = q.double().reshape(B, Nq, heads, 64); lgd = lg.reshape(B, Nq, heads, 64)
dq = (lgd - lgd.mean(-1, keepdim=True)) - (qd - qd.mean(-1, keepdim=True))
print("q exposed through one-hot K/V: max |dq|", float(dq.abs().max()), "elements above 2e-6:", int((dq.abs() > 2e-6).sum()), "of", dq.numel())
bad = (dq.abs() > 2e-6).nonzero()
for row in bad[:12].tolist():
    b, n, h, d = row
    print(f"   b{b} q{n} head {h} d {d}: q = {float(qd[b, n, h, d]):+.6f}  dq = {float(dq[b, n, h, d]):+.3e}   lane fr = {n % 32}, wave {(n % 128) // 32}, ks {d // 16}, half {(d % 16) // 8}, e {d % 8}")
# K exposed: Q one-hot (query n of a head selects d = n % 64), keys random
kk = seeded(B, 77, C, seed=17)
qq = (8 * eye).expand(B, 64, C).contiguous()             # S[n][key] = k[key][n]
vv = torch.zeros(B, 77, C); 
for h in range(heads): vv[:, :64, h * 64:(h + 1) * 64] = torch.eye(64)[:64]
out2 = pkg.qkv_attention(ctx, qq.cuda(), kk.cuda(), vv.cuda(), None, heads, 3).cpu().double()     # out[n][d'] = P[n][key=d'] for d' < 64
lg2 = out2.clamp_min(1e-300).log().reshape(B, 64, heads, 64)         # [b][n=d][h][key]
kd = kk.double()[:, :64].reshape(B, 64, heads, 64).permute(0, 3, 2, 1)   # -> [b][d][h][key]
dk = (lg2 - lg2.mean(-1, keepdim=True)) - (kd - kd.mean(-1, keepdim=True))
# (keys 64..76 also take part in the softmax but the common log-sum-exp cancels in the centred difference only approximately: use first 64 keys' mean)
print("k exposed through one-hot Q/V: max |dk|", float(dk.abs().max()), "elements above 2e-6:", int((dk.abs() > 2e-6).sum()), "of", dk.numel())
