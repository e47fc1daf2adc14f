"""Per-op parity: HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (max-abs error relative to the output's max-abs):
  * DTYPE_F32 (exact-fp32 MFMA, fp32 storage): 2e-5 -- fp32 round-off class (summation order differs from libtorch).
  * DTYPE_F16 (fp16 operands, fp32 accumulate): 4e-3 -- one fp16 rounding of the operands (2^-11 = 4.9e-4 each) plus
    the fp16 rounding of the stored result.
Inputs follow the reference's probe recipe arb_tensor = sin(arange) (src/bin/test/main.rs:51-54) or a seeded normal.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import config as OC, model as OM
from util import rel_err, seeded

pytestmark = pytest.mark.gpu

TOL = {0: 2e-5, 1: 4e-3, 2: 4e-3}
DTYPES = [0, 1, 2]


def arb(*shape):
    return torch.from_numpy(OC.arb_tensor(*shape))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,C,H,W,silu", [(2, 64, 8, 8, True), (1, 320, 16, 16, True), (2, 32, 4, 4, False),
                                          (1, 2560, 8, 8, True), (1, 128, 64, 32, False), (3, 960, 5, 7, True)])
def test_group_norm(pkg, ctx, dtype, B, C, H, W, silu):
    x = seeded(B, C, H, W, seed=1) * 3.0 + 0.7 + arb(B, C, H, W)
    gamma, beta = 1 + 0.1 * seeded(C, seed=2), 0.1 * seeded(C, seed=3)
    ref = OM.group_norm(x, gamma, beta)
    if silu:
        ref = OM.silu(ref)
    out = pkg.group_norm(ctx, x.cuda(), gamma.cuda(), beta.cuda(), 32, 1e-5, silu, dtype)
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_group_norm_large_mean(pkg, ctx, dtype):
    # |mean| >> std: the case a sum / sum-of-squares variance gets wrong; Welford + Chan merge must not
    x = seeded(1, 64, 32, 32, seed=5) * 0.05 + 40.0
    gamma, beta = torch.ones(64), torch.zeros(64)
    ref = OM.group_norm(x, gamma, beta)
    out = pkg.group_norm(ctx, x.cuda(), gamma.cuda(), beta.cuda(), 32, 1e-5, False, 0 if dtype == 0 else 2)
    assert rel_err(out, ref) < (2e-3 if dtype == 0 else 6e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(7, 64), (300, 640), (64, 1280), (5, 1536), (2, 2048 + 64)])
def test_layer_norm(pkg, ctx, dtype, rows, C):
    x = seeded(rows, C, seed=4) * 2.0 + 0.3
    gamma, beta = 1 + 0.1 * seeded(C, seed=5), 0.1 * seeded(C, seed=6)
    ref = OM.layer_norm(x, gamma, beta)
    out = pkg.layer_norm(ctx, x.cuda(), gamma.cuda(), beta.cuda(), 1e-5, dtype)
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N,geglu", [(300, 640, 640, False), (77, 1280, 3840, False), (256, 640, 5120, True),
                                         (130, 64, 128, False), (64, 2048, 256, False)])
def test_layer_norm_linear(pkg, ctx, dtype, M, K, N, geglu):
    # LayerNorm -> Linear as TransformerBlock::forward pairs them (unet/mod.rs:885-891).  dtype 1 (f16) runs the FOLDED path:
    # statistics from the producer's epilogue, rstd/mean applied in the consumer's epilogue (K = 2048: > 24 slots, generic loop)
    x = seeded(M, K, seed=4) * 2.0 + 0.3
    gamma, beta = 1 + 0.1 * seeded(K, seed=5), 0.1 * seeded(K, seed=6)
    w = seeded(K, N, seed=8) / math.sqrt(K)
    b = 0.1 * seeded(N, seed=9)
    eps = 1e-3
    h = OM.layer_norm(x.half().float() if dtype == 1 else x, gamma, beta, eps) @ w + b
    ref = h[:, :N // 2] * torch.nn.functional.gelu(h[:, N // 2:]) if geglu else h
    out = pkg.layer_norm_linear(ctx, x.cuda(), gamma.cuda(), beta.cuda(), w.cuda(), b.cuda(), eps, geglu, dtype)
    e = rel_err(out, ref)
    print(f"layer_norm_linear M={M} K={K} N={N} geglu={geglu} dtype={dtype}: rel err {e:.3e}")
    assert e < TOL[dtype]


@pytest.mark.parametrize("B,Cin,H,W,Cout,res,expect", [
    (2, 640, 64, 64, 640, False, True),        # ResBlock conv1 -> norm2 at the 64^2 level: 256x128 tiles
    (2, 1280, 32, 32, 1280, True, True),       # 32^2 level, K = 11520: the split-K form of the same kernel, + skip residual
    (2, 320, 64, 64, 640, True, True),         # channel-doubling ResBlock of the 64^2 level (K = 2880)
    (2, 1920, 64, 64, 640, False, False),      # K = 17280 at 64^2: 128x160 tiles save more than the statistics launch costs
    (1, 64, 16, 16, 128, False, False),        # a grid of one round: the selection prefers 96x128 tiles -> statistics pass stays
    (2, 320, 128, 128, 320, False, False),     # 128^2 level of the CFG pair: the selection prefers 256x160 tiles -> statistics pass stays
    (2, 128, 24, 24, 128, True, False),        # 576 rows per entry: not whole 256-row tiles
])
def test_conv2d_group_norm_statistics_from_producer(pkg, ctx, B, Cin, H, W, Cout, res, expect):
    # conv3x3 -> GroupNorm(32)+SiLU as ResBlock::forward pairs them (unet/mod.rs:1082-1106).  fused: the convolution's epilogue
    # leaves per-tile, per-channel (mean, M2) and the norm runs no statistics pass; both paths against the fp32 oracle, and the
    # shapes the 256x128 kernel does not take must fall back (fused_taken False) with identical results.
    x = (seeded(B, Cin, H, W, seed=31) * 1.2 + 0.3).half().float()
    w = seeded(Cout, Cin, 3, 3, seed=32) / math.sqrt(9 * Cin)
    b = 0.5 * seeded(Cout, seed=33) + 2.0                      # channel means well away from zero: |mean| > sigma
    r = seeded(B, Cout, H, W, seed=34).half().float() if res else None
    gamma, beta = 1 + 0.1 * seeded(Cout, seed=35), 0.1 * seeded(Cout, seed=36)
    h = F.conv2d(x, w, b, padding=1)
    if res:
        h = h + r
    ref = F.silu(OM.group_norm(h, gamma, beta, 32, 1e-5))
    args = (ctx, x.cuda(), w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), 1e-5, 32, True, None if r is None else r.cuda())
    out_f, took = pkg.conv2d_group_norm(*args, fused=True)
    out_p, took_p = pkg.conv2d_group_norm(*args, fused=False)
    e_f, e_p = rel_err(out_f, ref), rel_err(out_p, ref)
    print(f"conv->GN B={B} Cin={Cin} {H}x{W} Cout={Cout} res={res}: fused taken={took} rel err {e_f:.3e}, stats-pass path {e_p:.3e}")
    assert took == expect and not took_p
    assert e_f < TOL[1] and e_p < TOL[1]
    if not took:
        assert torch.equal(out_f, out_p)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("B,Nq,Nk,C", [(2, 1024, 77, 1280), (1, 4096, 77, 640), (2, 64, 77, 128), (1, 128, 96, 64),
                                        (3, 192, 5, 192), (2, 256, 33, 1280)])
def test_ln_query_cross_attention(pkg, ctx, fused, B, Nq, Nk, C):
    # attn2 up to its output projection (unet/mod.rs:731-795): LayerNorm -> query projection -> attention over the projected
    # context, 64 channels per head.  fused=True is the production path of the f16 UNet: the projection's waves run the
    # attention on their own accumulator tiles (128x128 tiles at 32^2, 256x128 at 64^2); fused=False = projection + kernel.
    x = (seeded(B, Nq, C, seed=21) * 1.5 + 0.2).half().float()
    gamma, beta = 1 + 0.1 * seeded(C, seed=5), 0.1 * seeded(C, seed=6)
    wq = seeded(C, C, seed=22) / math.sqrt(C)
    k, v = seeded(B, Nk, C, seed=23), seeded(B, Nk, C, seed=24)
    k[0, 0] *= 3.0                                   # one dominant key: the softmax is not near-uniform
    eps = 1e-5
    q = OM.layer_norm(x, gamma, beta, eps) @ wq
    ref = OM.qkv_attention(q, k, v, None, C // 64)
    out = pkg.ln_query_cross_attention(ctx, x.cuda(), gamma.cuda(), beta.cuda(), wq.cuda(), k.cuda(), v.cuda(), eps, fused)
    e = rel_err(out, ref)
    print(f"ln_query_cross_attention fused={fused} B={B} Nq={Nq} Nk={Nk} C={C}: rel err {e:.3e}")
    assert e < TOL[1]


@pytest.mark.parametrize("B,Nq,Nk,C", [(2, 1024, 77, 1280), (1, 4096, 77, 640), (2, 256, 33, 1280), (1, 128, 96, 64)])
def test_ln_query_cross_attention_split_precision(pkg, ctx, B, Nq, Nk, C):
    # fused=2 (IgemmParams::xa_k_lo; SDXL_DTYPE_F32_SPLIT_MIX_F16W): the epilogue's attention on (hi, lo) pairs of the context, of q (the projection's fp32
    # accumulators) and of P.  Reference: the same f16-rounded x and W -- what the f16 projection multiplies -- with everything behind the projection in
    # fp64; what is left is the output's own rounding to f16 (2^-11 relative per element) plus the fp32-class attention.  The plain f16 epilogue (fused=1)
    # additionally rounds q, the context and P to f16: it must be the less accurate one.
    x = (seeded(B, Nq, C, seed=21) * 1.5 + 0.2).half().float()
    gamma, beta = 1 + 0.1 * seeded(C, seed=5), 0.1 * seeded(C, seed=6)
    wq = (seeded(C, C, seed=22) / math.sqrt(C)).half().float()
    k, v = seeded(B, Nk, C, seed=23), seeded(B, Nk, C, seed=24)
    k[0, 0] *= 3.0
    # (the engine folds the LayerNorm: x W' with W' = f16(gamma W) -- round the folded matrix the same way for the reference)
    wfold = (gamma[:, None] * wq).half().double()
    xd = x.double()
    mu, var = xd.mean(-1, keepdim=True), xd.var(-1, unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    q = rstd * (xd @ wfold) - rstd * mu * wfold.sum(0) + (beta.double() @ wq.double())
    ref = OM.qkv_attention(q, k.double(), v.double(), None, C // 64).float()
    o2 = pkg.ln_query_cross_attention(ctx, x.cuda(), gamma.cuda(), beta.cuda(), wq.cuda(), k.cuda(), v.cuda(), 1e-5, 2)
    o1 = pkg.ln_query_cross_attention(ctx, x.cuda(), gamma.cuda(), beta.cuda(), wq.cuda(), k.cuda(), v.cuda(), 1e-5, 1)
    e2, e1 = rel_err(o2, ref), rel_err(o1, ref)
    print(f"ln_query_cross_attention split precision B={B} Nq={Nq} Nk={Nk} C={C}: rel err {e2:.3e} (plain f16 epilogue {e1:.3e})")
    assert e2 < 6e-4 and e2 < e1          # 2^-11 = 4.9e-4: the output rows are f16
    # sharper: an fp32-class result rounded once to f16 IS f16(reference) except where the reference sits on a rounding boundary -- and then one ulp away.
    # The un-fused twin (fp32 q through memory, stand-alone split-operand attention kernel, fused=3) is held to the same.
    o3 = pkg.ln_query_cross_attention(ctx, x.cuda(), gamma.cuda(), beta.cuda(), wq.cuda(), k.cuda(), v.cuda(), 1e-5, 3)
    # per element: half an f16 ulp AT the reference value (the single rounding of the output rows: 2^(floor(log2 |v|) - 11)) + an fp32-class residue
    tol = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(6.2e-5))) - 11.0) + 3e-6 * float(ref.abs().max())
    res = {}
    for name, o in (("fused", o2), ("attention kernel", o3), ("plain f16 epilogue", o1)):
        over = float(((o.cpu() - ref).abs() > tol).float().mean())
        worst = float(((o.cpu() - ref).abs() / tol).max())
        res[name] = (over, worst)
        print(f"   {name}: {over:.2e} of the outputs beyond (half an f16 ulp + 3e-6 max|ref|), worst {worst:.2f} x that")
    print(f"   fused vs attention kernel: {float((o2 != o3).float().mean()):.2e} of the f16 outputs differ, max |diff| / max|ref| {float((o2 - o3).abs().max() / ref.abs().max()):.2e}")
    for name in ("fused", "attention kernel"):
        assert res[name][0] == 0.0, (name, res[name])
    assert res["plain f16 epilogue"][0] > 0.02        # the f16 epilogue rounds q, the context and P: the bar does resolve the difference


def test_ln_query_cross_attention_refuses_long_context(pkg, ctx):
    # more than 96 keys do not fit the in-register softmax: the fused entry refuses, the two-kernel path takes it
    B, Nq, Nk, C = 1, 64, 100, 64
    x, k, v = seeded(B, Nq, C, seed=1), seeded(B, Nk, C, seed=2), seeded(B, Nk, C, seed=3)
    g, b_, wq = torch.ones(C), torch.zeros(C), seeded(C, C, seed=4) / 8
    with pytest.raises(RuntimeError):
        pkg.ln_query_cross_attention(ctx, x.cuda(), g.cuda(), b_.cuda(), wq.cuda(), k.cuda(), v.cuda(), 1e-5, True)
    out = pkg.ln_query_cross_attention(ctx, x.cuda(), g.cuda(), b_.cuda(), wq.cuda(), k.cuda(), v.cuda(), 1e-5, False)
    ref = OM.qkv_attention(OM.layer_norm(x.half().float(), g, b_, 1e-5) @ wq, k, v, None, 1)
    assert rel_err(out, ref) < TOL[1]


@pytest.mark.parametrize("dtype", DTYPES)
def test_layer_norm_linear_large_mean(pkg, ctx, dtype):
    # rows with |mean| >> sigma (outlier channels of a real residual stream): a (sum, sum^2) variance loses every digit here
    # (40^2 = 1600 against sigma^2 = 0.0025 in fp32); the folded path carries shifted per-slot (mean, M2) + a Chan merge.
    # Inputs are fp16-representable so the f16 path and the oracle see the same rows.  Row 0: zero variance (rstd = eps^-1/2).
    M, K, N = 192, 1280, 1280
    x = (40.0 + 0.05 * seeded(M, K, seed=14)).half().float()
    x[0] = 7.0
    x[1] = (-300.0 + 0.5 * seeded(K, seed=15)).half().float()
    gamma, beta = 1 + 0.1 * seeded(K, seed=5), 0.1 * seeded(K, seed=6)
    w = seeded(K, N, seed=8) / math.sqrt(K)
    ref = OM.layer_norm(x, gamma, beta, 1e-5) @ w
    out = pkg.layer_norm_linear(ctx, x.cuda(), gamma.cuda(), beta.cuda(), w.cuda(), None, 1e-5, False, dtype)
    e = rel_err(out, ref)
    print(f"layer_norm_linear large-mean dtype={dtype}: rel err {e:.3e}")
    assert e < (1e-3 if dtype == 0 else 1.1e-3)       # f16 stream: measured 5.2e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N", [(1, 64, 64), (77, 128, 192), (130, 20, 8), (256, 320, 1280), (1000, 640, 100),
                                   (64, 2816, 1280), (4096, 64, 64)])
def test_linear(pkg, ctx, dtype, M, K, N):
    x = seeded(M, K, seed=7)
    w = seeded(K, N, seed=8) / math.sqrt(K)
    b = 0.1 * seeded(N, seed=9)
    ref = x @ w + b
    out = pkg.linear(ctx, x.cuda(), w.cuda(), b.cuda(), False, dtype)
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("M,K,N,geglu", [(77, 128, 192, False), (256, 320, 1280, False), (1000, 640, 96, False), (64, 2816, 1280, False),
                                           (2048, 1280, 1280, False), (256, 640, 5120, True), (300, 1280, 10240, True), (130, 64, 128, True)])
def test_linear_split_operand(pkg, ctx, M, K, N, geglu):
    # SDXL_DTYPE_F32_SPLIT as an operator: HL16 operands, three MFMAs per product, incl. the exact-erf GEGLU epilogue (unet/mod.rs:942-956)
    x = seeded(M, K, seed=7).double()
    w = (seeded(K, N, seed=8) / math.sqrt(K)).double()
    b = (0.1 * seeded(N, seed=9)).double()
    y = x @ w + b
    ref = y[:, :N // 2] * torch.nn.functional.gelu(y[:, N // 2:]) if geglu else y
    out = pkg.linear(ctx, x.float().cuda(), w.float().cuda(), b.float().cuda(), geglu, 3)
    e = rel_err(out.double(), ref)
    print(f"linear split-operand M={M} K={K} N={N} geglu={geglu}: rel err vs fp64 {e:.3e}")
    assert e < 5e-6


@pytest.mark.parametrize("M,K,N,geglu", [(256, 320, 1280, False), (2048, 1280, 1280, False), (300, 1280, 10240, True)])
def test_linear_split_operand_exact_f16_weights(pkg, ctx, M, K, N, geglu):
    # weights that ARE f16 values (what a real SDXL record holds, src/bin/sample/main.rs:37): the packed lo halves are zero and the
    # kernel leaves out the w_lo x a_hi MFMAs -- the result must be bit-identical to the three-MFMA form, and as accurate
    x = seeded(M, K, seed=7)
    w = (seeded(K, N, seed=8) / math.sqrt(K)).half().float()
    b = 0.1 * seeded(N, seed=9)
    y = x.double() @ w.double() + b.double()
    ref = y[:, :N // 2] * torch.nn.functional.gelu(y[:, N // 2:]) if geglu else y
    out = pkg.linear(ctx, x.cuda(), w.cuda(), b.cuda(), geglu, 3)
    pkg.debug_set("hl_weights_exact", 0)
    try:
        out3 = pkg.linear(ctx, x.cuda(), w.cuda(), b.cuda(), geglu, 3)
    finally:
        pkg.debug_set("hl_weights_exact", 1)
    assert torch.equal(out, out3), "two-MFMA form differs from the three-MFMA form on exact-f16 weights"
    assert rel_err(out.double(), ref) < 5e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_asymmetric_identity(pkg, ctx, dtype):
    # A = I with an asymmetric B catches a transposed / mis-mapped MFMA C layout (guide rule 16)
    n = 128
    x = torch.eye(n)
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) / 251.0
    out = pkg.linear(ctx, x.cuda(), w.cuda(), None, False, dtype)
    assert rel_err(out, w) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N", [(64, 64, 128), (300, 640, 2 * 4 * 640 // 8), (33, 128, 64)])
def test_geglu(pkg, ctx, dtype, M, K, N):
    x = seeded(M, K, seed=10)
    w = seeded(K, N, seed=11) / math.sqrt(K)
    b = 0.1 * seeded(N, seed=12)
    pr = x @ w + b
    ref = pr[:, : N // 2] * F.gelu(pr[:, N // 2:])
    out = pkg.linear(ctx, x.cuda(), w.cuda(), b.cuda(), True, dtype)
    assert rel_err(out, ref) < TOL[dtype] * 2


CONVS = [  # B, Cin, H, W, Cout, k, stride, pad, upsample
    (1, 64, 8, 8, 64, 3, 1, 1, False),
    (2, 128, 16, 12, 192, 3, 1, 1, False),
    (1, 4, 16, 16, 64, 3, 1, 1, False),      # stem: per-element gather path
    (1, 64, 16, 16, 4, 3, 1, 1, False),      # conv_out: N=4
    (2, 64, 16, 16, 64, 3, 2, 1, False),     # Downsample (unet/mod.rs:765-772)
    (1, 64, 9, 7, 128, 3, 2, 1, False),      # odd sizes
    (1, 128, 8, 8, 64, 1, 1, 0, False),      # skip_connection 1x1
    (2, 64, 8, 8, 64, 3, 1, 1, True),        # Upsample::forward (:742-752)
    (1, 64, 16, 16, 64, 3, 2, 0, False),     # stride-2, pad 0 window (the PaddedConv2d taps, cropped)
    (1, 320, 32, 32, 320, 3, 1, 1, False),
    (1, 3, 16, 16, 32, 3, 1, 1, False),      # VAE encoder stem
    (1, 96, 8, 8, 64, 3, 1, 1, False),       # Cin % 64 != 0 -> generic gather
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,pad,up", CONVS)
def test_conv2d(pkg, ctx, dtype, B, Cin, H, W, Cout, k, stride, pad, up):
    x = seeded(B, Cin, H, W, seed=13)
    w = seeded(Cout, Cin, k, k, seed=14) / math.sqrt(Cin * k * k)
    b = 0.1 * seeded(Cout, seed=15)
    xi = OM.upsample_nearest2x(x) if up else x
    ref = F.conv2d(xi, w, b, stride=stride, padding=pad)
    out = pkg.conv2d(ctx, x.cuda(), w.cuda(), b.cuda(), stride, pad, up, dtype)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[dtype]


SPLIT_CONVS = [  # B, Cin, H, W, Cout, k, stride, pad, upsample -- the VAE's shapes in small: Cin % 32 == 0
    (1, 32, 8, 8, 32, 3, 1, 1, False),       # narrow output: wave tiles cut by N (staged epilogue)
    (2, 64, 16, 16, 128, 3, 1, 1, False),
    (1, 128, 32, 32, 256, 1, 1, 0, False),   # 1x1 (nin_shortcut, attention q / k / v / proj)
    (1, 64, 16, 16, 64, 3, 2, 0, False),     # PaddedConv2d taps (stride 2, pad 0)
    (1, 64, 8, 8, 64, 3, 1, 1, True),        # upsampler: nearest 2x in the gather
    (1, 512, 16, 16, 512, 3, 1, 1, False),   # K = 4608: 144 k-tiles of 32
    (3, 96, 9, 7, 160, 3, 1, 1, False),      # ragged rows / columns
    (1, 128, 16, 16, 3, 3, 1, 1, False),     # conv_out: N = 3
]


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,pad,up", SPLIT_CONVS)
def test_conv2d_split_operand(pkg, ctx, B, Cin, H, W, Cout, k, stride, pad, up):
    """SDXL_DTYPE_F32_SPLIT as an operator: operands as (hi, lo) f16 pairs, a*w ~ ah*wh + al*wh + ah*wl on the f16 MFMA, fp32
    accumulation -- must sit in the fp32 class against an fp64 reference.  Inputs span 5 decades (values whose lo half is an f16
    subnormal included) and the weights are small (|w| ~ 0.02 / sqrt(fan-in): without the packing's power-of-two scale their lo
    halves would all be subnormal)."""
    g = torch.Generator().manual_seed(16)
    x = seeded(B, Cin, H, W, seed=13) * torch.pow(10.0, torch.rand(B, Cin, 1, 1, generator=g) * 5.0 - 4.0)
    w = 0.02 * seeded(Cout, Cin, k, k, seed=14) / math.sqrt(Cin * k * k)
    b = 0.01 * seeded(Cout, seed=15)
    xi = OM.upsample_nearest2x(x) if up else x
    ref = F.conv2d(xi.double(), w.double(), b.double(), stride=stride, padding=pad)
    out = pkg.conv2d(ctx, x.cuda(), w.cuda(), b.cuda(), stride, pad, up, pkg.DTYPE_F32_SPLIT).cpu().double()
    out32 = pkg.conv2d(ctx, x.cuda(), w.cuda(), b.cuda(), stride, pad, up, pkg.DTYPE_F32).cpu().double()
    assert out.shape == ref.shape
    e = float((out - ref).abs().max() / ref.abs().max())
    e32 = float((out32 - ref).abs().max() / ref.abs().max())
    print(f"conv2d split-operand {(B, Cin, H, W, Cout, k, stride, pad, up)}: rel err vs fp64 {e:.3e} (exact-fp32 MFMA: {e32:.3e})")
    assert e < 2e-6, e


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("B,Nq,Nk,C,heads,masked", [
    (1, 64, 64, 64, 1, False), (2, 256, 256, 128, 2, False), (2, 300, 77, 640, 10, False),   # cross-attn Nk=77
    (1, 77, 77, 128, 2, True),                                                              # CLIP-style causal mask
    (1, 1024, 1024, 1280, 20, False), (1, 50, 130, 64, 2, False),                           # generic head dim 32
    (1, 96, 96, 512, 1, False),                                                             # VAE mid: 1 head, d=512
    (1, 40, 40, 96, 2, True)])
def test_qkv_attention(pkg, ctx, dtype, B, Nq, Nk, C, heads, masked):
    q, k, v = seeded(B, Nq, C, seed=16), seeded(B, Nk, C, seed=17), seeded(B, Nk, C, seed=18)
    mask = OM.attn_decoder_mask(max(Nq, Nk))[:Nq, :Nk].contiguous() if masked else None
    ref = OM.qkv_attention(q, k, v, mask, heads)
    out = pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None if mask is None else mask.cuda(), heads, dtype)
    assert rel_err(out, ref) < (1e-4 if dtype == 0 else 6e-3)


@pytest.mark.parametrize("B,Nq,Nk,C,heads", [(1, 64, 64, 64, 1), (2, 256, 256, 128, 2), (2, 300, 77, 640, 10), (1, 1024, 1024, 1280, 20),
                                             (1, 130, 200, 64, 1), (2, 64, 1, 64, 1), (1, 33, 192, 64, 1), (2, 4096, 4096, 128, 2)])
def test_qkv_attention_split_operand(pkg, ctx, B, Nq, Nk, C, heads):
    # SDXL_DTYPE_F32_SPLIT: fp32 Q / O, K and V^T as (hi, lo) f16 pairs, three MFMAs per product (attn_d64_hl_kernel) -- held to the
    # strict mode's bound against a float64 reference; ragged query blocks, key tails (Nk = 77, 1, 200), several heads / entries
    q, k, v = seeded(B, Nq, C, seed=16), seeded(B, Nk, C, seed=17), seeded(B, Nk, C, seed=18)
    ref = OM.qkv_attention(q.double(), k.double(), v.double(), None, heads)
    out = pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None, heads, 3)
    e = rel_err(out.double(), ref)
    print(f"split-operand attention B={B} Nq={Nq} Nk={Nk} C={C}: rel err {e:.3e}")
    assert e < 5e-6
    out2 = pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None, heads, 3)
    assert torch.equal(out, out2)


def test_qkv_attention_split_operand_rescale(pkg, ctx):
    B, N, C = 1, 320, 64
    q, k, v = seeded(B, N, C, seed=19), seeded(B, N, C, seed=20), seeded(B, N, C, seed=21)
    k[0, 200] = q[0, 3] * 6.0
    k[0, 290] = q[0, 77] * 2.5
    k[0, 10] = q[0, 130] * 9.0
    ref = OM.qkv_attention(q.double(), k.double(), v.double(), None, 1)
    out = pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None, 1, 3)
    assert rel_err(out.double(), ref) < 5e-6


@pytest.mark.parametrize("dtype,variant", [(0, 0), (1, 0), (1, 1), (1, 2), (1, 6), (1, 7), (1, 8)])
def test_qkv_attention_online_softmax_rescale(pkg, ctx, dtype, variant):
    # keys far above the rest in LATE tiles force the running-max rescale branch (guide rule 26): for the deferred-max
    # f16 kernel both the "exceeds the threshold" path (spikes) and the "stays below it" path (all other tiles) run
    B, N, C = 1, 320, 64
    q, k, v = seeded(B, N, C, seed=19), seeded(B, N, C, seed=20), seeded(B, N, C, seed=21)
    k[0, 200] = q[0, 3] * 6.0
    k[0, 290] = q[0, 77] * 2.5
    k[0, 10] = q[0, 130] * 9.0        # spike in the FIRST tile: later tiles sit far below the reference
    ref = OM.qkv_attention(q, k, v, None, 1)
    pkg.debug_set("attn_variant", variant)
    try:
        out = pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None, 1, dtype)
    finally:
        pkg.debug_set("attn_variant", 0)
    assert rel_err(out, ref) < (1e-4 if dtype == 0 else 6e-3)


@pytest.mark.parametrize("B,Nq,Nk,heads", [(1, 96, 96, 1), (2, 200, 77, 1), (1, 1024, 1024, 1), (2, 130, 333, 2), (1, 64, 1, 1),
                                           (1, 4096, 4096, 1)])
def test_qkv_attention_wide_head_flash(pkg, ctx, B, Nq, Nk, heads):
    # head dim 512 (the VAE mid block's single head, autoencoder/mod.rs:550-586): the flash kernel attn_hd_kernel -- ragged
    # query / key counts (tails of the 64-query blocks and 32-key tiles), several heads and batch entries
    C = 512 * heads
    q, k, v = seeded(B, Nq, C, seed=16), seeded(B, Nk, C, seed=17), seeded(B, Nk, C, seed=18)
    ref = OM.qkv_attention(q, k, v, None, heads)
    out = pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None, heads, 1)
    assert rel_err(out, ref) < 6e-3


def test_qkv_attention_wide_head_flash_rescale(pkg, ctx):
    # the running max must move late in the key sequence: one key aligned with one query far beyond the rest (the O / l
    # rescale branch is taken in a late tile for that query only), against a full fp32 reference
    B, N, C = 1, 512, 512
    q, k, v = seeded(B, N, C, seed=21) * 0.3, seeded(B, N, C, seed=22) * 0.3, seeded(B, N, C, seed=23)
    k[0, 400] = q[0, 37] * 3.0
    k[0, 17] = q[0, 300] * 2.0
    ref = OM.qkv_attention(q, k, v, None, 1)
    out = pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None, 1, 1)
    assert rel_err(out, ref) < 6e-3


# 6 = key-split kernel (64-query blocks, waves = query sub-tile x key half); 7 / 8 = mixed block sizes (the first 4/5 of the heads in
# 128- / 64-query blocks, the rest in 64-query key-split / 32-query key-quarter blocks: attn_d64_mix_kernel)
@pytest.mark.parametrize("variant", [1, 2, 6, 7, 8])
@pytest.mark.parametrize("B,Nq,Nk,C,heads", [(2, 256, 256, 128, 2), (2, 300, 77, 640, 10), (1, 1024, 1024, 1280, 20),
                                             (1, 130, 200, 64, 1), (2, 64, 1, 64, 1), (2, 100, 128, 128, 2), (1, 33, 192, 64, 1),
                                             (2, 300, 320, 320, 5), (3, 1000, 192, 640, 10)])
def test_qkv_attention_f16_variants(pkg, ctx, variant, B, Nq, Nk, C, heads):
    q, k, v = seeded(B, Nq, C, seed=16), seeded(B, Nk, C, seed=17), seeded(B, Nk, C, seed=18)
    ref = OM.qkv_attention(q, k, v, None, heads)
    pkg.debug_set("attn_variant", variant)
    try:
        out = pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None, heads, 1)
    finally:
        pkg.debug_set("attn_variant", 0)
    assert rel_err(out, ref) < 6e-3


@pytest.mark.parametrize("variant,heads,N,B", [(7, 10, 384, 3), (8, 20, 320, 3), (0, 10, 3328, 2), (0, 20, 256, 3)])
def test_qkv_attention_mixed_block_sizes_are_their_bodies(pkg, ctx, variant, heads, N, B):
    # attn_d64_mix_kernel only re-assigns (head, query block) to blocks: the first 4/5 of the heads must come out bit-identical to
    # the large-block kernel alone (7: variant 2, 8: key split), at level 0 the rest bit-identical to the key-split kernel alone; and
    # a batch entry must not depend on its neighbours (the head split is a function of one entry's shape).  Variant 0 = the automatic
    # choice: mixed at (10 heads, 3328 queries) -- 26 x 10 >= 256 blocks of 128 queries per entry --, key split alone at (20, 256)
    C = 64 * heads
    q, k, v = seeded(B, N, C, seed=31), seeded(B, N, C, seed=32), seeded(B, N, C, seed=33)

    def run(var, sl=slice(None)):
        pkg.debug_set("attn_variant", var)
        try:
            return pkg.qkv_attention(ctx, q[sl].cuda(), k[sl].cuda(), v[sl].cuda(), None, heads, 1)
        finally:
            pkg.debug_set("attn_variant", 0)

    out = run(variant)
    assert rel_err(out, OM.qkv_attention(q, k, v, None, heads)) < 6e-3
    assert torch.equal(run(variant, slice(1, 2)), out[1:2])
    big = 64 * (heads * 4 // 5)
    if variant == 0 and N < 3328:
        # automatic choice where the key-split kernel runs: the first 4/5 of the heads in whole-key blocks (= the key-split body), the rest as
        # two half-key blocks per 64 queries merged across workgroups (test_qkv_attention_key_halves_across_workgroups); knob off = key split alone
        assert torch.equal(out[..., :big], run(6)[..., :big])
        pkg.debug_set("attn_xsplit", 0)
        try:
            assert torch.equal(run(0), run(6))
        finally:
            pkg.debug_set("attn_xsplit", 1)
    else:
        assert torch.equal(out[..., :big], run(6 if variant == 8 else 2)[..., :big])
        if variant != 8:
            assert torch.equal(out[..., big:], run(6)[..., big:])


@pytest.mark.parametrize("B,heads,N", [(2, 20, 1024), (1, 20, 1024), (3, 10, 384), (1, 5, 128), (2, 5, 2048)])
def test_qkv_attention_key_halves_across_workgroups(pkg, ctx, B, heads, N):
    # attn_d64_mix_kernel level 2 (the automatic choice where the key-split kernel would run): the last 1/5 of the heads run as TWO blocks per
    # 64 queries, one per key half, and the second of the two to arrive merges the halves in the fixed order (half 0, half 1): the result must
    # not depend on the arrival order (repeated launches bit-identical), on the batch neighbours, and must match the oracle like the other heads
    C = 64 * heads
    q, k, v = seeded(B, N, C, seed=41), seeded(B, N, C, seed=42), seeded(B, N, C, seed=43)
    ref = OM.qkv_attention(q, k, v, None, heads)
    outs = [pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None, heads, 1) for _ in range(6)]
    big = 64 * max(1, heads * 4 // 5)
    assert rel_err(outs[0], ref) < 6e-3 and rel_err(outs[0][..., big:], ref[..., big:]) < 6e-3
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "the merge depends on which half arrived last"
    one = pkg.qkv_attention(ctx, q[B - 1:].cuda(), k[B - 1:].cuda(), v[B - 1:].cuda(), None, heads, 1)
    assert torch.equal(one, outs[0][B - 1:]), "a batch entry depends on its neighbours"
    pkg.debug_set("attn_xsplit", 0)
    try:
        off = pkg.qkv_attention(ctx, q.cuda(), k.cuda(), v.cuda(), None, heads, 1)
    finally:
        pkg.debug_set("attn_xsplit", 1)
    assert torch.equal(off[..., :big], outs[0][..., :big])            # the whole-key heads are the same blocks either way
    assert rel_err(off[..., big:], outs[0][..., big:]) < 2e-3         # the split heads differ by rounding only (two-level online softmax)


def test_attn_decoder_mask(pkg, ctx):
    assert torch.equal(pkg.attn_decoder_mask(ctx, 77).cpu(), OM.attn_decoder_mask(77))


@pytest.mark.parametrize("M,K,N", [(1024, 10240, 1280), (300, 11520, 200), (1000, 10304, 640), (64, 16384, 1280), (1, 12800, 64)])
def test_linear_split_k(pkg, ctx, M, K, N):
    # long contraction over a small output (rows <= 1024, N <= 1280, K >= 10240): three k-slices per 256x128 tile, combined inside
    # the launch by the last-arriving slice in slice order -> bit-reproducible, and the arrival counters re-arm themselves
    x = seeded(M, K, seed=7)
    w = seeded(K, N, seed=8) / math.sqrt(K)
    b = 0.1 * seeded(N, seed=9)
    ref = x @ w + b
    outs = [pkg.linear(ctx, x.cuda(), w.cuda(), b.cuda(), False, 1) for _ in range(3)]
    assert rel_err(outs[0], ref) < TOL[1]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), "split-K result is not bit-reproducible"


def test_conv_split_k_matches_batch_entries(pkg, ctx):
    # 3x3 conv at 32x32 (K = 9 * 1280 = 11520): the slices start inside the tap walk; the split depends on ONE batch entry's
    # shape only, so a batch of two equals two separate launches bit for bit
    x = seeded(2, 1280, 32, 32, seed=43)
    w = seeded(320, 1280, 3, 3, seed=44) / math.sqrt(1280 * 9)
    b = 0.1 * seeded(320, seed=45)
    ref = F.conv2d(x, w, b, padding=1)
    both = pkg.conv2d(ctx, x.cuda(), w.cuda(), b.cuda(), 1, 1, False, 1)
    assert rel_err(both, ref) < TOL[1]
    for i in range(2):
        one = pkg.conv2d(ctx, x[i:i + 1].cuda(), w.cuda(), b.cuda(), 1, 1, False, 1)
        assert torch.equal(one[0], both[i])


# ---------------------------------------------------------------------------------------------------------
# every fast-path implicit-GEMM tile / pipeline variant is forced in turn over shapes that exercise: fewer k-tiles than
# ring slots, ragged M / N tiles, GEGLU pairs, 3x3 taps with halo zero-fill, stride 2 and the fused nearest-2x gather
# production kernels (what the auto selection launches) -- always built; the A/B partners and dead-end experiments exist only
# in a measure build (`build.py --measure`) and are exercised when the loaded library is one
IGEMM_VARIANTS = [4, 6, 26, 35, 36, 38, 44, 45, 46, 47, 49]
IGEMM_MEASURE_VARIANTS = [40, 41, 42, 43, 1, 8, 10, 11, 12, 13, 14, 15, 16, 19, 20, 21, 22, 23, 24, 25, 33, 34, 37]


def _variant_built(pkg, variant):
    return variant in IGEMM_VARIANTS or "measure build" in pkg.lib().sdxl_build_info().decode()


@pytest.fixture
def igemm_variant(pkg):
    def set_variant(v):
        pkg.debug_set("igemm_variant", v)
    yield set_variant
    pkg.debug_set("igemm_variant", 0)


@pytest.mark.parametrize("variant", IGEMM_VARIANTS + IGEMM_MEASURE_VARIANTS)
def test_igemm_variants_linear(pkg, ctx, igemm_variant, variant):
    if not _variant_built(pkg, variant):
        pytest.skip("experimental variant: measure builds only")
    igemm_variant(variant)
    for (M, K, N, geglu) in [(300, 640, 320, False), (2048, 64, 128, False), (520, 128, 200, False), (257, 192, 136, False),
                             (300, 640, 640, True), (1024, 1280, 512, True), (4096, 320, 1280, False),
                             (600, 256, 320, True), (2048, 1280, 1280, True), (520, 192, 320, False)]:
        x = seeded(M, K, seed=40)
        w = seeded(K, N, seed=41) / math.sqrt(K)
        b = 0.1 * seeded(N, seed=42)
        pr = x @ w + b
        ref = pr[:, : N // 2] * F.gelu(pr[:, N // 2:]) if geglu else pr
        out = pkg.linear(ctx, x.cuda(), w.cuda(), b.cuda(), geglu, 1)
        e = rel_err(out, ref)
        assert e < TOL[1] * (2 if geglu else 1), f"variant {variant} M={M} K={K} N={N} geglu={geglu}: rel err {e}"


@pytest.mark.parametrize("variant", IGEMM_VARIANTS + IGEMM_MEASURE_VARIANTS)
def test_igemm_variants_conv(pkg, ctx, igemm_variant, variant):
    if not _variant_built(pkg, variant):
        pytest.skip("experimental variant: measure builds only")
    igemm_variant(variant)
    for (B, Cin, H, W, Cout, k, stride, pad, up) in [(1, 128, 20, 20, 192, 3, 1, 1, False), (2, 64, 16, 16, 64, 3, 2, 1, False),
                                                     (2, 64, 8, 8, 64, 3, 1, 1, True), (1, 320, 32, 32, 320, 3, 1, 1, False),
                                                     (1, 128, 9, 7, 64, 1, 1, 0, False), (2, 192, 17, 13, 128, 3, 1, 1, False)]:
        x = seeded(B, Cin, H, W, seed=43)
        w = seeded(Cout, Cin, k, k, seed=44) / math.sqrt(Cin * k * k)
        b = 0.1 * seeded(Cout, seed=45)
        xi = OM.upsample_nearest2x(x) if up else x
        ref = F.conv2d(xi, w, b, stride=stride, padding=pad)
        out = pkg.conv2d(ctx, x.cuda(), w.cuda(), b.cuda(), stride, pad, up, 1)
        e = rel_err(out, ref)
        assert e < TOL[1], f"variant {variant} conv {(B, Cin, H, W, Cout, k, stride, pad, up)}: rel err {e}"


@pytest.mark.parametrize("variant", [35, 36, 4, 6, 11, 13, 21, 23, 24, 25])
def test_igemm_variants_unet(pkg, ctx, igemm_variant, variant):
    if not _variant_built(pkg, variant):
        pytest.skip("experimental variant: measure builds only")
    # residual / time-embedding / transposed-V^T epilogues of the pipelined kernels, through a whole tiny UNet
    from util import to_pkg_cfg, unet_weights
    igemm_variant(variant)
    ocfg = OC.tiny_config()
    W = unet_weights(ocfg)
    B, H, Wd = 2, 16, 16
    x = torch.from_numpy(OC.arb_tensor(B, 4, H, Wd))
    context = torch.from_numpy(OC.arb_tensor(B, 5, ocfg.context_dim))
    y = torch.from_numpy(OC.arb_tensor(B, ocfg.adm_in_channels))
    t = torch.tensor([999, 1], dtype=torch.int32)
    ref = OM.unet_forward(ocfg, W, x, t.long(), context, y)
    u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), 1, seed=0)
    out = u.forward(x.cuda(), t.cuda(), context.cuda(), y.cuda()).cpu()
    assert rel_err(out, ref) < 4.5e-3          # <= 2x the measured class (test_gpu_models.FWD_TOL)


# ---------------------------------------------------------------------------------------------------------
# weights-in-registers GEMM (igemm_wreg.hip): plain f16 linear layers with N % 128 == 0, K % 64 == 0, K >= 128.  Variants 60 / 61 / 62
# force 96 / 128 / 64 rows per tile.  Shapes: fewer k-tiles per group than the prefetch depth (K = 128 .. 512), odd k-tile counts (the
# two k-groups carry different tile counts), ragged M tails, the UNet's own shapes, long contractions.
WREG_SHAPES = [(300, 640, 640), (2048, 1280, 1280), (100, 128, 128), (4096, 320, 640), (96, 192, 256), (1000, 5120, 1280),
               (33, 2560, 384), (257, 256, 128), (511, 448, 256), (64, 576, 128), (130, 704, 384), (2048, 1344, 256)]


@pytest.mark.parametrize("variant", [60, 62])
def test_igemm_wreg_linear(pkg, ctx, igemm_variant, variant):
    igemm_variant(variant)
    for (M, K, N) in WREG_SHAPES:
        x = seeded(M, K, seed=40)
        w = seeded(K, N, seed=41) / math.sqrt(K)
        b = 0.1 * seeded(N, seed=42)
        ref = x @ w + b
        out = pkg.linear(ctx, x.cuda(), w.cuda(), b.cuda(), False, 1)
        e = rel_err(out, ref)
        assert e < TOL[1], f"variant {variant} M={M} K={K} N={N}: rel err {e}"


def test_igemm_wreg_exact_and_tile_independent(pkg, ctx, igemm_variant):
    # small-integer operands: every product and every partial sum is exact in f16 x f16 -> f32, so ANY kernel must return the same
    # bits -- a stale fragment register or a tile read before it landed shows up as a wrong integer.  And on general data the three
    # tile heights of the weights-in-registers kernel share one k-summation order (even k-tiles + odd k-tiles): bit-identical.
    g = torch.Generator().manual_seed(3)
    for (M, K, N) in WREG_SHAPES:
        # (two data sets, alternating: what a stale register or LDS tile holds from the previous launch is then WRONG data)
        sets = []
        for _ in range(2):
            xi = torch.randint(-4, 5, (M, K), generator=g).float()
            wi = torch.randint(-3, 4, (K, N), generator=g).float()
            bi = torch.randint(-8, 9, (N,), generator=g).float()
            sets.append((xi.cuda(), wi.cuda(), bi.cuda(), xi @ wi + bi))
        x = seeded(M, K, seed=40)
        w = seeded(K, N, seed=41) / math.sqrt(K)
        outs = []
        for variant in (60, 62, 0):
            igemm_variant(variant)
            for rep in range(4):
                xi, wi, bi, ref = sets[rep & 1]
                o = pkg.linear(ctx, xi, wi, bi, False, 1).cpu()
                assert torch.equal(o, ref), f"variant {variant} M={M} K={K} N={N}: {(o != ref).sum().item()} wrong integers"
            if variant:
                outs.append(pkg.linear(ctx, x.cuda(), w.cuda(), None, False, 1))
        assert torch.equal(outs[0], outs[1]), f"M={M} K={K} N={N}: tile heights differ"


def test_igemm_wreg_row_statistics_match_the_pipe_kernels(pkg, ctx):
    # the statistics a producer leaves for the next folded LayerNorm must not depend on the kernel that produced the rows: the
    # identity projection's output is exact, so the folded-LayerNorm consumer sees bit-identical statistics -- and returns
    # bit-identical results -- with the weights-in-registers kernel (wave-pair exchange of pivot and sums) on or off
    for (M, K, N) in [(300, 1280, 1280), (2048, 640, 640), (96, 128, 256)]:
        x = (seeded(M, K, seed=11) * 1.5 + 0.3).half().float()
        gamma, beta = 1 + 0.1 * seeded(K, seed=5), 0.1 * seeded(K, seed=6)
        w = seeded(K, N, seed=8) / math.sqrt(K)
        outs = []
        for on in (1, 0):
            pkg.debug_set("igemm_wreg", on)
            try:
                outs.append(pkg.layer_norm_linear(ctx, x.cuda(), gamma.cuda(), beta.cuda(), w.cuda(), None, 1e-5, False, 1))
            finally:
                pkg.debug_set("igemm_wreg", 1)
        ref = OM.layer_norm(x, gamma, beta, 1e-5) @ w
        assert rel_err(outs[0], ref) < TOL[1]
        assert torch.equal(outs[0], outs[1]), f"M={M} K={K} N={N}: row statistics differ between the producers"


@pytest.mark.parametrize("mag", [1e5, 3e7, 1e-6])
def test_split_operand_operands_outside_the_f16_range(pkg, ctx, mag):
    # the fp32 stream of a real model is not bounded by 65504 (the SDXL VAE's hidden state is the known case) and may sit far below
    # f16's normal range: the HL16 conversion of such a tensor carries a power-of-two scale (launch_f32_to_hl_scaled) that the GEMM's
    # epilogue undoes -- without it hi = f16(x) is inf (NaN out of the product) or the lo halves lose their bits
    M, K, N = 300, 640, 256
    x = seeded(M, K, seed=7) * mag
    w = seeded(K, N, seed=8) / math.sqrt(K)
    b = 0.1 * seeded(N, seed=9) * mag
    ref = (x.double() @ w.double() + b.double()).float()
    out = pkg.linear(ctx, x.cuda(), w.cuda(), b.cuda(), False, 3)
    assert torch.isfinite(out).all()
    e = rel_err(out, ref)
    print(f"split-operand linear, |x| ~ {mag:g}: rel err {e:.3e}")
    assert e < 4e-6
    xc = seeded(1, 64, 20, 20, seed=43) * mag
    wc = seeded(64, 64, 3, 3, seed=44) / math.sqrt(64 * 9)
    bc = 0.1 * seeded(64, seed=45) * mag
    refc = F.conv2d(xc.double(), wc.double(), bc.double(), padding=1).float()
    outc = pkg.conv2d(ctx, xc.cuda(), wc.cuda(), bc.cuda(), 1, 1, False, 3)
    ec = rel_err(outc, refc)
    print(f"split-operand conv3x3, |x| ~ {mag:g}: rel err {ec:.3e}")
    assert torch.isfinite(outc).all() and ec < 4e-6

