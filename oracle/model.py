"""fp32 torch-CPU restatement of the reference graph -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Each function cites the reference lines it follows (paths relative to /root/reference).  Tensors use the
reference's own layouts: images/latents NCHW, tokens [B,N,C], Linear weight [d_in,d_out]
(y = x @ W + b, burn nn::Linear), conv weight [out,in,kh,kw].  ``W`` is a dict name -> torch tensor whose
names are the reference's struct field paths (oracle/config.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .config import UNetConfig, VAEConfig, unet_block_plan

Tensor = torch.Tensor
EPS = 1e-5  # GroupNormConfig / LayerNormConfig default (groupnorm/mod.rs:13-14, layernorm/mod.rs:12-13)


# ----------------------------------------------------------------------------- arithmetic emulation (round 5)
class Numerics:
    """Which arithmetic the restatement runs in.  The default (mode None) is plain fp32 -- every hook below is the identity and the
    results are bit for bit what they were before the hooks existed.  Two emulations, both computed in fp32 with explicit roundings:

    ``f16ref``   the REFERENCE's own GPU arithmetic: ``LibTorch<f16>`` (src/bin/sample/main.rs:122) holds every tensor -- the f16
                 parameters of the record (``HalfPrecisionSettings``, :37) and the output of EVERY burn op -- in f16; libtorch's half
                 kernels accumulate matmul / conv / reductions in fp32 and round the result once.  So: parameters rounded to f16, and
                 the output of every op of the restated graph rounded to f16, at the granularity the reference issues them (its
                 hand-written layernorm is seven ops, groupnorm/mod.rs:75-82; SiLU two, silu.rs:14-16; nn::Linear matmul + bias add;
                 the LibTorch attention override ONE fused scaled_dot_product_attention, backend.rs:31-80).
    ``operands`` the ENGINE's f16 mode seen from the oracle: only the two operands of the GEMMs of the listed classes are rounded to
                 f16 (fp32 accumulation, fp32 everything else) -- classes: qkv, attn (q, k, v and the probabilities), out (attention
                 out-projections), xattn (cross-attention projections + its attention), geglu, ff, and conv = conv_res (the two 3x3 convs of every
                 ResBlock) + conv_skip (1x1 skip connections) + conv_io (the UNet's first and last conv) + conv_updown + conv_proj (proj_in / proj_out).
    """

    def __init__(self):
        self.mode = None
        self.classes = frozenset()
        self._wcache = {}

    def set(self, mode=None, classes=()):
        assert mode in (None, "f16ref", "operands", "fp8lo")
        self.mode, self.classes = mode, frozenset(classes)
        self._wcache = {}

    @staticmethod
    def _h(x: Tensor) -> Tensor:
        return x.half().float()

    def r(self, x: Tensor) -> Tensor:
        """output of one reference op"""
        return self._h(x) if self.mode == "f16ref" else x

    def on(self, cls: str) -> bool:
        """operands emulation: is GEMM class cls demoted?  ("conv" names all its sub-classes conv_res / conv_skip / conv_io / conv_updown / conv_proj)"""
        return self.mode == "operands" and (cls in self.classes or cls.split("_")[0] in self.classes)

    # ``fp8lo`` (numerics model only; no kernel exists): activation a = hi + lo with hi = f16(a) and lo carried in fp8 e4m3
    # times a per-tensor power of two; weights w16 = f16(w) and, for the lo product, w8 = e4m3(w * 2^k):  a . w ~ hi . w16 + lo8 . w8  --
    # one f16 MFMA + one fp8 MFMA (2x rate), 3 bytes per element instead of the (hi, lo) f16 pair's 4.  The a . w_lo term is NOT there:
    # the model is meant for f16-representable weights (what the reference's records hold).
    @staticmethod
    def _q8(x: Tensor) -> Tensor:
        m = float(x.abs().max())
        if not (m > 0.0):
            return x
        s = 2.0 ** math.floor(math.log2(448.0 / m))       # e4m3 (fn) tops out at 448
        return (x * s).to(torch.float8_e4m3fn).float() / s

    def fp8lo_pair(self, x: Tensor):
        hi = self._h(x)
        return hi, self._q8(x - hi)

    def w(self, W, name: str, cls: str) -> Tensor:
        """a parameter tensor as the arithmetic holds it (cached: the 31-step trajectory asks 62 times)"""
        t = W[name]
        if self.mode == "f16ref" or self.on(cls):
            # keyed by the tensor itself, not by its name alone: a second weight dict with the same names (base and refiner UNet, regenerated
            # weights) inside one session must not receive the first model's rounded tensors
            key = (name, id(t))
            c = self._wcache.get(key)
            if c is None or c[0] is not t:
                c = self._wcache[key] = (t, self._h(t))       # (the source tensor is kept alive: an id cannot be recycled under the cache)
            return c[1]
        return t

    def a(self, x: Tensor, cls: str) -> Tensor:
        """activation operand of a GEMM of class cls"""
        return self._h(x) if self.on(cls) else x


NUM = Numerics()


def to_torch(weights: Dict[str, "np.ndarray"]) -> Dict[str, Tensor]:  # noqa: F821
    return {k: torch.from_numpy(v) for k, v in weights.items()}


# ----------------------------------------------------------------------------- primitives

def silu(x: Tensor) -> Tensor:
    """src/model/silu.rs:14-16: x * sigmoid(x)."""
    return NUM.r(x * NUM.r(torch.sigmoid(x)))


def layernorm_fn(x: Tensor, eps: float) -> Tensor:
    """src/model/groupnorm/mod.rs:75-82 == layernorm/mod.rs:42-49:
    u = x - mean(x, last); u / sqrt(mean(u*u, last) + eps)   (biased variance, eps inside the sqrt)."""
    if NUM.mode != "f16ref":
        u = x - x.mean(dim=-1, keepdim=True)
        return u / torch.sqrt((u * u).mean(dim=-1, keepdim=True) + eps)
    r = NUM.r     # seven ops, each output an f16 tensor: mean_dim, sub, mul, mean_dim, add_scalar, sqrt, div
    u = r(x - r(x.mean(dim=-1, keepdim=True)))
    return r(u / r(torch.sqrt(r(r(r(u * u).mean(dim=-1, keepdim=True)) + eps))))


def group_norm(x: Tensor, gamma: Tensor, beta: Tensor, n_group: int = 32, eps: float = EPS) -> Tensor:
    """GroupNorm::forward src/model/groupnorm/mod.rs:52-73 (x is [B,C,...])."""
    shape = x.shape
    b = shape[0]
    aff = [1] * x.dim()
    aff[1] = shape[1]
    y = layernorm_fn(x.reshape(b, n_group, -1), eps).reshape(shape)
    return NUM.r(NUM.r(y * gamma.reshape(aff)) + beta.reshape(aff))


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = EPS) -> Tensor:
    """LayerNorm::forward src/model/layernorm/mod.rs:34-40."""
    return NUM.r(NUM.r(layernorm_fn(x, eps) * gamma) + beta)


def _eps(W, p: str) -> float:
    """per-norm eps: the reference's .npy loaders read it per module (groupnorm/load.rs:19, layernorm/load.rs:17);
    the parameter enumeration carries it as `<norm>.eps` (default 1e-5 = the Config default the .mpk path gets)."""
    e = W.get(p + ".eps")
    return EPS if e is None else float(e.reshape(-1)[0])


def gn(x: Tensor, W, p: str, n_group: int = 32) -> Tensor:
    return group_norm(x, NUM.w(W, p + ".gamma", "-"), NUM.w(W, p + ".beta", "-"), n_group, _eps(W, p))


def ln(x: Tensor, W, p: str) -> Tensor:
    return layer_norm(x, NUM.w(W, p + ".gamma", "-"), NUM.w(W, p + ".beta", "-"), _eps(W, p))


def linear(x: Tensor, W: Dict[str, Tensor], name: str, cls: str = "-") -> Tensor:
    """burn nn::Linear: y = x @ W[d_in,d_out] (+ b).  cls: GEMM class for the operand-rounding emulation (Numerics)."""
    if NUM.mode == "fp8lo" and cls != "-":
        hi, lo8 = NUM.fp8lo_pair(x)
        w = W[name + ".weight"]
        y = hi @ NUM._h(w) + lo8 @ NUM._q8(w)
    else:
        y = NUM.r(NUM.a(x, cls) @ NUM.w(W, name + ".weight", cls))
    if (name + ".bias") not in W:
        return y
    return NUM.r(y + NUM.w(W, name + ".bias", "-"))


def conv2d(x: Tensor, W: Dict[str, Tensor], name: str, stride: int = 1, padding: int = 0, cls: str = "conv_res") -> Tensor:
    if NUM.mode == "fp8lo":
        hi, lo8 = NUM.fp8lo_pair(x)
        w = W[name + ".weight"]
        return (F.conv2d(hi, NUM._h(w), W[name + ".bias"], stride=stride, padding=padding) +
                F.conv2d(lo8, NUM._q8(w), None, stride=stride, padding=padding))
    return NUM.r(F.conv2d(NUM.a(x, cls), NUM.w(W, name + ".weight", cls), NUM.w(W, name + ".bias", "-"), stride=stride, padding=padding))


def qkv_attention(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor], n_head: int, cls: str = "attn") -> Tensor:
    """Generic path src/backend.rs:88-128: q,k each scaled by d^-0.25, softmax over keys, [B,N,H*d] in/out.
    (f16ref: the LibTorch override, backend.rs:31-80, is ONE fused scaled_dot_product_attention -- fp32 inside, output rounded once;
    operands emulation: q, k, v and the probabilities rounded to f16, as the engine's flash kernel holds them.)"""
    if NUM.mode == "fp8lo":          # (the attention itself stays the f16 flash kernel: its class is 2e-5 of the forward)
        q, k, v = NUM._h(q), NUM._h(k), NUM._h(v)
    q, k, v = NUM.a(q, cls), NUM.a(k, cls), NUM.a(v, cls)
    n_batch, n_qctx, n_state = q.shape
    n_ctx = k.shape[1]
    scale = (n_state / n_head) ** -0.25
    n_hstate = n_state // n_head
    qh = q.reshape(n_batch, n_qctx, n_head, n_hstate).transpose(1, 2) * scale
    kh = (k.reshape(n_batch, n_ctx, n_head, n_hstate).transpose(1, 2).transpose(-1, -2)) * scale
    vh = v.reshape(n_batch, n_ctx, n_head, n_hstate).transpose(1, 2)
    qk = qh @ kh
    if mask is not None:
        qk = qk + mask[:n_qctx, :n_ctx]
    w = torch.softmax(qk, dim=3)
    if NUM.on(cls):
        # the engine keeps P = exp2(s - m) <= 1 unnormalised in f16 and divides the fp32 sums at the end
        mx = w.amax(dim=3, keepdim=True)
        return (NUM.a(w / mx, cls) @ vh * mx).transpose(1, 2).flatten(2, 3)
    return NUM.r((w @ vh).transpose(1, 2).flatten(2, 3))


def attn_decoder_mask(seq_length: int) -> Tensor:
    """src/backend.rs:130-136: -inf strictly above the diagonal, 0 elsewhere."""
    m = torch.zeros(seq_length, seq_length)
    return m.masked_fill(torch.ones(seq_length, seq_length, dtype=torch.bool).triu(1), float("-inf"))


def upsample_nearest2x(x: Tensor) -> Tensor:
    """reshape + repeat(3,2) + repeat(5,2) (unet/mod.rs:744-749, autoencoder/mod.rs:313-318)."""
    b, c, h, w = x.shape
    return x.reshape(b, c, h, 1, w, 1).repeat(1, 1, 1, 2, 1, 2).reshape(b, c, 2 * h, 2 * w)


# ----------------------------------------------------------------------------- UNet

def timestep_embedding(timesteps: Tensor, dim: int, max_period: int = 10000) -> Tensor:
    """src/model/unet/mod.rs:21-39 (cos first, then sin)."""
    half = dim // 2
    freqs = NUM.r(torch.exp(NUM.r(torch.arange(half, dtype=torch.float32) * (-math.log(max_period) / half))))
    args = NUM.r(NUM.r(timesteps.float())[:, None] * freqs[None])
    return torch.cat([NUM.r(args.cos()), NUM.r(args.sin())], dim=1)


def conditioning_embedding(pooled: Tensor, dim: int, size: Tensor, crop: Tensor, ar: Tensor) -> Tensor:
    """src/model/unet/mod.rs:41-57."""
    cat = torch.cat([size, crop, ar], dim=1)
    b, w = cat.shape
    emb = timestep_embedding(cat.reshape(b * w), dim, 10000).reshape(b, w * dim)
    return torch.cat([pooled, emb], dim=1)


def res_block(x: Tensor, emb: Tensor, W, p: str) -> Tensor:
    """ResBlock::forward src/model/unet/mod.rs:1082-1106."""
    h = conv2d(silu(gn(x, W, p + ".norm_in")), W, p + ".conv_in", padding=1)
    e = linear(silu(emb), W, p + ".lin_embed")
    h = NUM.r(h + e[:, :, None, None])
    h = conv2d(silu(gn(h, W, p + ".norm_out")), W, p + ".conv_out", padding=1)
    if (p + ".skip_connection.weight") in W:
        return NUM.r(conv2d(x, W, p + ".skip_connection", cls="conv_skip") + h)
    return NUM.r(x + h)


def multi_head_attention(x: Tensor, context: Optional[Tensor], W, p: str, n_head: int) -> Tensor:
    """MultiHeadAttention::forward src/model/unet/mod.rs:1005-1023."""
    xa = x if context is None else context
    cq, ca = ("qkv", "attn") if context is None else ("xattn", "xattn")
    q = linear(x, W, p + ".query", cq)
    k = linear(xa, W, p + ".key", cq)
    v = linear(xa, W, p + ".value", cq)
    return linear(qkv_attention(q, k, v, None, n_head, ca), W, p + ".out", "out")


def geglu(x: Tensor, W, p: str) -> Tensor:
    """GEGLU::forward src/model/unet/mod.rs:942-956 (burn nn::Gelu = exact erf GELU)."""
    pr = linear(x, W, p + ".proj", "geglu")
    n = pr.shape[-1] // 2
    return NUM.r(pr[..., :n] * NUM.r(F.gelu(pr[..., n:])))


def transformer_block(x: Tensor, context: Tensor, W, p: str, n_head: int) -> Tensor:
    """TransformerBlock::forward src/model/unet/mod.rs:885-891."""
    x = NUM.r(x + multi_head_attention(ln(x, W, p + ".norm1"), None, W, p + ".attn1", n_head))
    x = NUM.r(x + multi_head_attention(ln(x, W, p + ".norm2"), context, W, p + ".attn2", n_head))
    h = ln(x, W, p + ".norm3")
    return NUM.r(x + linear(geglu(h, W, p + ".mlp.geglu"), W, p + ".mlp.lin", "ff"))   # MLP::forward :915-918


def spatial_transformer(x: Tensor, context: Tensor, W, p: str, n_head: int, depth: int) -> Tensor:
    """SpatialTransformer::forward src/model/unet/mod.rs:820-845."""
    b, c, h, w = x.shape
    x_in = x
    t = gn(x, W, p + ".norm").reshape(b, c, h * w).transpose(1, 2)
    t = linear(t, W, p + ".proj_in", "conv_proj")
    for j in range(depth):
        t = transformer_block(t, context, W, f"{p}.blocks.{j}", n_head)
    t = linear(t, W, p + ".proj_out", "conv_proj").transpose(1, 2).reshape(b, c, h, w)
    return NUM.r(x_in + t)


def _unet_block(x, emb, ctx, W, p, b):
    k = b["kind"]
    if k == "Conv":
        return conv2d(x, W, p, padding=1, cls="conv_io")
    if k == "Down":                                  # DownsampleConfig::init unet/mod.rs:765-772
        return conv2d(x, W, p, stride=2, padding=1, cls="conv_updown")
    if k == "Res":
        return res_block(x, emb, W, p)
    x = res_block(x, emb, W, p + ".res")
    if k in ("ResT", "ResTU"):                       # :571-577, :657-664
        x = spatial_transformer(x, ctx, W, p + ".transformer", b["n_head"], b["depth"])
    if k in ("ResTU", "ResU"):                       # Upsample::forward :742-752
        x = conv2d(upsample_nearest2x(x), W, p + ".upsample.conv", padding=1, cls="conv_updown")
    return x


def unet_forward(cfg: UNetConfig, W, x: Tensor, timesteps: Tensor, context: Tensor, label: Tensor) -> Tensor:
    """UNet::forward src/model/unet/mod.rs:450-492.  x [B,4,H,W], timesteps [B] int, context [B,77,ctx], label [B,adm]."""
    inp, mid, out = unet_block_plan(cfg)
    x, context, label = NUM.r(x), NUM.r(context), NUM.r(label)   # f16ref: Conditioning::convert hands the f16 backend f16 tensors (stablediffusion/mod.rs:559-580)
    t_emb = timestep_embedding(timesteps, cfg.model_channels, 10000)
    t_emb = linear(silu(linear(t_emb, W, "lin1_time_embed")), W, "lin2_time_embed")
    l_emb = linear(silu(linear(label, W, "lin1_label_embed")), W, "lin2_label_embed")
    emb = NUM.r(t_emb + l_emb)
    saved = []
    for i, b in enumerate(inp):
        x = _unet_block(x, emb, context, W, f"input_blocks.{i}", b)
        saved.append(x)
    x = res_block(x, emb, W, "middle_block.res1")                                   # :713-719
    x = spatial_transformer(x, context, W, "middle_block.transformer", mid["n_head"], mid["depth"])
    x = res_block(x, emb, W, "middle_block.res2")
    for i, b in enumerate(out):
        x = torch.cat([x, saved.pop()], dim=1)                                      # :484
        x = _unet_block(x, emb, context, W, f"output_blocks.{i}", b)
    x = silu(gn(x, W, "norm_out"))
    return conv2d(x, W, "conv_out", padding=1, cls="conv_io")


# ----------------------------------------------------------------------------- VAE

def vae_resnet_block(x: Tensor, W, p: str) -> Tensor:
    """ResnetBlock::forward src/model/autoencoder/mod.rs:500-516."""
    h = conv2d(silu(gn(x, W, p + ".norm1")), W, p + ".conv1", padding=1)
    h = conv2d(silu(gn(h, W, p + ".norm2")), W, p + ".conv2", padding=1)
    if (p + ".nin_shortcut.weight") in W:
        return conv2d(x, W, p + ".nin_shortcut") + h
    return x + h


def vae_attn_block(x: Tensor, W, p: str) -> Tensor:
    """ConvSelfAttentionBlock::forward src/model/autoencoder/mod.rs:550-586 (1 head, 1x1-conv q/k/v)."""
    b, c, hh, ww = x.shape
    h = gn(x, W, p + ".norm")
    q = conv2d(h, W, p + ".q").reshape(b, c, hh * ww).transpose(1, 2)
    k = conv2d(h, W, p + ".k").reshape(b, c, hh * ww).transpose(1, 2)
    v = conv2d(h, W, p + ".v").reshape(b, c, hh * ww).transpose(1, 2)
    wv = qkv_attention(q, k, v, None, 1).transpose(1, 2).reshape(b, c, hh, ww)
    return x + conv2d(wv, W, p + ".proj_out")


def vae_mid(x: Tensor, W, p: str) -> Tensor:
    """Mid::forward src/model/autoencoder/mod.rs:443-449."""
    x = vae_resnet_block(x, W, p + ".block_1")
    x = vae_attn_block(x, W, p + ".attn")
    return vae_resnet_block(x, W, p + ".block_2")


def vae_decoder_forward(cfg: VAEConfig, W, x: Tensor) -> Tensor:
    """Decoder::forward src/model/autoencoder/mod.rs:203-216; DecoderBlock::forward :306-324."""
    x = conv2d(x, W, "decoder.conv_in", padding=1)
    x = vae_mid(x, W, "decoder.mid")
    n = len(cfg.dec_channels)
    for i in range(n):
        p = f"decoder.blocks.{i}"
        for r in ("res1", "res2", "res3"):
            x = vae_resnet_block(x, W, f"{p}.{r}")
        if i != n - 1:
            x = conv2d(upsample_nearest2x(x), W, p + ".upsampler", padding=1)
    x = silu(gn(x, W, "decoder.norm_out"))
    return conv2d(x, W, "decoder.conv_out", padding=1)


def decode_latent_vae(cfg: VAEConfig, W, latent: Tensor) -> Tensor:
    """Autoencoder::decode_latent src/model/autoencoder/mod.rs:67-70."""
    return vae_decoder_forward(cfg, W, conv2d(latent, W, "post_quant_conv"))


def padded_conv2d_s2(x: Tensor, W, name: str) -> Tensor:
    """PaddedConv2d with Padding(left 0,right 1,top 0,bottom 1), k=3, stride 2
    (autoencoder/mod.rs:229-238, 334-407): symmetric pad 2 then slice from skip=1 -- i.e. the same taps as
    an asymmetric (0,1,0,1) zero-pad followed by a valid stride-2 conv."""
    b, c, h, w = x.shape
    pad_actual = 2                     # calc_padding(0,1): ceil((1-0)/2)*2 + 0 = 2
    desired_h = (0 + 1 + h - 3) // 2 + 1
    desired_w = (0 + 1 + w - 3) // 2 + 1
    skip = (pad_actual - 0) // 2
    y = F.conv2d(x, W[name + ".weight"], W[name + ".bias"], stride=2, padding=pad_actual)
    return y[:, :, skip:skip + desired_h, skip:skip + desired_w]


def vae_encoder_forward(cfg: VAEConfig, W, x: Tensor) -> Tensor:
    """Encoder::forward src/model/autoencoder/mod.rs:131-144; EncoderBlock::forward :258-268."""
    x = conv2d(x, W, "encoder.conv_in", padding=1)
    n = len(cfg.enc_channels)
    for i in range(n):
        p = f"encoder.blocks.{i}"
        x = vae_resnet_block(x, W, p + ".res1")
        x = vae_resnet_block(x, W, p + ".res2")
        if i != n - 1:
            x = padded_conv2d_s2(x, W, p + ".downsampler")
    x = vae_mid(x, W, "encoder.mid")
    x = silu(gn(x, W, "encoder.norm_out"))
    return conv2d(x, W, "encoder.conv_out", padding=1)


def encode_image_vae(cfg: VAEConfig, W, x: Tensor) -> Tensor:
    """Autoencoder::encode_image src/model/autoencoder/mod.rs:59-65 (channels 0..4 = the mean)."""
    return conv2d(vae_encoder_forward(cfg, W, x), W, "quant_conv")[:, 0:4]
