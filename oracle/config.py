"""Configs + canonical parameter enumeration + seeded synthetic weights (oracle side).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Mirrors:
  * ``UNetConfig`` / ``UNetConfig::init``            /root/reference/src/model/unet/mod.rs:59-430
  * ``DiffuserConfig`` fields                         /root/reference/src/model/stablediffusion/mod.rs:269-305
  * ``AutoencoderConfig::init``                       /root/reference/src/model/autoencoder/mod.rs:27-44
  * tensor conventions (Linear weight is [d_in,d_out], conv weight [out,in,kh,kw])
                                                      /root/reference/python/save.py:20-25,56-72, src/model/load.rs:62-74,119-156

The parameter *names* follow the reference's struct field names; the enumeration order is the order the
C-ABI's ``sdxl_*_param_spec`` reports (tests check the two enumerations against each other).
No real SDXL weights exist on this box (no network), so weights are synthetic: a counter-based integer hash
(FNV-1a of the parameter name, splitmix64 finaliser) -> uniform value with a prescribed std.  The recipe is
integer-exact and uses one fp32 multiply + one fp32 add, so numpy (here) and the HIP fill kernel
(csrc/weights.hip) produce bit-identical tensors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

# ----------------------------------------------------------------------------- configs


@dataclass
class UNetConfig:
    """Fields of the reference's UNetConfig (unet/mod.rs:59-69) + DiffuserConfig.is_refiner."""
    adm_in_channels: int
    model_channels: int
    channel_mults: List[int]
    n_head_channels: int
    transformer_depths: List[int]
    context_dim: int
    in_channels: int = 4
    out_channels: int = 4
    is_refiner: bool = False


def sdxl_base_config() -> UNetConfig:
    # implied .cfg values: unet/mod.rs:92-111 comment, python/unet.py:132-160
    return UNetConfig(2816, 320, [1, 2, 4], 64, [0, 2, 10], 2048)


def sdxl_refiner_config() -> UNetConfig:
    # python/unet.py:163-200
    return UNetConfig(2560, 384, [1, 2, 4, 4], 64, [0, 4, 4, 4], 1280, is_refiner=True)


def tiny_config() -> UNetConfig:
    """Small arch of the same family (same flavour as bin/test's tiny probes, test/main.rs:128-140)."""
    return UNetConfig(128, 64, [1, 2, 4], 64, [0, 1, 2], 128)


def tiny_refiner_config() -> UNetConfig:
    return UNetConfig(96, 64, [1, 2, 4, 4], 64, [0, 1, 1, 1], 64, is_refiner=True)


@dataclass
class VAEConfig:
    """AutoencoderConfig::init hard-codes these (autoencoder/mod.rs:27-35)."""
    enc_channels: List[Tuple[int, int]] = field(
        default_factory=lambda: [(128, 128), (128, 256), (256, 512), (512, 512)])
    dec_channels: List[Tuple[int, int]] = field(
        default_factory=lambda: [(512, 512), (512, 512), (512, 256), (256, 128)])
    n_group: int = 32
    enc_out_channels: int = 8
    scale_factor: float = 0.13025  # python/dump.py:37


def sdxl_vae_config() -> VAEConfig:
    return VAEConfig()


def tiny_vae_config() -> VAEConfig:
    return VAEConfig(enc_channels=[(32, 32), (32, 64), (64, 64), (64, 64)],
                     dec_channels=[(64, 64), (64, 64), (64, 32), (32, 32)])


# ----------------------------------------------------------------------------- parameter specs

KIND_LINEAR_W = 0   # [d_in, d_out]   (burn nn::Linear, python/save.py:23)
KIND_CONV_W = 1     # [out, in, kh, kw]
KIND_BIAS = 2
KIND_GAMMA = 3
KIND_BETA = 4
KIND_EPS = 5        # [1] scalar: the norm's eps
DEFAULT_EPS = 1e-5  # GroupNormConfig / LayerNormConfig default (groupnorm/mod.rs:13-14, layernorm/mod.rs:12-13)


@dataclass
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    kind: int
    scale: np.float32   # value = (u - 0.5) * scale + mean,  u uniform in [0,1) on a 2^-24 grid
    mean: np.float32

    @property
    def numel(self) -> int:
        n = 1
        for d in self.shape:
            n *= d
        return n


_SQRT12 = math.sqrt(12.0)


def _wscale(fan_in: int, gain: float) -> np.float32:
    return np.float32(_SQRT12 * gain / math.sqrt(float(fan_in)))


class _Spec:
    def __init__(self):
        self.items: List[ParamSpec] = []

    def linear(self, name, d_in, d_out, bias=True, gain=1.0):
        self.items.append(ParamSpec(name + ".weight", (d_in, d_out), KIND_LINEAR_W, _wscale(d_in, gain), np.float32(0)))
        if bias:
            self.items.append(ParamSpec(name + ".bias", (d_out,), KIND_BIAS, np.float32(_SQRT12 * 0.02), np.float32(0)))

    def conv(self, name, c_in, c_out, k, gain=1.0):
        self.items.append(ParamSpec(name + ".weight", (c_out, c_in, k, k), KIND_CONV_W, _wscale(c_in * k * k, gain), np.float32(0)))
        self.items.append(ParamSpec(name + ".bias", (c_out,), KIND_BIAS, np.float32(_SQRT12 * 0.02), np.float32(0)))

    def norm(self, name, c):
        self.items.append(ParamSpec(name + ".gamma", (c,), KIND_GAMMA, np.float32(_SQRT12 * 0.02), np.float32(1)))
        self.items.append(ParamSpec(name + ".beta", (c,), KIND_BETA, np.float32(_SQRT12 * 0.02), np.float32(0)))
        # per-module eps, read per norm by the reference's .npy loaders (groupnorm/load.rs:19, layernorm/load.rs:17);
        # synthetic value = the Config default 1e-5 exactly (scale 0)
        self.items.append(ParamSpec(name + ".eps", (1,), KIND_EPS, np.float32(0), np.float32(DEFAULT_EPS)))


RES_GAIN = 0.5  # residual-branch output layers: keeps the residual stream's variance moderate


def _res_block(s: _Spec, p, c_in, c_emb, c_out):
    # ResBlockConfig::init unet/mod.rs:1032-1067
    s.norm(p + ".norm_in", c_in)
    s.conv(p + ".conv_in", c_in, c_out, 3)
    s.linear(p + ".lin_embed", c_emb, c_out)
    s.norm(p + ".norm_out", c_out)
    s.conv(p + ".conv_out", c_out, c_out, 3, gain=RES_GAIN)
    if c_in != c_out:
        s.conv(p + ".skip_connection", c_in, c_out, 1)


def _mha(s: _Spec, p, n_state, n_ctx_state):
    # MultiHeadAttentionConfig::init unet/mod.rs:965-994 (q,k,v without bias)
    s.linear(p + ".query", n_state, n_state, bias=False)
    s.linear(p + ".key", n_ctx_state, n_state, bias=False)
    s.linear(p + ".value", n_ctx_state, n_state, bias=False)
    s.linear(p + ".out", n_state, n_state, gain=RES_GAIN)


def _transformer(s: _Spec, p, c, ctx, depth):
    # SpatialTransformerConfig::init unet/mod.rs:790-810; TransformerBlockConfig::init :854-873
    s.norm(p + ".norm", c)
    s.linear(p + ".proj_in", c, c)
    for j in range(depth):
        q = f"{p}.blocks.{j}"
        s.norm(q + ".norm1", c)
        _mha(s, q + ".attn1", c, c)
        s.norm(q + ".norm2", c)
        _mha(s, q + ".attn2", c, ctx)
        s.norm(q + ".norm3", c)
        s.linear(q + ".mlp.geglu.proj", c, 2 * 4 * c)       # GEGLUConfig::init :927-934
        s.linear(q + ".mlp.lin", 4 * c, c, gain=RES_GAIN)   # MLPConfig::init :899-907
    s.linear(p + ".proj_out", c, c, gain=RES_GAIN)


def unet_block_plan(cfg: UNetConfig):
    """Block lists exactly as UNetConfig::init builds them (unet/mod.rs:115-173, 238-248, 250-328).

    Returns (input_blocks, middle, output_blocks); each block is a dict with 'kind' in
    {Conv, Res, Down, ResT, ResTU, ResU} and its channel / head / depth fields.
    """
    mc = cfg.model_channels
    mults = cfg.channel_mults
    n_levels = len(mults)
    emb = 4 * mc
    nh = lambda ch: ch // cfg.n_head_channels  # noqa: E731
    assert mc % cfg.n_head_channels == 0      # unet/mod.rs:73-76

    inp = [dict(kind="Conv", c_in=cfg.in_channels, c_out=mc)]
    for level in range(n_levels):
        c_in = mults[max(level - 1, 0)] * mc
        c_out = mults[level] * mc
        if level != 1 and level != 2:
            inp.append(dict(kind="Res", c_in=c_in, c_emb=emb, c_out=c_out))
            inp.append(dict(kind="Res", c_in=c_out, c_emb=emb, c_out=c_out))
        else:
            d = cfg.transformer_depths[level]
            inp.append(dict(kind="ResT", c_in=c_in, c_emb=emb, c_out=c_out, n_head=nh(c_out), depth=d))
            inp.append(dict(kind="ResT", c_in=c_out, c_emb=emb, c_out=c_out, n_head=nh(c_out), depth=d))
        if level != n_levels - 1:
            inp.append(dict(kind="Down", c=c_out))

    c_mid = mults[-1] * mc
    # NB: the reference passes n_channels_embed = channels_in_middle here (unet/mod.rs:240-247); for every
    # shipped config 4*model_channels == mults[-1]*model_channels, and emb fed at run time is 4*mc wide.
    mid = dict(kind="ResTRes", c=c_mid, c_emb=c_mid, n_head=nh(c_mid), depth=cfg.transformer_depths[-1])

    out = []
    for level in reversed(range(n_levels)):
        nxt = level + 1 if level != n_levels - 1 else level
        c_out = mults[level] * mc
        in1 = mults[nxt] * mc + c_out
        in2 = 2 * c_out
        in3 = c_out + mults[max(level - 1, 0)] * mc
        if level != 1 and level != 2:
            out.append(dict(kind="Res", c_in=in1, c_emb=emb, c_out=c_out))
            out.append(dict(kind="Res", c_in=in2, c_emb=emb, c_out=c_out))
            out.append(dict(kind="ResU" if level != 0 else "Res", c_in=in3, c_emb=emb, c_out=c_out))
        else:
            d = cfg.transformer_depths[level]
            out.append(dict(kind="ResT", c_in=in1, c_emb=emb, c_out=c_out, n_head=nh(c_out), depth=d))
            out.append(dict(kind="ResT", c_in=in2, c_emb=emb, c_out=c_out, n_head=nh(c_out), depth=d))
            out.append(dict(kind="ResTU", c_in=in3, c_emb=emb, c_out=c_out, n_head=nh(c_out), depth=d))
    return inp, mid, out


def _block_params(s: _Spec, p, b, ctx):
    k = b["kind"]
    if k == "Conv":
        s.conv(p, b["c_in"], b["c_out"], 3)
    elif k == "Down":
        s.conv(p, b["c"], b["c"], 3)
    elif k == "Res":
        _res_block(s, p, b["c_in"], b["c_emb"], b["c_out"])
    elif k in ("ResT", "ResTU"):
        _res_block(s, p + ".res", b["c_in"], b["c_emb"], b["c_out"])
        _transformer(s, p + ".transformer", b["c_out"], ctx, b["depth"])
        if k == "ResTU":
            s.conv(p + ".upsample.conv", b["c_out"], b["c_out"], 3)
    elif k == "ResU":
        _res_block(s, p + ".res", b["c_in"], b["c_emb"], b["c_out"])
        s.conv(p + ".upsample.conv", b["c_out"], b["c_out"], 3)
    else:
        raise ValueError(k)


def unet_param_specs(cfg: UNetConfig) -> List[ParamSpec]:
    s = _Spec()
    mc = cfg.model_channels
    emb = 4 * mc
    s.linear("lin1_time_embed", mc, emb)
    s.linear("lin2_time_embed", emb, emb)
    s.linear("lin1_label_embed", cfg.adm_in_channels, emb)
    s.linear("lin2_label_embed", emb, emb)
    inp, mid, out = unet_block_plan(cfg)
    for i, b in enumerate(inp):
        _block_params(s, f"input_blocks.{i}", b, cfg.context_dim)
    _res_block(s, "middle_block.res1", mid["c"], mid["c_emb"], mid["c"])
    _transformer(s, "middle_block.transformer", mid["c"], cfg.context_dim, mid["depth"])
    _res_block(s, "middle_block.res2", mid["c"], mid["c_emb"], mid["c"])
    for i, b in enumerate(out):
        _block_params(s, f"output_blocks.{i}", b, cfg.context_dim)
    s.norm("norm_out", mc)
    s.conv("conv_out", mc, cfg.out_channels, 3)
    return s.items


def _vae_resnet(s: _Spec, p, c_in, c_out):
    # ResnetBlockConfig::init autoencoder/mod.rs:457-490
    s.norm(p + ".norm1", c_in)
    s.conv(p + ".conv1", c_in, c_out, 3)
    s.norm(p + ".norm2", c_out)
    s.conv(p + ".conv2", c_out, c_out, 3, gain=RES_GAIN)
    if c_in != c_out:
        s.conv(p + ".nin_shortcut", c_in, c_out, 1)


def _vae_mid(s: _Spec, p, c):
    # MidConfig::init autoencoder/mod.rs:420-433; ConvSelfAttentionBlockConfig::init :523-540
    _vae_resnet(s, p + ".block_1", c, c)
    s.norm(p + ".attn.norm", c)
    for n in ("q", "k", "v"):
        s.conv(f"{p}.attn.{n}", c, c, 1)
    s.conv(p + ".attn.proj_out", c, c, 1, gain=RES_GAIN)
    _vae_resnet(s, p + ".block_2", c, c)


def vae_decoder_param_specs(cfg: VAEConfig) -> List[ParamSpec]:
    """post_quant_conv + Decoder (autoencoder/mod.rs:35,152-191,274-303)."""
    s = _Spec()
    s.conv("post_quant_conv", 4, 4, 1)
    c0 = cfg.dec_channels[0][0]
    s.conv("decoder.conv_in", 4, c0, 3)
    _vae_mid(s, "decoder.mid", c0)
    for i, (ci, co) in enumerate(cfg.dec_channels):
        p = f"decoder.blocks.{i}"
        _vae_resnet(s, p + ".res1", ci, co)
        _vae_resnet(s, p + ".res2", co, co)
        _vae_resnet(s, p + ".res3", co, co)
        if i != len(cfg.dec_channels) - 1:
            s.conv(p + ".upsampler", co, co, 3)
    cl = cfg.dec_channels[-1][1]
    s.norm("decoder.norm_out", cl)
    s.conv("decoder.conv_out", cl, 3, 3)
    return s.items


def vae_encoder_param_specs(cfg: VAEConfig) -> List[ParamSpec]:
    """Encoder + quant_conv (autoencoder/mod.rs:34,79-129,227-256)."""
    s = _Spec()
    c0 = cfg.enc_channels[0][1]
    s.conv("encoder.conv_in", 3, c0, 3)
    for i, (ci, co) in enumerate(cfg.enc_channels):
        p = f"encoder.blocks.{i}"
        _vae_resnet(s, p + ".res1", ci, co)
        _vae_resnet(s, p + ".res2", co, co)
        if i != len(cfg.enc_channels) - 1:
            s.conv(p + ".downsampler", co, co, 3)
    cl = cfg.enc_channels[-1][0]
    _vae_mid(s, "encoder.mid", cl)
    s.norm("encoder.norm_out", cl)
    s.conv("encoder.conv_out", cl, cfg.enc_out_channels, 3)
    s.conv("quant_conv", cfg.enc_out_channels, cfg.enc_out_channels, 1)
    return s.items


# ----------------------------------------------------------------------------- synthetic values

_MASK = (1 << 64) - 1


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & _MASK
    return h


def synth_values(name: str, numel: int, scale: np.float32, mean: np.float32, seed: int) -> np.ndarray:
    """value[i] = (u_i - 0.5) * scale + mean with u_i = top-24-bits(splitmix64(key + i*G)) * 2^-24."""
    key = np.uint64(fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & _MASK))
    out = np.empty(numel, dtype=np.float32)
    chunk = 1 << 24
    with np.errstate(over="ignore"):
        for s in range(0, numel, chunk):
            e = min(numel, s + chunk)
            z = key + np.arange(s, e, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            u = (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
            out[s:e] = (u - np.float32(0.5)) * np.float32(scale) + np.float32(mean)
    return out


def synth_weights(specs: List[ParamSpec], seed: int = 0) -> Dict[str, np.ndarray]:
    return {p.name: synth_values(p.name, p.numel, p.scale, p.mean, seed).reshape(p.shape) for p in specs}


def alphas_cumprod(n: int = 1000) -> np.ndarray:
    """sgm LegacyDDPMDiscretization (python/dump.py:29-31): beta = linspace(sqrt(.00085), sqrt(.012), n)^2."""
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, n, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas, axis=0).astype(np.float32)


def arb_tensor(*dims: int) -> np.ndarray:
    """Reference's deterministic probe input (src/bin/test/main.rs:51-54, python/dump.py:17-19)."""
    n = int(np.prod(dims))
    return np.sin(np.arange(n, dtype=np.float32)).astype(np.float32).reshape(dims)
