"""stable-diffusion-xl-burn_amd -- MI355X-native SDXL sampling engine, host-side mirror of the reference API.

The product is ``lib/libsdxl_mi355.so`` (hand-written HIP kernels for gfx950 + a C++ engine behind the C ABI of
``include/sdxl_mi355.h``).  This module is the thin ctypes binding the test-suite and ``bench.py`` drive -- the same
symbols a Rust ``cc``+``bindgen`` shim would bind -- exposed under the reference's own names
(``Diffuser.sample_latent`` / ``UNet.forward`` / ``LatentDecoder.latent_to_image`` / ``Conditioning`` / ``RawImages`` /
``qkv_attention``; reference src/model/stablediffusion/mod.rs, src/model/unet/mod.rs, src/backend.rs).

PyTorch appears here only as plumbing (device buffers, streams, torch.distributed); no arithmetic of the hot path runs
in torch, and there is NO fallback: if the HIP library is missing or no GPU is visible every entry point raises.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDXL_MEASURE_LIB=1 (tools/ only): the measurement build (`build.py --measure`: A/B partners, measurement modes, timeline stamps)
LIB_PATH = os.path.join(_HERE, "lib", "libsdxl_mi355_measure.so" if os.environ.get("SDXL_MEASURE_LIB") == "1" else "libsdxl_mi355.so")
if os.environ.get("SDXL_LIB_PATH"):      # tools/ only: an explicitly named build of the same library (A/B of two commits on one box)
    LIB_PATH = os.environ["SDXL_LIB_PATH"]

DTYPE_F32 = 0        # strict parity: fp32 storage + exact fp32 MFMA
DTYPE_F16 = 1        # fp16 storage / MFMA operands, fp32 accumulate
DTYPE_F16_F32RES = 2  # fp16 MFMA operands, fp32 residual stream
DTYPE_F32_SPLIT_MIX_F16W = 5  # DTYPE_F32_SPLIT_MIX for f16-representable parameters (the reference's records): + QKV projection, both out-projections, FF-out, cross-attention query projection on f16 operands, LayerNorms through an f16 shadow (csrc/capi.hip mix_of); falls back to _MIX's classes on other parameters (UNet.mix_classes())
DTYPE_F32_SPLIT_MIX_F16W_GEGLU2 = 6  # DTYPE_F32_SPLIT_MIX_F16W with the GEGLU projection's activations as (hi, lo) f16 pairs along a doubled K: ~12 % slower, inside the scaled bound on every fixture incl. the 4-step stress one
DTYPE_F32_SPLIT_F16W = 7  # DTYPE_F32_SPLIT for f16-representable parameters: same fp32-class arithmetic, the transformer's linear layers on the f16 kernels (HL16 rows read as f16 rows of twice the width), fused split-precision cross-attention; falls back to DTYPE_F32_SPLIT on other parameters
DTYPE_F32_SPLIT_MIX = 4  # UNet / Diffuser: DTYPE_F32_SPLIT with the self-attention and the GEGLU projection on plain f16 operands (the two classes the measured precision frontier affords)
DTYPE_F32_SPLIT = 3   # fp32-class arithmetic on the f16 matrix pipe (operands as (hi, lo) f16 pairs, 3 MFMAs per product): UNet / Diffuser / LatentDecoder and the conv2d / linear / qkv_attention operators

_c_p = ctypes.c_void_p
_f_p = ctypes.c_void_p   # device pointers travel as integers


SEED_F16_WEIGHTS = 1 << 63   # include/sdxl_mi355.h SDXL_SEED_F16_WEIGHTS: synthetic parameters rounded to f16 (what a real record holds)


class EngineError(RuntimeError):
    pass


class _UNetConfigC(ctypes.Structure):
    _fields_ = [("adm_in_channels", ctypes.c_int32), ("in_channels", ctypes.c_int32), ("out_channels", ctypes.c_int32),
                ("model_channels", ctypes.c_int32), ("n_levels", ctypes.c_int32), ("channel_mults", ctypes.c_int32 * 8),
                ("n_head_channels", ctypes.c_int32), ("transformer_depths", ctypes.c_int32 * 8),
                ("context_dim", ctypes.c_int32), ("is_refiner", ctypes.c_int32)]


class _VaeConfigC(ctypes.Structure):
    _fields_ = [("n_blocks", ctypes.c_int32), ("enc_in", ctypes.c_int32 * 8), ("enc_out", ctypes.c_int32 * 8),
                ("dec_in", ctypes.c_int32 * 8), ("dec_out", ctypes.c_int32 * 8), ("n_group", ctypes.c_int32),
                ("enc_out_channels", ctypes.c_int32), ("scale_factor", ctypes.c_double)]


class _ClipConfigC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("n_vocab", "n_state", "embed_dim", "n_head", "n_ctx", "n_layer", "quick_gelu")]


class _ConditioningC(ctypes.Structure):
    _fields_ = [("unconditional_context_full", _f_p), ("unconditional_context_open_clip", _f_p),
                ("context_full", _f_p), ("context_open_clip", _f_p), ("unconditional_channel_context", _f_p),
                ("unconditional_channel_context_refiner", _f_p), ("channel_context", _f_p),
                ("channel_context_refiner", _f_p), ("n", ctypes.c_int32), ("n_ctx", ctypes.c_int32),
                ("height", ctypes.c_int32), ("width", ctypes.c_int32)]


# every symbol include/sdxl_mi355.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "sdxl_last_error", "sdxl_build_info", "sdxl_ctx_create", "sdxl_ctx_destroy", "sdxl_ctx_synchronize",
    "sdxl_unet_config_base", "sdxl_unet_config_refiner", "sdxl_vae_config_default",
    "sdxl_unet_param_count", "sdxl_unet_param_spec", "sdxl_vae_param_count", "sdxl_vae_param_spec",
    "sdxl_unet_create", "sdxl_unet_create_synthetic", "sdxl_unet_destroy", "sdxl_unet_forward", "sdxl_unet_set_graph", "sdxl_unet_set_split_cfg", "sdxl_unet_set_fused_cross_attention", "sdxl_unet_set_gn_from_producer", "sdxl_unet_mix_classes",
    "sdxl_qkv_attention", "sdxl_attn_decoder_mask",
    "sdxl_diffuser_create", "sdxl_diffuser_create_synthetic", "sdxl_diffuser_destroy", "sdxl_diffuser_unet",
    "sdxl_sample_latent", "sdxl_sample_latent_with_inpainting", "sdxl_refine_latent", "sdxl_step_count",
    "sdxl_diffuser_enable_step_timing", "sdxl_diffuser_step_times", "sdxl_diffuser_set_trace",
    "sdxl_vae_create", "sdxl_vae_create_synthetic", "sdxl_vae_destroy", "sdxl_vae_decode_latent",
    "sdxl_latent_to_image", "sdxl_vae_encode_image", "sdxl_image_to_latent",
    "sdxl_unet_weight_arena", "sdxl_vae_weight_arena", "sdxl_diffuser_create_empty", "sdxl_vae_create_empty",
    "sdxl_unet_profile", "sdxl_unet_eager_forward_ms", "sdxl_bench_igemm", "sdxl_bench_attention", "sdxl_debug_set", "sdxl_debug_warm_schedule",
    "sdxl_group_norm", "sdxl_layer_norm", "sdxl_conv2d", "sdxl_linear", "sdxl_layer_norm_linear", "sdxl_ln_query_cross_attention", "sdxl_conv2d_group_norm",
    "sdxl_clip_config_clip_l", "sdxl_clip_config_open_clip_bigg", "sdxl_clip_param_count", "sdxl_clip_param_spec",
    "sdxl_clip_create", "sdxl_clip_create_synthetic", "sdxl_clip_destroy", "sdxl_clip_forward_hidden",
    "sdxl_clip_forward_hidden_pooled", "sdxl_conditioning_embedding", "sdxl_clip_weight_arena",
    "sdxl_unet_create_f16", "sdxl_diffuser_create_f16", "sdxl_vae_create_f16", "sdxl_clip_create_f16",
    "sdxl_comm_unique_id", "sdxl_comm_create", "sdxl_comm_destroy", "sdxl_bcast_buffer", "sdxl_unet_bcast_weights",
    "sdxl_vae_bcast_weights", "sdxl_clip_bcast_weights", "sdxl_bcast_plan",
]

_lib = None


def lib() -> ctypes.CDLL:
    """Loads the HIP engine; fails loudly when it has not been built (no CPU / eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(f"{LIB_PATH} is missing: run `python stable-diffusion-xl-burn_amd/build.py` "
                              "(hipcc --offload-arch=gfx950). There is no fallback path.")
        # torch's ROCm wheel bundles its own libamdhip64: it must be in the process BEFORE this library is loaded, so both
        # bind to ONE HIP runtime.  Loaded the other way round (library first, torch later) the process ends up with two
        # runtimes and the second one reports "no ROCm-capable device" (seen with build() followed by smoke() in one process).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = ctypes.CDLL(LIB_PATH)
        l.sdxl_last_error.restype = ctypes.c_char_p
        l.sdxl_build_info.restype = ctypes.c_char_p
        l.sdxl_diffuser_unet.restype = ctypes.c_void_p
        l.sdxl_diffuser_unet.argtypes = [ctypes.c_void_p]
        for name in ("sdxl_ctx_destroy", "sdxl_unet_destroy", "sdxl_diffuser_destroy", "sdxl_vae_destroy", "sdxl_clip_destroy",
                     "sdxl_comm_destroy"):
            getattr(l, name).restype = None
            getattr(l, name).argtypes = [ctypes.c_void_p]
        _lib = l
    return _lib


def _check(rc: int):
    if rc != 0:
        raise EngineError(lib().sdxl_last_error().decode())


def _torch():
    import torch
    return torch


def _dev(t, dtype=None):
    """contiguous CUDA tensor of the expected dtype -> (tensor kept alive, device pointer)"""
    torch = _torch()
    if dtype is None:
        dtype = torch.float32
    if not t.is_cuda:
        raise EngineError("expected a CUDA (ROCm) tensor: the engine has no CPU path")
    t = t.to(dtype).contiguous()
    return t, ctypes.c_void_p(t.data_ptr())


def _stream() -> ctypes.c_void_p:
    s = _torch().cuda.current_stream().cuda_stream
    return ctypes.c_void_p(s if s else None)


# ----------------------------------------------------------------------------------------------------------------

@dataclass
class UNetConfig:
    """reference UNetConfig (unet/mod.rs:59-69) + DiffuserConfig.is_refiner (stablediffusion/mod.rs:269-278)"""
    adm_in_channels: int
    model_channels: int
    channel_mults: List[int]
    n_head_channels: int
    transformer_depths: List[int]
    context_dim: int
    in_channels: int = 4
    out_channels: int = 4
    is_refiner: bool = False

    def to_c(self) -> _UNetConfigC:
        c = _UNetConfigC()
        c.adm_in_channels, c.in_channels, c.out_channels = self.adm_in_channels, self.in_channels, self.out_channels
        c.model_channels, c.n_levels = self.model_channels, len(self.channel_mults)
        for i, (m, d) in enumerate(zip(self.channel_mults, self.transformer_depths)):
            c.channel_mults[i] = m
            c.transformer_depths[i] = d
        c.n_head_channels, c.context_dim, c.is_refiner = self.n_head_channels, self.context_dim, int(self.is_refiner)
        return c


def sdxl_base_config() -> UNetConfig:
    return UNetConfig(2816, 320, [1, 2, 4], 64, [0, 2, 10], 2048)


def sdxl_refiner_config() -> UNetConfig:
    return UNetConfig(2560, 384, [1, 2, 4, 4], 64, [0, 4, 4, 4], 1280, is_refiner=True)


@dataclass
class VAEConfig:
    enc_channels: List[Tuple[int, int]] = field(default_factory=lambda: [(128, 128), (128, 256), (256, 512), (512, 512)])
    dec_channels: List[Tuple[int, int]] = field(default_factory=lambda: [(512, 512), (512, 512), (512, 256), (256, 128)])
    n_group: int = 32
    enc_out_channels: int = 8
    scale_factor: float = 0.13025

    def to_c(self) -> _VaeConfigC:
        c = _VaeConfigC()
        c.n_blocks = len(self.dec_channels)
        for i, ((ei, eo), (di, do)) in enumerate(zip(self.enc_channels, self.dec_channels)):
            c.enc_in[i], c.enc_out[i], c.dec_in[i], c.dec_out[i] = ei, eo, di, do
        c.n_group, c.enc_out_channels, c.scale_factor = self.n_group, self.enc_out_channels, self.scale_factor
        return c


@dataclass
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    kind: int
    scale: float
    mean: float


def _specs(count_fn, spec_fn) -> List[ParamSpec]:
    n = count_fn()
    if n < 0:
        raise EngineError(lib().sdxl_last_error().decode())
    out = []
    name = ctypes.c_char_p()
    ndim, kind = ctypes.c_int(), ctypes.c_int()
    shape = (ctypes.c_int64 * 4)()
    sc, mean = ctypes.c_float(), ctypes.c_float()
    for i in range(n):
        _check(spec_fn(i, ctypes.byref(name), ctypes.byref(ndim), shape, ctypes.byref(kind), ctypes.byref(sc),
                       ctypes.byref(mean)))
        out.append(ParamSpec(name.value.decode(), tuple(int(shape[j]) for j in range(ndim.value)), kind.value,
                             sc.value, mean.value))
    return out


def unet_param_specs(cfg: UNetConfig) -> List[ParamSpec]:
    """host-only (no GPU needed): the order / layouts sdxl_unet_create expects its flat weight buffer in"""
    c = cfg.to_c()
    l = lib()
    return _specs(lambda: l.sdxl_unet_param_count(ctypes.byref(c)),
                  lambda i, *a: l.sdxl_unet_param_spec(ctypes.byref(c), i, *a))


def vae_param_specs(cfg: VAEConfig, encoder: bool) -> List[ParamSpec]:
    c = cfg.to_c()
    l = lib()
    return _specs(lambda: l.sdxl_vae_param_count(ctypes.byref(c), int(encoder)),
                  lambda i, *a: l.sdxl_vae_param_spec(ctypes.byref(c), int(encoder), i, *a))


def clip_param_specs(cfg: "CLIPConfig") -> List[ParamSpec]:
    c = cfg.to_c()
    l = lib()
    return _specs(lambda: l.sdxl_clip_param_count(ctypes.byref(c)),
                  lambda i, *a: l.sdxl_clip_param_spec(ctypes.byref(c), i, *a))


def step_count(n_steps: int, step_start: int = 0, n_train: int = 1000) -> int:
    return lib().sdxl_step_count(n_steps, step_start, n_train)


def flatten_weights(specs: Sequence[ParamSpec], weights: dict) -> np.ndarray:
    """name -> ndarray (reference layouts) => the flat fp32 buffer of the C ABI"""
    parts = []
    for p in specs:
        w = np.asarray(weights[p.name], dtype=np.float32)
        if tuple(w.shape) != tuple(p.shape):
            raise EngineError(f"{p.name}: expected shape {p.shape}, got {w.shape}")
        parts.append(w.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))


class Context:
    """one per GPU (reference: LibTorchDevice::Cuda(0), src/bin/sample/main.rs:131)"""

    def __init__(self, device_id: int = 0):
        self.h = ctypes.c_void_p()
        _check(lib().sdxl_ctx_create(device_id, ctypes.byref(self.h)))
        self.device_id = device_id

    def synchronize(self):
        _check(lib().sdxl_ctx_synchronize(self.h))

    def __del__(self):
        if getattr(self, "h", None) and self.h and _lib is not None:
            _lib.sdxl_ctx_destroy(self.h)
            self.h = None


@dataclass
class Conditioning:
    """reference Conditioning<B> (stablediffusion/mod.rs:544-555); CUDA fp32 tensors of the same ranks"""
    context_full: "object" = None                       # [n,77,ctx_full]
    channel_context: "object" = None                    # [n,adm]
    unconditional_context_full: "object" = None         # [77,ctx_full]
    unconditional_channel_context: "object" = None      # [adm]
    context_open_clip: "object" = None                  # [n,77,1280]
    channel_context_refiner: "object" = None
    unconditional_context_open_clip: "object" = None
    unconditional_channel_context_refiner: "object" = None
    resolution: Tuple[int, int] = (1024, 1024)          # (height, width)

    def to_c(self):
        keep = []
        c = _ConditioningC()
        first = None
        for name in ("unconditional_context_full", "unconditional_context_open_clip", "context_full", "context_open_clip",
                     "unconditional_channel_context", "unconditional_channel_context_refiner", "channel_context",
                     "channel_context_refiner"):
            t = getattr(self, name)
            if t is None:
                setattr(c, name, None)
                continue
            t, p = _dev(t)
            keep.append(t)
            setattr(c, name, p)
            if name in ("context_full", "context_open_clip") and first is None:
                first = t
        if first is None:
            raise EngineError("Conditioning needs context_full or context_open_clip")
        c.n, c.n_ctx = int(first.shape[0]), int(first.shape[1])
        c.height, c.width = int(self.resolution[0]), int(self.resolution[1])
        return c, keep


class UNet:
    """reference UNet<B> (src/model/unet/mod.rs:432-493)"""

    def __init__(self, ctx: Context, cfg: UNetConfig, dtype: int = DTYPE_F16, weights: Optional[np.ndarray] = None,
                 seed: int = 0, _borrowed=None):
        self.ctx, self.cfg, self.dtype = ctx, cfg, dtype
        self._owned = _borrowed is None
        if _borrowed is not None:
            self.h = _borrowed
            return
        self.h = ctypes.c_void_p()
        c = cfg.to_c()
        if weights is None:
            _check(lib().sdxl_unet_create_synthetic(ctx.h, ctypes.byref(c), dtype, ctypes.c_uint64(seed), ctypes.byref(self.h)))
        elif np.asarray(weights).dtype == np.float16:     # flat f16 (burn HalfPrecisionSettings records): no fp32 expansion
            w = np.ascontiguousarray(weights)
            _check(lib().sdxl_unet_create_f16(ctx.h, ctypes.byref(c), dtype, w.ctypes.data_as(ctypes.c_void_p), ctypes.byref(self.h)))
        else:
            w = np.ascontiguousarray(weights, dtype=np.float32)
            _check(lib().sdxl_unet_create(ctx.h, ctypes.byref(c), dtype, w.ctypes.data_as(ctypes.c_void_p), ctypes.byref(self.h)))

    def forward(self, x, timesteps, context, label):
        """UNet::forward(x[B,4,H,W], timesteps[B] int, context[B,n_ctx,ctx], label[B,adm]) -> [B,4,H,W]  (:450-492)"""
        torch = _torch()
        x, px = _dev(x)
        ts, pt = _dev(timesteps, torch.int32)
        context, pc = _dev(context)
        label, pl = _dev(label)
        B, _, H, W = x.shape
        out = torch.empty((B, self.cfg.out_channels, H, W), device=x.device, dtype=torch.float32)
        _check(lib().sdxl_unet_forward(self.h, _stream(), px, pt, pc, pl, B, H, W, int(context.shape[1]),
                                      ctypes.c_void_p(out.data_ptr())))
        return out

    def set_split_cfg(self, enabled: bool, release_offset: int = 0):
        """per-handle: run the CFG pair (batch-2 forward) as two concurrent batch-1 chains; bit-identical results"""
        _check(lib().sdxl_unet_set_split_cfg(self.h, int(enabled), int(release_offset)))

    def set_fused_cross_attention(self, enabled: bool):
        """per-handle (default on): cross-attention inside the query projection's epilogue"""
        _check(lib().sdxl_unet_set_fused_cross_attention(self.h, int(enabled)))

    def set_gn_from_producer(self, enabled: bool):
        """per-handle (default on): GroupNorm statistics from the producing convolution's epilogue where its kernel can"""
        _check(lib().sdxl_unet_set_gn_from_producer(self.h, int(enabled)))

    def set_graph(self, enabled: bool):
        _check(lib().sdxl_unet_set_graph(self.h, int(enabled)))

    def mix_classes(self) -> int:
        """MIX_* classes on plain f16 operands (F32_SPLIT_MIX* models; an F16W model on parameters that are not f16 values reports F32_SPLIT_MIX's 1 | 2 | 1024)"""
        v = ctypes.c_int(0)
        _check(lib().sdxl_unet_mix_classes(self.h, ctypes.byref(v)))
        return v.value

    def weight_arena(self) -> Tuple[int, int]:
        base, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().sdxl_unet_weight_arena(self.h, ctypes.byref(base), ctypes.byref(n)))
        return int(base.value or 0), int(n.value)

    def weight_arena_tensor(self):
        """zero-copy uint8 view of the packed weight arena (for the one-time RCCL broadcast from rank 0)"""
        return arena_as_tensor(*self.weight_arena(), self.ctx.device_id)

    PROFILE_CLASSES = ("igemm", "attention", "groupnorm", "layernorm", "other")

    def profile(self, B: int, H: int, W: int):
        """one eager forward with hipEvents around every launch -> {class: (ms, launches, flops)}"""
        ms, ln, fl = (ctypes.c_float * 5)(), (ctypes.c_int * 5)(), (ctypes.c_double * 5)()
        _check(lib().sdxl_unet_profile(self.h, _stream(), B, H, W, ms, ln, fl))
        return {c: (float(ms[i]), int(ln[i]), float(fl[i])) for i, c in enumerate(self.PROFILE_CLASSES)}

    def eager_forward_ms(self, B: int, H: int, W: int) -> float:
        """the chain profile() runs without the per-launch events (one event pair around it, best of three)"""
        v = ctypes.c_float(0)
        _check(lib().sdxl_unet_eager_forward_ms(self.h, _stream(), B, H, W, ctypes.byref(v)))
        return float(v.value)

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "h", None) and _lib is not None:
            _lib.sdxl_unet_destroy(self.h)
            self.h = None


@dataclass
class CLIPConfig:
    """reference CLIPConfig (src/model/clip/mod.rs:19-28)"""
    n_vocab: int
    n_state: int
    embed_dim: int
    n_head: int
    n_ctx: int
    n_layer: int
    quick_gelu: bool

    def to_c(self) -> _ClipConfigC:
        return _ClipConfigC(self.n_vocab, self.n_state, self.embed_dim, self.n_head, self.n_ctx, self.n_layer,
                            int(self.quick_gelu))


def clip_l_config() -> CLIPConfig:
    return CLIPConfig(49408, 768, 768, 12, 77, 12, True)


def open_clip_bigg_config() -> CLIPConfig:
    return CLIPConfig(49408, 1280, 1280, 20, 77, 32, False)


class CLIP:
    """reference CLIP<B> (src/model/clip/mod.rs:62-151): text transformer on the GPU, token ids in"""

    def __init__(self, ctx: Context, cfg: CLIPConfig, dtype: int = DTYPE_F16, weights: Optional[np.ndarray] = None,
                 seed: int = 0):
        self.ctx, self.cfg, self.dtype = ctx, cfg, dtype
        self.h = ctypes.c_void_p()
        c = cfg.to_c()
        if weights is None:
            _check(lib().sdxl_clip_create_synthetic(ctx.h, ctypes.byref(c), dtype, ctypes.c_uint64(seed), ctypes.byref(self.h)))
        elif np.asarray(weights).dtype == np.float16:
            w = np.ascontiguousarray(weights)
            _check(lib().sdxl_clip_create_f16(ctx.h, ctypes.byref(c), dtype, w.ctypes.data_as(ctypes.c_void_p), ctypes.byref(self.h)))
        else:
            w = np.ascontiguousarray(weights, dtype=np.float32)
            _check(lib().sdxl_clip_create(ctx.h, ctypes.byref(c), dtype, w.ctypes.data_as(ctypes.c_void_p), ctypes.byref(self.h)))

    def max_sequence_length(self) -> int:
        return self.cfg.n_ctx

    def num_layers(self) -> int:
        return self.cfg.n_layer

    def _tokens(self, tokens):
        torch = _torch()
        t = torch.as_tensor(tokens)
        if t.dim() != 2:
            raise EngineError("tokens must be [n_batch, seq_len]")
        if int(t.min()) < 0 or int(t.max()) >= self.cfg.n_vocab:
            raise EngineError("token id outside the vocabulary")
        return _dev(t.to(f"cuda:{self.ctx.device_id}"), torch.int32)

    def forward_hidden(self, tokens, hidden_idx: int):
        """CLIP::forward_hidden (:94-112) -> [n, seq, n_state]"""
        torch = _torch()
        t, pt = self._tokens(tokens)
        n, seq = t.shape
        out = torch.empty((n, seq, self.cfg.n_state), device=t.device, dtype=torch.float32)
        _check(lib().sdxl_clip_forward_hidden(self.h, _stream(), pt, n, seq, int(hidden_idx), ctypes.c_void_p(out.data_ptr())))
        return out

    def forward_hidden_pooled(self, tokens, hidden_idx: int):
        """CLIP::forward_hidden_pooled (:114-151) -> ([n, seq, n_state], [n, embed_dim])"""
        torch = _torch()
        t, pt = self._tokens(tokens)
        n, seq = t.shape
        hidden = torch.empty((n, seq, self.cfg.n_state), device=t.device, dtype=torch.float32)
        pooled = torch.empty((n, self.cfg.embed_dim), device=t.device, dtype=torch.float32)
        _check(lib().sdxl_clip_forward_hidden_pooled(self.h, _stream(), pt, n, seq, int(hidden_idx),
                                                     ctypes.c_void_p(hidden.data_ptr()), ctypes.c_void_p(pooled.data_ptr())))
        return hidden, pooled

    def weight_arena(self) -> Tuple[int, int]:
        base, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().sdxl_clip_weight_arena(self.h, ctypes.byref(base), ctypes.byref(n)))
        return int(base.value or 0), int(n.value)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.sdxl_clip_destroy(self.h)
            self.h = None


def conditioning_embedding(ctx: Context, pooled, dim: int, size, crop, ar):
    """reference conditioning_embedding (src/model/unet/mod.rs:41-57): [pooled | timestep_embedding(size|crop|ar)]"""
    torch = _torch()
    pooled, pp = _dev(pooled)
    vals = torch.cat([torch.as_tensor(v).to(pooled.device).reshape(pooled.shape[0], -1) for v in (size, crop, ar)], dim=1)
    vals, pv = _dev(vals, torch.int32)
    n, E = pooled.shape
    w = int(vals.shape[1])
    out = torch.empty((n, E + w * dim), device=pooled.device, dtype=torch.float32)
    _check(lib().sdxl_conditioning_embedding(ctx.h, _stream(), pp, n, E, pv, w, dim, ctypes.c_void_p(out.data_ptr())))
    return out


class Embedder:
    """reference Embedder<B> (stablediffusion/mod.rs:652-757).  The tokenizers are optional: without them (no asset files)
    only tokens_to_conditioning is available -- which is also what a Rust caller of the C ABI uses."""

    def __init__(self, ctx: Context, clip: CLIP, open_clip: CLIP, clip_tokenizer=None, open_clip_tokenizer=None):
        self.ctx, self.clip, self.open_clip = ctx, clip, open_clip
        self.clip_tokenizer, self.open_clip_tokenizer = clip_tokenizer, open_clip_tokenizer

    def _finish(self, full, open_ctx, pooled, size, crop, ar):
        """tail of Embedder::context / unconditional_context (:697-757): the two label vectors of one prompt"""
        torch = _torch()
        n = int(torch.as_tensor(ar).shape[0])
        if pooled.shape[0] != n:
            raise EngineError("the reference concatenates a [1, E] pooled embedding with [n_batch, .] size embeddings: n_batch must be 1")
        aesthetic = torch.full((n, 1), 6, dtype=torch.int32)                                           # :709,740
        return (full, open_ctx, conditioning_embedding(self.ctx, pooled, 256, size, crop, ar),
                conditioning_embedding(self.ctx, pooled, 256, size, crop, aesthetic))

    def tokens_to_conditioning(self, clip_ids, open_ids, uncond_clip_ids, uncond_open_ids, size, crop, ar) -> Conditioning:
        """text_to_conditioning (:661-696) after tokenize_text: ids [1, n_ctx]; size / crop [n, 2] ints, ar [2] ints"""
        torch = _torch()
        size, crop, ar = torch.as_tensor(size), torch.as_tensor(crop), torch.as_tensor(ar)
        n = int(size.shape[0])
        bar = ar.reshape(1, -1).repeat(n, 1)
        # the reference encodes "" and the prompt in two passes (:680-683); here they ride one batch-2 pass per encoder (the
        # rows of a batch are independent -- bit-identical to separate passes, tests/test_gpu_clip.py) and are split after
        cat = lambda a, b: torch.cat([torch.as_tensor(a).reshape(1, -1), torch.as_tensor(b).reshape(1, -1)], dim=0)   # noqa: E731
        clip_ctx = self.clip.forward_hidden(cat(uncond_clip_ids, clip_ids), self.clip.num_layers() - 1)      # :759-770
        open_ctx, pooled = self.open_clip.forward_hidden_pooled(cat(uncond_open_ids, open_ids), self.open_clip.num_layers() - 1)
        full = torch.cat([clip_ctx, open_ctx], dim=2)
        ucf, uco, ucc, uccr = self._finish(full[0:1], open_ctx[0:1], pooled[0:1], size, crop, bar)
        cf, co, cc, ccr = self._finish(full[1:2], open_ctx[1:2], pooled[1:2], size, crop, bar)
        return Conditioning(context_full=cf, channel_context=cc, unconditional_context_full=ucf.squeeze(0),
                            unconditional_channel_context=ucc.squeeze(0), context_open_clip=co, channel_context_refiner=ccr,
                            unconditional_context_open_clip=uco.squeeze(0), unconditional_channel_context_refiner=uccr.squeeze(0),
                            resolution=(int(ar[0]), int(ar[1])))

    def text_to_conditioning(self, text: str, size, crop, ar) -> Conditioning:
        """Embedder::text_to_conditioning (:661-696); the unconditional prompt is "" (:703-705)"""
        if self.clip_tokenizer is None or self.open_clip_tokenizer is None:
            raise EngineError("Embedder was built without tokenizers (asset files not available): use tokens_to_conditioning")
        from .tokenizer import tokenize_text
        torch = _torch()
        ids = lambda t, tok, m: torch.tensor([tokenize_text(t, tok, m.max_sequence_length())], dtype=torch.int32)   # noqa: E731
        return self.tokens_to_conditioning(ids(text, self.clip_tokenizer, self.clip), ids(text, self.open_clip_tokenizer, self.open_clip),
                                           ids("", self.clip_tokenizer, self.clip), ids("", self.open_clip_tokenizer, self.open_clip),
                                           size, crop, ar)


class _ArenaView:
    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def arena_as_tensor(ptr: int, nbytes: int, device_id: int = 0):
    torch = _torch()
    return torch.as_tensor(_ArenaView(ptr, nbytes), device=f"cuda:{device_id}")


def debug_set(key: str, value: int):
    _check(lib().sdxl_debug_set(key.encode(), int(value)))


def bench_igemm(ctx: "Context", B: int, H: int, W: int, Cin: int, Cout: int, ksize: int = 1, geglu: bool = False,
                iters: int = 20) -> float:
    """mean launch duration (ms) of the implicit-GEMM kernel alone on seeded random f16 data"""
    ms = ctypes.c_float()
    _check(lib().sdxl_bench_igemm(ctx.h, None, B, H, W, Cin, Cout, ksize, int(geglu), iters, ctypes.byref(ms)))   # geglu: bit0 GEGLU, bit1 LN-folded input, bit2 row statistics out
    return float(ms.value)


def bench_attention(ctx: "Context", B: int, H: int, Nq: int, Nk: int, iters: int = 20) -> float:
    """mean launch duration (ms) of the fused d=64 f16 attention kernel alone on seeded random data"""
    ms = ctypes.c_float()
    _check(lib().sdxl_bench_attention(ctx.h, None, B, H, Nq, Nk, iters, ctypes.byref(ms)))
    return float(ms.value)


def default_alphas_cumprod(n: int = 1000) -> np.ndarray:
    """sgm LegacyDDPMDiscretization (reference python/dump.py:29-31): the loaded `alpha_cumulative_products` param"""
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, n, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas).astype(np.float32)


class Diffuser:
    """reference Diffuser<B> (src/model/stablediffusion/mod.rs:308-542)"""

    def __init__(self, ctx: Context, cfg: UNetConfig, dtype: int = DTYPE_F16, weights: Optional[np.ndarray] = None,
                 seed: int = 0, alphas_cumprod: Optional[np.ndarray] = None, empty: bool = False):
        self.ctx, self.cfg, self.dtype = ctx, cfg, dtype
        a = np.ascontiguousarray(default_alphas_cumprod() if alphas_cumprod is None else alphas_cumprod, dtype=np.float32)
        self.n_train = int(a.shape[0])
        self.h = ctypes.c_void_p()
        c = cfg.to_c()
        ap = a.ctypes.data_as(ctypes.c_void_p)
        if empty:   # replica rank: arena laid out, contents arrive by broadcast
            _check(lib().sdxl_diffuser_create_empty(ctx.h, ctypes.byref(c), dtype, ap, self.n_train, ctypes.byref(self.h)))
        elif weights is None:
            _check(lib().sdxl_diffuser_create_synthetic(ctx.h, ctypes.byref(c), dtype, ctypes.c_uint64(seed), ap,
                                                       self.n_train, ctypes.byref(self.h)))
        elif np.asarray(weights).dtype == np.float16:
            w = np.ascontiguousarray(weights)
            _check(lib().sdxl_diffuser_create_f16(ctx.h, ctypes.byref(c), dtype, w.ctypes.data_as(ctypes.c_void_p), ap,
                                                 self.n_train, ctypes.byref(self.h)))
        else:
            w = np.ascontiguousarray(weights, dtype=np.float32)
            _check(lib().sdxl_diffuser_create(ctx.h, ctypes.byref(c), dtype, w.ctypes.data_as(ctypes.c_void_p), ap,
                                             self.n_train, ctypes.byref(self.h)))
        self.diffusion = UNet(ctx, cfg, dtype, _borrowed=ctypes.c_void_p(lib().sdxl_diffuser_unet(self.h)))

    def _latent_shape(self, cond: Conditioning):
        n = int((cond.context_full if cond.context_full is not None else cond.context_open_clip).shape[0])
        return (n, 4, cond.resolution[0] // 8, cond.resolution[1] // 8)

    def sample_latent(self, conditioning: Conditioning, unconditional_guidance_scale: float, n_steps: int, noise0):
        """Diffuser::sample_latent (:317-332); noise0 plays gen_noise()"""
        torch = _torch()
        c, keep = conditioning.to_c()
        noise0, pn = _dev(noise0)
        assert tuple(noise0.shape) == self._latent_shape(conditioning)
        out = torch.empty_like(noise0)
        _check(lib().sdxl_sample_latent(self.h, _stream(), ctypes.byref(c), ctypes.c_double(unconditional_guidance_scale),
                                       n_steps, pn, ctypes.c_void_p(out.data_ptr())))
        return out

    def sample_latent_with_inpainting(self, conditioning, unconditional_guidance_scale, n_steps, reference, mask, noise0,
                                      step_noise):
        """Diffuser::sample_latent_with_inpainting (:334-353); mask True = keep generated; step_noise [iters,n,4,h,w]"""
        torch = _torch()
        c, keep = conditioning.to_c()
        noise0, pn = _dev(noise0)
        reference, pr = _dev(reference)
        mask, pm = _dev(mask, torch.uint8)
        step_noise, ps = _dev(step_noise)
        out = torch.empty_like(noise0)
        _check(lib().sdxl_sample_latent_with_inpainting(self.h, _stream(), ctypes.byref(c),
                                                       ctypes.c_double(unconditional_guidance_scale), n_steps, pr, pm, pn,
                                                       ps, ctypes.c_void_p(out.data_ptr())))
        return out

    def refine_latent(self, latent, conditioning, unconditional_guidance_scale, step_start, n_steps, noise):
        """Diffuser::refine_latent (:355-376)"""
        torch = _torch()
        c, keep = conditioning.to_c()
        latent, pl = _dev(latent)
        noise, pn = _dev(noise)
        out = torch.empty_like(latent)
        _check(lib().sdxl_refine_latent(self.h, _stream(), pl, ctypes.byref(c), ctypes.c_double(unconditional_guidance_scale),
                                       step_start, n_steps, pn, ctypes.c_void_p(out.data_ptr())))
        return out

    def enable_step_timing(self, enabled: bool = True):
        _check(lib().sdxl_diffuser_enable_step_timing(self.h, int(enabled)))

    def set_trace(self, trace=None):
        """trace: float32 device tensor [steps, n, 4, h, w] that receives the latent after every DDIM iteration of the
        following trajectories (None switches it off).  The tensor must stay alive while tracing is on."""
        self._trace = trace
        if trace is None:
            _check(lib().sdxl_diffuser_set_trace(self.h, None, 0))
        else:
            assert trace.is_cuda and trace.dtype == _torch().float32 and trace.is_contiguous()
            _check(lib().sdxl_diffuser_set_trace(self.h, ctypes.c_void_p(trace.data_ptr()), int(trace.shape[0])))

    def step_times_ms(self) -> List[float]:
        buf = (ctypes.c_float * 1024)()
        n = lib().sdxl_diffuser_step_times(self.h, buf, 1024)
        return [float(buf[i]) for i in range(max(n, 0))]

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            self.diffusion = None
            _lib.sdxl_diffuser_destroy(self.h)
            self.h = None


@dataclass
class RawImages:
    """reference RawImages (stablediffusion/mod.rs:170-174): u8 HWC buffers"""
    buffer: "object"   # uint8 CUDA tensor [n, height, width, 3]
    width: int
    height: int


class LatentDecoder:
    """reference LatentDecoder<B> (src/model/stablediffusion/mod.rs:193-267) over Autoencoder (autoencoder/mod.rs:46-70)"""

    def __init__(self, ctx: Context, cfg: Optional[VAEConfig] = None, dtype: int = DTYPE_F16,
                 decoder_weights: Optional[np.ndarray] = None, encoder_weights: Optional[np.ndarray] = None,
                 seed: int = 0, with_encoder: bool = False, empty: bool = False):
        self.ctx, self.cfg, self.dtype = ctx, cfg or VAEConfig(), dtype
        self.h = ctypes.c_void_p()
        c = self.cfg.to_c()
        if empty:
            _check(lib().sdxl_vae_create_empty(ctx.h, ctypes.byref(c), dtype, int(with_encoder), ctypes.byref(self.h)))
        elif decoder_weights is None and encoder_weights is None:
            _check(lib().sdxl_vae_create_synthetic(ctx.h, ctypes.byref(c), dtype, ctypes.c_uint64(seed), int(with_encoder),
                                                  ctypes.byref(self.h)))
        elif any(w is not None and np.asarray(w).dtype == np.float16 for w in (decoder_weights, encoder_weights)):
            d = None if decoder_weights is None else np.ascontiguousarray(decoder_weights, dtype=np.float16)
            e = None if encoder_weights is None else np.ascontiguousarray(encoder_weights, dtype=np.float16)
            _check(lib().sdxl_vae_create_f16(ctx.h, ctypes.byref(c), dtype,
                                            None if d is None else d.ctypes.data_as(ctypes.c_void_p),
                                            None if e is None else e.ctypes.data_as(ctypes.c_void_p), ctypes.byref(self.h)))
        else:
            d = None if decoder_weights is None else np.ascontiguousarray(decoder_weights, dtype=np.float32)
            e = None if encoder_weights is None else np.ascontiguousarray(encoder_weights, dtype=np.float32)
            _check(lib().sdxl_vae_create(ctx.h, ctypes.byref(c), dtype,
                                        None if d is None else d.ctypes.data_as(ctypes.c_void_p),
                                        None if e is None else e.ctypes.data_as(ctypes.c_void_p), ctypes.byref(self.h)))

    def decode_latent(self, latent):
        """LatentDecoder::decode_latent (:263-266): [n,4,h,w] -> [n,3,8h,8w]"""
        torch = _torch()
        latent, pl = _dev(latent)
        n, _, h, w = latent.shape
        out = torch.empty((n, 3, 8 * h, 8 * w), device=latent.device, dtype=torch.float32)
        _check(lib().sdxl_vae_decode_latent(self.h, _stream(), pl, n, h, w, ctypes.c_void_p(out.data_ptr())))
        return out

    def latent_to_image(self, latent) -> RawImages:
        """LatentDecoder::latent_to_image (:200-237)"""
        torch = _torch()
        latent, pl = _dev(latent)
        n, _, h, w = latent.shape
        out = torch.empty((n, 8 * h, 8 * w, 3), device=latent.device, dtype=torch.uint8)
        _check(lib().sdxl_latent_to_image(self.h, _stream(), pl, n, h, w, ctypes.c_void_p(out.data_ptr())))
        return RawImages(out, 8 * w, 8 * h)

    def encode_image(self, image):
        """LatentDecoder::encode_image (:257-261): [n,3,H,W] in [-1,1] -> [n,4,H/8,W/8]"""
        torch = _torch()
        image, pi = _dev(image)
        n, _, H, W = image.shape
        out = torch.empty((n, 4, H // 8, W // 8), device=image.device, dtype=torch.float32)
        _check(lib().sdxl_vae_encode_image(self.h, _stream(), pi, n, H, W, ctypes.c_void_p(out.data_ptr())))
        return out

    def image_to_latent(self, images: RawImages):
        """LatentDecoder::image_to_latent (:239-255)"""
        torch = _torch()
        buf, pb = _dev(images.buffer, torch.uint8)
        n = buf.shape[0]
        out = torch.empty((n, 4, images.height // 8, images.width // 8), device=buf.device, dtype=torch.float32)
        _check(lib().sdxl_image_to_latent(self.h, _stream(), pb, n, images.height, images.width, ctypes.c_void_p(out.data_ptr())))
        return out

    def weight_arena(self) -> Tuple[int, int]:
        base, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib().sdxl_vae_weight_arena(self.h, ctypes.byref(base), ctypes.byref(n)))
        return int(base.value or 0), int(n.value)

    def weight_arena_tensor(self):
        return arena_as_tensor(*self.weight_arena(), self.ctx.device_id)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.sdxl_vae_destroy(self.h)
            self.h = None


# ---------------------------------------------------------------------------------------------------------------- ops

def qkv_attention(ctx: Context, q, k, v, mask, n_head: int, dtype: int = DTYPE_F16):
    """Backend::qkv_attention (src/backend.rs:4-19): q [B,Nq,C], k,v [B,Nk,C], additive mask [Nq,Nk] or None"""
    torch = _torch()
    q, pq = _dev(q)
    k, pk = _dev(k)
    v, pv = _dev(v)
    pm = None
    if mask is not None:
        mask, pm = _dev(mask)
    B, Nq, C = q.shape
    out = torch.empty_like(q)
    _check(lib().sdxl_qkv_attention(ctx.h, _stream(), pq, pk, pv, pm, B, Nq, int(k.shape[1]), C, n_head, dtype,
                                   ctypes.c_void_p(out.data_ptr())))
    return out


def attn_decoder_mask(ctx: Context, seq_length: int):
    """Backend::attn_decoder_mask (src/backend.rs:130-136)"""
    torch = _torch()
    out = torch.empty((seq_length, seq_length), device=f"cuda:{ctx.device_id}", dtype=torch.float32)
    _check(lib().sdxl_attn_decoder_mask(ctx.h, _stream(), seq_length, ctypes.c_void_p(out.data_ptr())))
    torch.cuda.synchronize()
    return out


def group_norm(ctx: Context, x, gamma, beta, n_group: int = 32, eps: float = 1e-5, silu: bool = False,
               dtype: int = DTYPE_F16):
    """GroupNorm::forward (groupnorm/mod.rs:52-73) on NCHW input, optional fused SILU (silu.rs:14-16)"""
    torch = _torch()
    x, px = _dev(x)
    gamma, pg = _dev(gamma)
    beta, pb = _dev(beta)
    B, C = x.shape[0], x.shape[1]
    HW = int(np.prod(x.shape[2:]))
    out = torch.empty_like(x)
    _check(lib().sdxl_group_norm(ctx.h, _stream(), px, pg, pb, B, C, HW, n_group, ctypes.c_float(eps), int(silu), dtype,
                                ctypes.c_void_p(out.data_ptr())))
    return out


def layer_norm(ctx: Context, x, gamma, beta, eps: float = 1e-5, dtype: int = DTYPE_F16):
    """LayerNorm::forward (layernorm/mod.rs:34-40)"""
    torch = _torch()
    x, px = _dev(x)
    gamma, pg = _dev(gamma)
    beta, pb = _dev(beta)
    C = x.shape[-1]
    out = torch.empty_like(x)
    _check(lib().sdxl_layer_norm(ctx.h, _stream(), px, pg, pb, int(x.numel() // C), C, ctypes.c_float(eps), dtype,
                                ctypes.c_void_p(out.data_ptr())))
    return out


def conv2d(ctx: Context, x, weight, bias, stride: int = 1, padding: int = 0, upsample: bool = False,
           dtype: int = DTYPE_F16):
    """burn Conv2d (weight [Cout,Cin,k,k]); upsample=True applies the reference's nearest-2x first (unet/mod.rs:744-750)"""
    torch = _torch()
    x, px = _dev(x)
    weight, pw = _dev(weight)
    pb = None
    if bias is not None:
        bias, pb = _dev(bias)
    B, Cin, H, W = x.shape
    Cout, _, k, _ = weight.shape
    Hs, Ws = (2 * H, 2 * W) if upsample else (H, W)
    Ho, Wo = (Hs + 2 * padding - k) // stride + 1, (Ws + 2 * padding - k) // stride + 1
    out = torch.empty((B, Cout, Ho, Wo), device=x.device, dtype=torch.float32)
    _check(lib().sdxl_conv2d(ctx.h, _stream(), px, pw, pb, B, Cin, H, W, Cout, k, stride, padding, int(upsample), dtype,
                            ctypes.c_void_p(out.data_ptr())))
    return out


def linear(ctx: Context, x, weight, bias, geglu: bool = False, dtype: int = DTYPE_F16):
    """burn nn::Linear: x[...,K] @ weight[K,N] + bias; geglu=True -> GEGLU::forward (unet/mod.rs:942-956)"""
    torch = _torch()
    x, px = _dev(x)
    weight, pw = _dev(weight)
    pb = None
    if bias is not None:
        bias, pb = _dev(bias)
    K, N = weight.shape
    M = int(x.numel() // K)
    out = torch.empty(tuple(x.shape[:-1]) + ((N // 2) if geglu else N,), device=x.device, dtype=torch.float32)
    _check(lib().sdxl_linear(ctx.h, _stream(), px, pw, pb, M, K, N, int(geglu), dtype, ctypes.c_void_p(out.data_ptr())))
    return out


def layer_norm_linear(ctx: Context, x, gamma, beta, weight, bias, eps: float = 1e-5, geglu: bool = False,
                      dtype: int = DTYPE_F16):
    """LayerNorm::forward (layernorm/mod.rs:34-49) then nn::Linear, as TransformerBlock::forward pairs them
    (unet/mod.rs:885-891).  DTYPE_F16 runs the LayerNorm FOLDED into the GEMM (the UNet's f16 path), the other dtypes the
    stand-alone LayerNorm kernel."""
    torch = _torch()
    x, px = _dev(x)
    gamma, pg = _dev(gamma)
    beta, pbeta = _dev(beta)
    weight, pw = _dev(weight)
    pb = None
    if bias is not None:
        bias, pb = _dev(bias)
    K, N = weight.shape
    M = int(x.numel() // K)
    out = torch.empty(tuple(x.shape[:-1]) + ((N // 2) if geglu else N,), device=x.device, dtype=torch.float32)
    _check(lib().sdxl_layer_norm_linear(ctx.h, _stream(), px, pg, pbeta, ctypes.c_float(eps), pw, pb, M, K, N, int(geglu),
                                       dtype, ctypes.c_void_p(out.data_ptr())))
    return out


def ln_query_cross_attention(ctx: Context, x, gamma, beta, wq, k, v, eps: float = 1e-5, fused: bool = True):
    """attn2 of a transformer block up to its output projection (unet/mod.rs:731-795): LayerNorm -> query projection (no
    bias) -> qkv_attention over the projected context k, v [B,Nk,C] with 64 channels per head.  f16 engine arithmetic.
    fused=True runs the attention inside the projection's epilogue (one launch), False as projection + attention kernel; fused=2: that epilogue at
    split precision (context, q and P as (hi, lo) f16 pairs, three MFMAs per product: what SDXL_DTYPE_F32_SPLIT_MIX_F16W runs)."""
    torch = _torch()
    x, px = _dev(x)
    gamma, pg = _dev(gamma)
    beta, pbeta = _dev(beta)
    wq, pw = _dev(wq)
    k, pk = _dev(k)
    v, pv = _dev(v)
    B, Nq, C = x.shape
    Nk = int(k.shape[1])
    out = torch.empty((B, Nq, C), device=x.device, dtype=torch.float32)
    _check(lib().sdxl_ln_query_cross_attention(ctx.h, _stream(), px, pg, pbeta, ctypes.c_float(eps), pw, pk, pv, int(B), int(Nq),
                                              Nk, int(C), int(fused), ctypes.c_void_p(out.data_ptr())))
    return out


def conv2d_group_norm(ctx: Context, x, weight, bias, gamma, beta, eps: float = 1e-5, n_group: int = 32, silu: bool = True,
                      residual=None, fused: bool = True):
    """conv3x3 (pad 1, + residual) -> GroupNorm (+SiLU) as ResBlock::forward pairs them (unet/mod.rs:1082-1106), f16 engine.
    fused=True asks the convolution's epilogue for the GroupNorm statistics.  Returns (out, fused_taken)."""
    torch = _torch()
    x, px = _dev(x)
    weight, pw = _dev(weight)
    gamma, pg = _dev(gamma)
    beta, pbeta = _dev(beta)
    pb = pr = None
    if bias is not None:
        bias, pb = _dev(bias)
    if residual is not None:
        residual, pr = _dev(residual)
    B, Cin, H, W = x.shape
    Cout = int(weight.shape[0])
    out = torch.empty((B, Cout, H, W), device=x.device, dtype=torch.float32)
    took = ctypes.c_int(0)
    _check(lib().sdxl_conv2d_group_norm(ctx.h, _stream(), px, pw, pb, pr, pg, pbeta, ctypes.c_float(eps), int(B), int(Cin), int(H),
                                       int(W), Cout, int(n_group), int(silu), int(fused), ctypes.byref(took),
                                       ctypes.c_void_p(out.data_ptr())))
    return out, bool(took.value)


# ---------------------------------------------------------------------------------------------------------------- multi-GPU
def bcast_plan(nbytes: int, world: int, rank: int) -> Tuple[int, int, int, int]:
    """(piece_off, piece_len, tail_off, tail_len) of the library's weight-broadcast schedule for this rank: the root scatters
    `world` equal 256-byte-aligned pieces over its links, an in-place all-gather completes them, the tail is a small
    broadcast (csrc/comm.cpp).  Host-only: works without a GPU."""
    v = [ctypes.c_size_t() for _ in range(4)]
    _check(lib().sdxl_bcast_plan(ctypes.c_size_t(nbytes), world, rank, *[ctypes.byref(x) for x in v]))
    return tuple(int(x.value) for x in v)


class Comm:
    """RCCL communicator of the engine (one process per GPU).  `unique_id()` on rank 0 -> ship the 128 bytes to every rank
    over any host channel (here: the caller's torch.distributed group) -> Comm(device, rank, world, id)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (ctypes.c_char * 128)()
        _check(lib().sdxl_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, device_id: int, rank: int, world: int, uid: bytes):
        assert len(uid) == 128
        self.h = ctypes.c_void_p()
        self.rank, self.world = rank, world
        _check(lib().sdxl_comm_create(device_id, rank, world, ctypes.c_char_p(uid), ctypes.byref(self.h)))

    def bcast_unet(self, unet: "UNet", root: int = 0):
        _check(lib().sdxl_unet_bcast_weights(self.h, unet.h, root))

    def bcast_vae(self, vae: "LatentDecoder", root: int = 0):
        _check(lib().sdxl_vae_bcast_weights(self.h, vae.h, root))

    def bcast_clip(self, clip: "CLIP", root: int = 0):
        _check(lib().sdxl_clip_bcast_weights(self.h, clip.h, root))

    def bcast_buffer(self, tensor, root: int = 0):
        _check(lib().sdxl_bcast_buffer(self.h, None, ctypes.c_void_p(tensor.data_ptr()), ctypes.c_size_t(tensor.numel() * tensor.element_size()), root))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.sdxl_comm_destroy(self.h)
            self.h = None
