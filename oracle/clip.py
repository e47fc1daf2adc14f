"""CLIP text transformer + Embedder restatement -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows /root/reference/src/model/clip/mod.rs (CLIP::forward_hidden :94-112, forward_hidden_pooled :114-151,
ResidualDecoderAttentionBlock::forward :194-199, MultiHeadSelfAttention::forward :243-257, MLP::forward :296-306,
QuickGELU :309-320) and the Embedder in /root/reference/src/model/stablediffusion/mod.rs:626-801.  fp32 torch-CPU
functional ops; the attention is the crate's own generic `qkv_attention` (oracle.model.qkv_attention).
PARITY UNPINNED by the reference (no expected values exist for this path either; its one #[test] is the tokenizer KAT).
Pinned instead against an independent implementation of the same architecture: HF transformers' CLIPTextModelWithProjection
(the model python/clip.py dumps CLIP-L from) on shared random weights -- tests/test_cpu_oracle_and_abi.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .config import DEFAULT_EPS, KIND_BETA, KIND_BIAS, KIND_EPS, KIND_GAMMA, KIND_LINEAR_W, ParamSpec, _SQRT12, _wscale
from .model import attn_decoder_mask, conditioning_embedding, linear, ln, qkv_attention
from .pipeline import Conditioning

Tensor = torch.Tensor


@dataclass
class CLIPConfig:
    """clip/mod.rs:19-28"""
    n_vocab: int
    n_state: int
    embed_dim: int
    n_head: int
    n_ctx: int
    n_layer: int
    quick_gelu: bool


def clip_l_config() -> CLIPConfig:            # OpenAI CLIP ViT-L/14 text tower (SDXL's first encoder)
    return CLIPConfig(49408, 768, 768, 12, 77, 12, True)


def open_clip_bigg_config() -> CLIPConfig:    # OpenCLIP ViT-bigG/14 text tower (SDXL's second encoder)
    return CLIPConfig(49408, 1280, 1280, 20, 77, 32, False)


def tiny_clip_config() -> CLIPConfig:
    return CLIPConfig(49408, 128, 128, 2, 77, 3, True)


def tiny_open_clip_config() -> CLIPConfig:
    return CLIPConfig(49408, 192, 160, 3, 77, 4, False)


def clip_param_specs(cfg: CLIPConfig) -> List[ParamSpec]:
    """field order of CLIP / ResidualDecoderAttentionBlock / MultiHeadSelfAttention / MLP (clip/mod.rs:62-69,178-184,
    231-238,282-290); every Linear has a bias (nn::LinearConfig default)"""
    s: List[ParamSpec] = []

    def lin(name, d_in, d_out, gain=1.0):
        s.append(ParamSpec(name + ".weight", (d_in, d_out), KIND_LINEAR_W, _wscale(d_in, gain), np.float32(0)))
        s.append(ParamSpec(name + ".bias", (d_out,), KIND_BIAS, np.float32(_SQRT12 * 0.02), np.float32(0)))

    def norm(name, c):
        s.append(ParamSpec(name + ".gamma", (c,), KIND_GAMMA, np.float32(_SQRT12 * 0.02), np.float32(1)))
        s.append(ParamSpec(name + ".beta", (c,), KIND_BETA, np.float32(_SQRT12 * 0.02), np.float32(0)))
        s.append(ParamSpec(name + ".eps", (1,), KIND_EPS, np.float32(0), np.float32(DEFAULT_EPS)))   # layernorm/load.rs:17

    c = cfg.n_state
    s.append(ParamSpec("token_embedding.weight", (cfg.n_vocab, c), KIND_LINEAR_W, np.float32(_SQRT12 * 0.5), np.float32(0)))
    s.append(ParamSpec("position_embedding", (cfg.n_ctx, c), KIND_LINEAR_W, np.float32(_SQRT12 * 0.1), np.float32(0)))
    for i in range(cfg.n_layer):
        p = f"blocks.{i}"
        lin(p + ".attn.query", c, c)
        lin(p + ".attn.key", c, c)
        lin(p + ".attn.value", c, c)
        lin(p + ".attn.out", c, c, 0.5)
        norm(p + ".attn_ln", c)
        lin(p + ".mlp.fc1", c, 4 * c)
        lin(p + ".mlp.fc2", 4 * c, c, 0.5)
        norm(p + ".mlp_ln", c)
    norm("layer_norm", c)
    s.append(ParamSpec("text_projection", (c, cfg.embed_dim), KIND_LINEAR_W, _wscale(c, 1.0), np.float32(0)))
    return s


def _block(x: Tensor, mask: Tensor, W, p: str, cfg: CLIPConfig) -> Tensor:
    """ResidualDecoderAttentionBlock::forward (:194-199)"""
    h = ln(x, W, p + ".attn_ln")
    q, k, v = linear(h, W, p + ".attn.query"), linear(h, W, p + ".attn.key"), linear(h, W, p + ".attn.value")
    x = x + linear(qkv_attention(q, k, v, mask, cfg.n_head), W, p + ".attn.out")            # :243-257
    h = linear(ln(x, W, p + ".mlp_ln"), W, p + ".mlp.fc1")
    h = h * torch.sigmoid(h * 1.702) if cfg.quick_gelu else F.gelu(h)                        # :296-320
    return x + linear(h, W, p + ".mlp.fc2")


def _embed(cfg: CLIPConfig, W, tokens: Tensor) -> Tensor:
    seq = tokens.shape[1]
    return W["token_embedding.weight"][tokens.long()] + W["position_embedding"][:seq].unsqueeze(0)   # :99-105


def forward_hidden(cfg: CLIPConfig, W, tokens: Tensor, hidden_idx: int) -> Tensor:
    """CLIP::forward_hidden (:94-112): output of the first hidden_idx blocks, no final LayerNorm"""
    mask = attn_decoder_mask(tokens.shape[1])
    x = _embed(cfg, W, tokens)
    for i in range(hidden_idx):
        x = _block(x, mask, W, f"blocks.{i}", cfg)
    return x


def forward_hidden_pooled(cfg: CLIPConfig, W, tokens: Tensor, hidden_idx: int) -> Tuple[Tensor, Tensor]:
    """CLIP::forward_hidden_pooled (:114-151): (input of block hidden_idx, LayerNorm(last)[eot] @ text_projection)"""
    mask = attn_decoder_mask(tokens.shape[1])
    x = _embed(cfg, W, tokens)
    h_out = torch.empty_like(x)
    for i in range(cfg.n_layer):
        if i == hidden_idx:
            h_out = x.clone()
        x = _block(x, mask, W, f"blocks.{i}", cfg)
    eot = tokens.long().argmax(dim=1)           # eot_token is the highest id of each sequence (:139-140)
    normed = ln(x, W, "layer_norm")
    o = normed[torch.arange(tokens.shape[0]), eot]
    return h_out, o @ W["text_projection"]


class Embedder:
    """stablediffusion/mod.rs:652-757; tokenisation is passed in as ids (the tokenizers are host string code)"""

    def __init__(self, clip_cfg: CLIPConfig, clip_W, open_cfg: CLIPConfig, open_W):
        self.clip_cfg, self.clip_W, self.open_cfg, self.open_W = clip_cfg, clip_W, open_cfg, open_W

    def _context(self, clip_ids: Tensor, open_ids: Tensor, size: Tensor, crop: Tensor, ar: Tensor):
        """Embedder::context / unconditional_context (:697-757) for one token sequence each"""
        clip_ctx = forward_hidden(self.clip_cfg, self.clip_W, clip_ids, self.clip_cfg.n_layer - 1)          # :759-770
        open_ctx, pooled = forward_hidden_pooled(self.open_cfg, self.open_W, open_ids, self.open_cfg.n_layer - 1)
        n = ar.shape[0]
        aesthetic = torch.full((n, 1), 6, dtype=torch.int64)                                                # :709,740
        return (torch.cat([clip_ctx, open_ctx], dim=2), open_ctx,
                conditioning_embedding(pooled, 256, size, crop, ar),
                conditioning_embedding(pooled, 256, size, crop, aesthetic))

    def tokens_to_conditioning(self, clip_ids, open_ids, uncond_clip_ids, uncond_open_ids, size, crop, ar) -> Conditioning:
        """text_to_conditioning (:661-696) after tokenize_text: ids are [1, n_ctx]; size/crop [n,2], ar [2]"""
        n = size.shape[0]
        resolution = (int(ar[0]), int(ar[1]))
        bar = ar.unsqueeze(0).repeat(n, 1)
        ucf, uco, ucc, uccr = self._context(uncond_clip_ids, uncond_open_ids, size, crop, bar)
        cf, co, cc, ccr = self._context(clip_ids, open_ids, size, crop, bar)
        return Conditioning(ucf.squeeze(0), uco.squeeze(0), cf, co, ucc.squeeze(0), uccr.squeeze(0), cc, ccr, resolution)
