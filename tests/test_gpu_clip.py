"""Embedder parity through the C ABI (SURVEY section 8f "next" row 1): CLIP::forward_hidden / forward_hidden_pooled and
Embedder::text_to_conditioning after tokenisation, against oracle/clip.py on seeded synthetic weights.

Tolerances follow test_gpu_models.py: DTYPE_F32 1e-4 relative on a forward; the fp16-operand modes are held to the same
2.5e-3 / 1.3e-3 bounds (<= 2x the measured values) (fp16 rounding of weights, embeddings and activations through <= 12 pre-LN blocks).
Token ids are passed in (the tokenizer asset files do not travel to the GPU box; the tokenizers are CPU-tested).
"""
import numpy as np
import pytest
import torch

from oracle import clip as OCL, config as OC, model as OM
from util import max_abs, rel_err

pytestmark = pytest.mark.gpu

TOL = {0: 1e-4, 1: 2.5e-3, 2: 1.3e-3}   # f16 modes: <= 2x measured (1.25e-3 / 6.4e-4, profiles/r02_gpu_tests_final.log)


def _pcfg(pkg, c):
    return pkg.CLIPConfig(c.n_vocab, c.n_state, c.embed_dim, c.n_head, c.n_ctx, c.n_layer, c.quick_gelu)


def _ids(n, seq, seed, pad, eot_at=None):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, 49000, (n, seq), generator=g, dtype=torch.int64)
    ids[:, 0] = 49406
    for b in range(n):
        e = (eot_at[b] if eot_at else 3 + 5 * b) % seq
        ids[b, e] = 49407
        ids[b, e + 1:] = pad
    return ids


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("which", ["tiny_clip", "tiny_open_clip"])
def test_clip_forward_hidden(pkg, ctx, dtype, which):
    ocfg = OCL.tiny_clip_config() if which == "tiny_clip" else OCL.tiny_open_clip_config()
    W = OM.to_torch(OC.synth_weights(OCL.clip_param_specs(ocfg), 5))
    m = pkg.CLIP(ctx, _pcfg(pkg, ocfg), dtype, seed=5)
    ids = _ids(2, 77, 1, 49407 if which == "tiny_clip" else 0)
    for hidden_idx in (0, ocfg.n_layer - 1, ocfg.n_layer):
        out = m.forward_hidden(ids, hidden_idx)
        ref = OCL.forward_hidden(ocfg, W, ids, hidden_idx)
        assert out.shape == ref.shape
        e = rel_err(out, ref)
        print(f"{which} dtype={dtype} hidden_idx={hidden_idx} rel_err={e:.3e}")
        assert torch.isfinite(out).all() and e < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1])
def test_clip_forward_hidden_pooled(pkg, ctx, dtype):
    ocfg = OCL.tiny_open_clip_config()
    W = OM.to_torch(OC.synth_weights(OCL.clip_param_specs(ocfg), 6))
    m = pkg.CLIP(ctx, _pcfg(pkg, ocfg), dtype, seed=6)
    ids = _ids(3, 77, 2, 0, eot_at=[4, 76, 30])
    ids[2, 40] = 49407                      # a second eot later in the sequence: argmax takes the FIRST (clip/mod.rs:139-140)
    hidden, pooled = m.forward_hidden_pooled(ids, ocfg.n_layer - 1)
    rh, rp = OCL.forward_hidden_pooled(ocfg, W, ids, ocfg.n_layer - 1)
    eh, ep = rel_err(hidden, rh), rel_err(pooled, rp)
    print(f"dtype={dtype} hidden rel_err={eh:.3e} pooled rel_err={ep:.3e}")
    assert pooled.shape == (3, ocfg.embed_dim) and eh < TOL[dtype] and ep < TOL[dtype]
    # the tap is the same tensor forward_hidden returns
    assert torch.equal(hidden, m.forward_hidden(ids, ocfg.n_layer - 1))


@pytest.mark.parametrize("dtype", [0, 1])
def test_clip_short_sequence_and_single_batch(pkg, ctx, dtype):
    ocfg = OCL.tiny_clip_config()
    W = OM.to_torch(OC.synth_weights(OCL.clip_param_specs(ocfg), 7))
    m = pkg.CLIP(ctx, _pcfg(pkg, ocfg), dtype, seed=7)
    for n, seq in ((1, 77), (1, 16), (4, 9), (9, 64)):        # ragged key counts, more than 8 sequences (pooling GEMV chunks)
        ids = _ids(n, seq, 10 + seq, 49407)
        h, p = m.forward_hidden_pooled(ids, 1)
        rh, rp = OCL.forward_hidden_pooled(ocfg, W, ids, 1)
        assert rel_err(h, rh) < TOL[dtype] and rel_err(p, rp) < TOL[dtype], (n, seq)


@pytest.mark.parametrize("dtype", [0, 1])
def test_clip_is_causal_and_batch_independent(pkg, ctx, dtype):
    ocfg = OCL.tiny_clip_config()
    m = pkg.CLIP(ctx, _pcfg(pkg, ocfg), dtype, seed=8)
    ids = _ids(2, 77, 3, 49407, eot_at=[20, 50])
    a = m.forward_hidden(ids, ocfg.n_layer)
    ids2 = ids.clone(); ids2[0, 30] = 1234
    b = m.forward_hidden(ids2, ocfg.n_layer)
    assert torch.equal(a[0, :30], b[0, :30]) and not torch.equal(a[0, 30:], b[0, 30:])     # decoder mask, bit-exact prefix
    assert torch.equal(a[1], b[1])
    assert torch.equal(m.forward_hidden(ids[1:], ocfg.n_layer)[0], a[1])                   # same rows alone or in a batch


def test_clip_host_weights_equal_synthetic(pkg, ctx):
    ocfg = OCL.tiny_open_clip_config()
    pc = _pcfg(pkg, ocfg)
    specs = pkg.clip_param_specs(pc)
    flat = pkg.flatten_weights(specs, OC.synth_weights(OCL.clip_param_specs(ocfg), 9))
    ids = _ids(2, 77, 4, 0)
    ha, pa = pkg.CLIP(ctx, pc, 1, weights=flat).forward_hidden_pooled(ids, 2)
    hb, pb = pkg.CLIP(ctx, pc, 1, seed=9).forward_hidden_pooled(ids, 2)
    assert torch.equal(ha, hb) and torch.equal(pa, pb)


def test_clip_errors_are_reported(pkg, ctx):
    ocfg = OCL.tiny_clip_config()
    m = pkg.CLIP(ctx, _pcfg(pkg, ocfg), 1, seed=1)
    with pytest.raises(pkg.EngineError):
        m.forward_hidden(torch.zeros(1, 78, dtype=torch.int64), 1)          # longer than the position table
    with pytest.raises(pkg.EngineError):
        m.forward_hidden(torch.zeros(1, 77, dtype=torch.int64), ocfg.n_layer + 1)
    with pytest.raises(pkg.EngineError):
        m.forward_hidden(torch.full((1, 77), 49408, dtype=torch.int64), 1)   # id outside the vocabulary
    with pytest.raises(pkg.EngineError):
        pkg.CLIP(ctx, pkg.CLIPConfig(49408, 96, 96, 2, 77, 1, True), 1)      # 48 channels per head


def test_conditioning_embedding(pkg, ctx):
    pooled = torch.randn(2, 160, generator=torch.Generator().manual_seed(0))
    size, crop, ar = torch.tensor([[1024, 768], [512, 640]]), torch.tensor([[0, 0], [16, 32]]), torch.tensor([[1024, 1024]] * 2)
    out = pkg.conditioning_embedding(ctx, pooled.cuda(), 256, size, crop, ar)
    ref = OM.conditioning_embedding(pooled, 256, size, crop, ar)
    assert out.shape == (2, 160 + 6 * 256)
    assert torch.equal(out[:, :160].cpu(), pooled)
    assert max_abs(out, ref) < 2e-4          # fp32 sin / cos of arguments up to 1024 (same bound as the timestep embedding)


@pytest.mark.parametrize("dtype", [0, 1])
def test_embedder_tokens_to_conditioning(pkg, ctx, dtype):
    c1, c2 = OCL.tiny_clip_config(), OCL.tiny_open_clip_config()
    oe = OCL.Embedder(c1, OM.to_torch(OC.synth_weights(OCL.clip_param_specs(c1), 1)),
                      c2, OM.to_torch(OC.synth_weights(OCL.clip_param_specs(c2), 2)))
    e = pkg.Embedder(ctx, pkg.CLIP(ctx, _pcfg(pkg, c1), dtype, seed=1), pkg.CLIP(ctx, _pcfg(pkg, c2), dtype, seed=2))
    ids_c, ids_o = _ids(1, 77, 5, 49407, eot_at=[9]), _ids(1, 77, 5, 0, eot_at=[9])
    un_c = torch.full((1, 77), 49407, dtype=torch.int64); un_c[0, 0] = 49406       # tokenize_text("") with pad = eot
    un_o = torch.zeros((1, 77), dtype=torch.int64); un_o[0, 0], un_o[0, 1] = 49406, 49407
    size, crop, ar = torch.tensor([[1024, 1024]]), torch.tensor([[0, 0]]), torch.tensor([1024, 1024])
    got = e.tokens_to_conditioning(ids_c, ids_o, un_c, un_o, size, crop, ar)
    ref = oe.tokens_to_conditioning(ids_c, ids_o, un_c, un_o, size, crop, ar)
    assert got.resolution == ref.resolution == (1024, 1024)
    for name in ("context_full", "context_open_clip", "unconditional_context_full", "unconditional_context_open_clip",
                 "channel_context", "channel_context_refiner", "unconditional_channel_context",
                 "unconditional_channel_context_refiner"):
        a, b = getattr(got, name), getattr(ref, name)
        assert tuple(a.shape) == tuple(b.shape), name
        assert rel_err(a, b) < TOL[dtype], (name, rel_err(a, b))
    with pytest.raises(pkg.EngineError):
        e.text_to_conditioning("a cat", size, crop, ar)      # no tokenizer assets on this Embedder: loud, not silent
