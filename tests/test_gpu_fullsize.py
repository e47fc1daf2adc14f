"""Full-size (BASELINE configs[1]) properties of the hot path: SDXL-base architecture, 1024x1024 (latent 128x128), CFG pair.

Accuracy against the oracle at this size lives in test_gpu_baseline_parity.py (committed oracle fixtures for config 1, one
1024^2 UNet::forward, one 1024^2 decode and the 31-step config-2 trajectory); here the HIP path is held to size-independent
properties:
  * batch independence: the CFG pair is one batch-2 forward; entry i must equal a separate batch-1 forward bit for bit;
  * determinism: eager run, hipGraph capture and replay give identical bits;
  * the sampler stays finite over a short CFG trajectory and the decoder returns a full-range u8 image.
"""
import pytest
import torch

from util import rel_err, seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def base_inputs(pkg):
    cfg = pkg.sdxl_base_config()
    x = seeded(2, 4, 128, 128, seed=50)
    ctxt = seeded(2, 77, cfg.context_dim, seed=51)
    y = seeded(2, cfg.adm_in_channels, seed=52)
    t = torch.tensor([999, 333], dtype=torch.int32)
    return cfg, x, t, ctxt, y


def test_fullsize_unet_properties(pkg, ctx, base_inputs):
    cfg, x, t, ctxt, y = base_inputs
    u = pkg.UNet(ctx, cfg, pkg.DTYPE_F16, seed=0)
    outs = [u.forward(x.cuda(), t.cuda(), ctxt.cuda(), y.cuda()).cpu() for _ in range(3)]       # eager, capture, replay
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), "hipGraph replay differs from the eager run"
    for i in range(2):
        one = u.forward(x[i:i + 1].cuda(), t[i:i + 1].cuda(), ctxt[i:i + 1].cuda(), y[i:i + 1].cuda()).cpu()
        assert torch.equal(one[0], outs[0][i]), f"batch entry {i} depends on its batch neighbour"
    # (accuracy at this size is held against the ORACLE in test_gpu_baseline_parity.py::test_unet_forward_1024_matches_oracle)
    # the V^T part of the fused QKV projections: operand-swapped k-loop + direct transposed store (default) against the LDS-staged
    # transposed epilogue (knob off) -- same products, same k order, same affine expression: the whole forward must not move a bit
    del u
    pkg.debug_set("igemm_tsw", 0)
    try:
        u0 = pkg.UNet(ctx, cfg, pkg.DTYPE_F16, seed=0)
        staged = u0.forward(x.cuda(), t.cuda(), ctxt.cuda(), y.cuda()).cpu()
    finally:
        pkg.debug_set("igemm_tsw", 1)
    assert torch.equal(staged, outs[0]), "operand-swapped V^T epilogue changes the result"
    # weight warming (spare workgroups of a launch read a later launch's weights): outs[0] above is the recording forward (no warming),
    # outs[1] / outs[2] the captured graph WITH the warming workgroups; and with the knob off nothing may move either
    del u0
    pkg.debug_set("igemm_warm", 0)
    try:
        u1 = pkg.UNet(ctx, cfg, pkg.DTYPE_F16, seed=0)
        cold = [u1.forward(x.cuda(), t.cuda(), ctxt.cuda(), y.cuda()).cpu() for _ in range(2)]
    finally:
        pkg.debug_set("igemm_warm", 1)
    assert torch.equal(cold[0], outs[0]) and torch.equal(cold[1], outs[0]), "weight warming changes the result"


@pytest.mark.parametrize("dtype_name", ["F32_SPLIT", "F32_SPLIT_MIX", "F32_SPLIT_MIX_F16W", "F32_SPLIT_MIX_F16W_GEGLU2", "F32_SPLIT_F16W"])
def test_fullsize_split_modes_batch_independence(pkg, ctx, base_inputs, dtype_name):
    """The split-operand engines at full size: an entry of the CFG pair must equal a separate batch-1 forward bit for bit.  Until round 6 the HL16 copy of
    an fp32 stream tensor (skip / up- / down-sampling / proj_out operands) took ONE power-of-two scale from the absmax of the whole batched tensor, so an
    entry's lo halves could depend on its batch neighbour; the scale is now per entry (launch_f32_to_hl_scaled, IgemmParams::a_scale_rpb).
    The two entries below differ by a factor of 64 in magnitude, so a shared scale WOULD change the smaller entry's bits."""
    cfg, x, t, ctxt, y = base_inputs
    x = x.clone()
    x[1] *= 64.0
    dt = getattr(pkg, "DTYPE_" + dtype_name)
    u = pkg.UNet(ctx, cfg, dt, seed=pkg.SEED_F16_WEIGHTS if "F16W" in dtype_name else 0)
    outs = [u.forward(x.cuda(), t.cuda(), ctxt.cuda(), y.cuda()).cpu() for _ in range(3)]       # eager, capture, replay
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), "hipGraph replay differs from the eager run"
    for i in range(2):
        one = u.forward(x[i:i + 1].cuda(), t[i:i + 1].cuda(), ctxt[i:i + 1].cuda(), y[i:i + 1].cuda()).cpu()
        assert torch.equal(one[0], outs[0][i]), f"{dtype_name}: batch entry {i} depends on its batch neighbour"


def test_fullsize_f16w_fused_split_cross_attention_matches_the_attention_kernel(pkg, ctx, base_inputs):
    """SDXL_DTYPE_F32_SPLIT_MIX_F16W runs the 77-key cross-attention at split precision INSIDE the f16 query projection's epilogue (MIX class 512,
    IgemmParams::xa_k_lo).  Without that class the same q (fp32) goes through memory into the stand-alone split-operand attention kernel: the same
    three-MFMA arithmetic, the same single rounding of the result to f16 rows -- the two forwards may differ where an fp32 value sits on an f16 rounding
    boundary, nowhere else."""
    cfg, x, t, ctxt, y = base_inputs
    outs = {}
    try:
        for m in (447, 959):
            pkg.debug_set("mix_classes", m)
            u = pkg.UNet(ctx, cfg, pkg.DTYPE_F32_SPLIT_MIX, seed=pkg.SEED_F16_WEIGHTS)
            assert u.mix_classes() == m
            o = [u.forward(x.cuda(), t.cuda(), ctxt.cuda(), y.cuda()).cpu() for _ in range(3)]
            assert torch.equal(o[0], o[1]) and torch.equal(o[1], o[2]), "hipGraph replay differs from the eager run"
            outs[m] = o[0]
            del u
    finally:
        pkg.debug_set("mix_classes", -1)
    e = rel_err(outs[959], outs[447])
    print(f"F16W forward 1024^2: fused split-precision cross-attention vs attention kernel: rel diff {e:.3e}")
    # (op level the two forms agree except for 0.2 % one-ulp boundary flips of the f16 rows -- test_ln_query_cross_attention_split_precision; 70 transformer
    #  blocks of a random-weight UNet amplify those to ~1.6e-4 of the output, the size of the mode's own distance from the oracle: this bar catches gross errors only)
    assert e < 4e-4


def _prompt_ids(seed, n_tok, pad):
    g = torch.Generator().manual_seed(seed)
    ids = torch.full((1, 77), pad, dtype=torch.int64)
    ids[0, 0] = 49406
    ids[0, 1:1 + n_tok] = torch.randint(256, 49000, (n_tok,), generator=g)
    ids[0, 1 + n_tok] = 49407
    return ids


def test_fullsize_embedder(pkg, ctx):
    """SURVEY 8f row 1 at SDXL size: CLIP ViT-L/14 text (12 blocks x 768, QuickGELU) against the CPU oracle, OpenCLIP bigG
    (32 blocks x 1280, GELU) fp16 against this engine's strict-fp32 mode (itself oracle-checked on the tiny configs), the
    decoder mask as a bit-exact prefix property, and the Conditioning shapes the base / refiner UNets expect."""
    from oracle import clip as OCL, config as OC, model as OM
    lcfg = OCL.clip_l_config()
    W = OM.to_torch(OC.synth_weights(OCL.clip_param_specs(lcfg), 11))
    clip_l = pkg.CLIP(ctx, pkg.clip_l_config(), pkg.DTYPE_F16, seed=11)
    ids = _prompt_ids(1, 12, 49407)
    out = clip_l.forward_hidden(ids, lcfg.n_layer - 1)
    e = rel_err(out, OCL.forward_hidden(lcfg, W, ids, lcfg.n_layer - 1))
    print(f"CLIP-L f16 vs fp32 oracle rel err {e:.3e}")
    assert e < 3.6e-3                                   # measured 1.8e-3
    del W

    ids_o = _prompt_ids(1, 12, 0)
    bigg = pkg.CLIP(ctx, pkg.open_clip_bigg_config(), pkg.DTYPE_F16, seed=12)
    h16, p16 = bigg.forward_hidden_pooled(ids_o, 31)
    assert h16.shape == (1, 77, 1280) and p16.shape == (1, 1280) and torch.isfinite(h16).all() and torch.isfinite(p16).all()
    ids2 = ids_o.clone(); ids2[0, 40] = 777
    h2, p2 = bigg.forward_hidden_pooled(ids2, 31)
    assert torch.equal(h2[0, :40], h16[0, :40]) and not torch.equal(h2[0, 40:], h16[0, 40:])
    assert torch.equal(p2, p16)                           # pooled row = first eot (index 13), upstream of the edit
    ocfg = OCL.open_clip_bigg_config()
    Wo = OM.to_torch(OC.synth_weights(OCL.clip_param_specs(ocfg), 12))
    h32, p32 = OCL.forward_hidden_pooled(ocfg, Wo, ids_o, 31)
    eh, ep = rel_err(h16, h32), rel_err(p16, p32)
    print(f"OpenCLIP bigG f16 vs fp32 oracle: hidden {eh:.3e} pooled {ep:.3e}")
    assert eh < 5.7e-3 and ep < 3.3e-3                   # measured 2.8e-3 / 1.6e-3
    del Wo

    emb = pkg.Embedder(ctx, clip_l, bigg)
    un_c = torch.full((1, 77), 49407, dtype=torch.int64); un_c[0, 0] = 49406
    un_o = torch.zeros((1, 77), dtype=torch.int64); un_o[0, 0], un_o[0, 1] = 49406, 49407
    size, crop, ar = torch.tensor([[1024, 1024]]), torch.tensor([[0, 0]]), torch.tensor([1024, 1024])
    first = emb.tokens_to_conditioning(ids, ids_o, un_c, un_o, size, crop, ar)        # eager
    cond = emb.tokens_to_conditioning(ids, ids_o, un_c, un_o, size, crop, ar)         # captures the hipGraphs
    assert torch.equal(first.context_full, cond.context_full) and torch.equal(first.channel_context, cond.channel_context)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    cond = emb.tokens_to_conditioning(ids, ids_o, un_c, un_o, size, crop, ar)
    t1.record(); torch.cuda.synchronize()
    print(f"Embedder (2 prompts x (CLIP-L + bigG), f16, graph replay): {t0.elapsed_time(t1):.2f} ms")
    assert torch.equal(first.context_full, cond.context_full) and torch.equal(first.channel_context_refiner, cond.channel_context_refiner)
    base, refiner = pkg.sdxl_base_config(), pkg.sdxl_refiner_config()
    assert cond.context_full.shape == (1, 77, base.context_dim) and cond.unconditional_context_full.shape == (77, base.context_dim)
    assert cond.channel_context.shape == (1, base.adm_in_channels) and cond.unconditional_channel_context.shape == (base.adm_in_channels,)
    assert cond.context_open_clip.shape == (1, 77, refiner.context_dim)
    assert cond.channel_context_refiner.shape == (1, refiner.adm_in_channels)
    assert torch.equal(cond.context_full[..., 768:], cond.context_open_clip)


def test_fullsize_sample_and_decode(pkg, ctx):
    """text ids -> Embedder -> Diffuser::sample_latent -> LatentDecoder::latent_to_image, everything at SDXL size"""
    cfg = pkg.sdxl_base_config()
    emb = pkg.Embedder(ctx, pkg.CLIP(ctx, pkg.clip_l_config(), pkg.DTYPE_F16, seed=11),
                       pkg.CLIP(ctx, pkg.open_clip_bigg_config(), pkg.DTYPE_F16, seed=12))
    un_c = torch.full((1, 77), 49407, dtype=torch.int64); un_c[0, 0] = 49406
    un_o = torch.zeros((1, 77), dtype=torch.int64); un_o[0, 0], un_o[0, 1] = 49406, 49407
    cond = emb.tokens_to_conditioning(_prompt_ids(1, 12, 49407), _prompt_ids(1, 12, 0), un_c, un_o,
                                      torch.tensor([[1024, 1024]]), torch.tensor([[0, 0]]), torch.tensor([1024, 1024]))
    del emb
    d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F16, seed=0)
    g = torch.Generator(device="cuda").manual_seed(7)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
    noise = r(1, 4, 128, 128)
    assert pkg.step_count(4) == 4                        # step_size = 1000 / 4 -> t = 999, 749, 499, 249
    lat = d.sample_latent(cond, 7.5, 4, noise)
    assert lat.shape == (1, 4, 128, 128) and torch.isfinite(lat).all()
    lat2 = d.sample_latent(cond, 7.5, 4, noise)
    assert torch.equal(lat, lat2), "trajectory is not deterministic"
    ld = pkg.LatentDecoder(ctx, None, pkg.DTYPE_F16, seed=0)
    img = ld.latent_to_image(lat)
    buf = img.buffer
    assert buf.dtype == torch.uint8 and buf.numel() == 1024 * 1024 * 3
    assert int(buf.max()) > int(buf.min()), "constant image"


def test_fullsize_refiner_and_inpainting(pkg, ctx):
    """BASELINE configs[3] / configs[4] at full size: the 4-level refiner UNet (model_channels 384, C up to 1536 -> 24
    LayerNorm slots, context 1280) through refine_latent, and the inpainting path (u8 image -> VAE encoder -> masked
    trajectory).  Properties: finite, deterministic, the inpainting blend keeps the reference outside the mask at t -> 0."""
    rcfg = pkg.sdxl_refiner_config()
    d = pkg.Diffuser(ctx, rcfg, pkg.DTYPE_F16, seed=0)
    g = torch.Generator(device="cuda").manual_seed(9)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
    cond = pkg.Conditioning(context_open_clip=r(1, 77, rcfg.context_dim), channel_context_refiner=r(1, rcfg.adm_in_channels),
                            unconditional_context_open_clip=r(77, rcfg.context_dim),
                            unconditional_channel_context_refiner=r(rcfg.adm_in_channels), resolution=(1024, 1024))
    latent, noise = r(1, 4, 128, 128), r(1, 4, 128, 128)
    out = d.refine_latent(latent, cond, 7.5, 800, 20, noise)       # (0..200).rev().step_by(50) -> 4 refiner iterations
    assert out.shape == latent.shape and torch.isfinite(out).all()
    assert torch.equal(out, d.refine_latent(latent, cond, 7.5, 800, 20, noise))
    del d

    # inpainting: encode a synthetic u8 image, keep generated content only in latent rows 0..24 (README: crop rows 0..200 px)
    ld = pkg.LatentDecoder(ctx, None, pkg.DTYPE_F16, seed=0, with_encoder=True)
    img = (torch.rand(1, 1024, 1024, 3, device="cuda", generator=g) * 255).to(torch.uint8)
    ref_latent = ld.image_to_latent(pkg.RawImages(img, 1024, 1024))
    assert ref_latent.shape == (1, 4, 128, 128) and torch.isfinite(ref_latent).all()
    cfg = pkg.sdxl_base_config()
    db = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F16, seed=0)
    condb = pkg.Conditioning(context_full=r(1, 77, cfg.context_dim), channel_context=r(1, cfg.adm_in_channels),
                             unconditional_context_full=r(77, cfg.context_dim),
                             unconditional_channel_context=r(cfg.adm_in_channels), resolution=(1024, 1024))
    iters = pkg.step_count(4)
    mask = torch.zeros(1, 4, 128, 128, dtype=torch.bool, device="cuda")
    mask[:, :, 0:25, :] = True
    step_noise = torch.zeros(iters, 1, 4, 128, 128, device="cuda")     # zero re-noise: the kept region must track the reference
    out = db.sample_latent_with_inpainting(condb, 7.5, 4, ref_latent, mask, r(1, 4, 128, 128), step_noise)
    assert torch.isfinite(out).all()
