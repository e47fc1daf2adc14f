"""Model-level parity through the C ABI: UNet::forward, Diffuser::{sample_latent, refine_latent,
sample_latent_with_inpainting}, LatentDecoder::{decode_latent, latent_to_image, encode_image, image_to_latent}
against the CPU oracle on seeded synthetic weights (tiny architectures of the SDXL family, so the oracle runs in
seconds; full-size properties are in test_gpu_fullsize.py).

Tolerances: BASELINE.json's north_star asks for latents within 1e-3 (per-pixel, fp32) of the CPU reference.
  * DTYPE_F32 (strict-parity mode) is held to 1e-3 absolute on latents after the whole trajectory and 1e-4 relative on
    a single UNet forward.
  * DTYPE_F16 / DTYPE_F16_F32RES are reported against looser, explicitly stated bounds (fp16 operand rounding,
    2^-11 per element, through ~40 layers and 4..8 DDIM steps); the reference's own f16 GPU path (sample/main.rs:122)
    rounds at least as much.  The measured errors are printed (run pytest -s) and recorded in DESIGN.md.
"""
import numpy as np
import pytest
import torch

from oracle import config as OC, model as OM, pipeline as OP
from util import max_abs, rel_err, seeded, to_pkg_cfg, to_pkg_vcfg, unet_weights

pytestmark = pytest.mark.gpu

# f16-operand bounds: <= 2x the largest value measured for the class (profiles/r02_gpu_tests_final.log: forward 1.4e-3 on 16^2 /
# 2.3e-3 on 32^2 inputs; F16_F32RES 1.2e-3), so a 2x regression fails
# dtype 3 = SDXL_DTYPE_F32_SPLIT: fp32 residual stream, GEMM operands as (hi, lo) f16 pairs (3 MFMAs per product), fp32 attention --
# held to the strict mode's bounds
# dtype 4 = SDXL_DTYPE_F32_SPLIT_MIX (round 5): dtype 3 with the self-attention and the GEGLU projection on f16 operands -- between 2 and 3
# dtype 5 = SDXL_DTYPE_F32_SPLIT_MIX_F16W: dtype 4 + QKV projection, self-attention out-projection and FF-out on f16 operands (for f16-representable parameters)
FWD_TOL = {0: 1e-5, 1: 4.5e-3, 2: 2.5e-3, 3: 1e-5, 4: 1.5e-3, 5: 2.0e-3, 6: 2.0e-3, 7: 1e-5}      # 7 = SDXL_DTYPE_F32_SPLIT_F16W: dtype 3's arithmetic on the f16 kernels (f16-representable weights)      # measured 2.3e-6 / 1.4e-3 (16^2), 2.26e-3 (32^2) / 1.2e-3 / 1.8e-6
EPS_TOL = {0: 4e-5, 1: 5.8e-3, 2: 5.5e-3, 3: 4e-5}    # the low-variance per-norm-eps probe (measured 1.2e-5 / 2.9e-3 / 2.7e-3)
LAT_ABS_F32 = 1e-3           # north_star: latents within 1e-3 of the fp32 CPU reference (strict-parity mode)
LAT_REL_F16 = 6.0e-3         # fp16-operand modes: max-abs error relative to max|latent|; measured 3.0e-3 (4 CFG-7.5 steps) and 3.8e-3
                             # (5-step inpainting) on the tiny net -> <= 2x measured


F16W_CLASSES = 1 | 2 | 4 | 8 | 16 | 32 | 128 | 256 | 512     # SDXL_DTYPE_F32_SPLIT_MIX_F16W (capi.hip mix_of): + cross-attention query projection, LayerNorm shadow, fused split-precision cross-attention (round 6)


MIX_CLASSES = 1 | 2 | 1024      # SDXL_DTYPE_F32_SPLIT_MIX: f16 self-attention + GEGLU on f16 activations x (hi, lo) weight pairs along K (round 6)


def weights_for(pkg, ocfg, dtype):
    """(oracle weights, synthetic seed of the engine) for a dtype: SDXL_DTYPE_F32_SPLIT_MIX_F16W is FOR f16-representable parameters (on others the
    engine falls back to F32_SPLIT_MIX's classes), so dtype 5 is tested on the seeded weights rounded to f16 on both sides"""
    W = unet_weights(ocfg)
    if dtype not in (5, 6, 7):
        return W, 0
    return {k: (v if k.endswith(".eps") else v.half().float()) for k, v in W.items()}, pkg.SEED_F16_WEIGHTS


def lat_tol(dtype, ref):
    return LAT_ABS_F32 if dtype in (0, 3, 7) else LAT_REL_F16 * float(ref.abs().max())


def _cond(ocfg, n, res, n_ctx=9, refiner=False, seed=30):
    c = dict(ctx=seeded(n, n_ctx, ocfg.context_dim, seed=seed), uctx=seeded(n_ctx, ocfg.context_dim, seed=seed + 1),
             y=seeded(n, ocfg.adm_in_channels, seed=seed + 2), uy=seeded(ocfg.adm_in_channels, seed=seed + 3))
    if refiner:
        oc = OP.Conditioning(None, c["uctx"], None, c["ctx"], None, c["uy"], None, c["y"], res)
    else:
        oc = OP.Conditioning(c["uctx"], None, c["ctx"], None, c["uy"], None, c["y"], None, res)
    return c, oc


def _pkg_cond(pkg, c, res, refiner=False):
    if refiner:
        return pkg.Conditioning(context_open_clip=c["ctx"].cuda(), channel_context_refiner=c["y"].cuda(),
                                unconditional_context_open_clip=c["uctx"].cuda(),
                                unconditional_channel_context_refiner=c["uy"].cuda(), resolution=res)
    return pkg.Conditioning(context_full=c["ctx"].cuda(), channel_context=c["y"].cuda(),
                            unconditional_context_full=c["uctx"].cuda(), unconditional_channel_context=c["uy"].cuda(),
                            resolution=res)


@pytest.mark.parametrize("dtype", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("which", ["tiny", "tiny_refiner"])
def test_unet_forward(pkg, ctx, dtype, which):
    ocfg = OC.tiny_config() if which == "tiny" else OC.tiny_refiner_config()
    W, wseed = weights_for(pkg, ocfg, dtype)
    B, H, Wd = 2, 16, 16
    x = torch.from_numpy(OC.arb_tensor(B, 4, H, Wd))          # reference probe recipe, bin/test/main.rs:133
    context = torch.from_numpy(OC.arb_tensor(B, 5, ocfg.context_dim))
    y = torch.from_numpy(OC.arb_tensor(B, ocfg.adm_in_channels))
    t = torch.tensor([999, 1], dtype=torch.int32)
    ref = OM.unet_forward(ocfg, W, x, t.long(), context, y)
    specs = pkg.unet_param_specs(to_pkg_cfg(pkg, ocfg))
    flat = pkg.flatten_weights(specs, {k: v.numpy() for k, v in W.items()})
    u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), dtype, weights=flat)
    assert u.mix_classes() == {4: MIX_CLASSES, 5: F16W_CLASSES, 6: F16W_CLASSES | 2048, 7: 4096 | 512 | 256}.get(dtype, 0)
    outs = [u.forward(x.cuda(), t.cuda(), context.cuda(), y.cuda()).cpu() for _ in range(3)]   # eager, capture, replay
    e = rel_err(outs[0], ref)
    print(f"unet_forward[{which}] dtype={dtype}: rel err {e:.3e}")
    print(f"  replay diffs {max_abs(outs[0], outs[1]):.3e} {max_abs(outs[1], outs[2]):.3e}")
    assert e < FWD_TOL[dtype]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), "hipGraph replay differs from eager run"
    # device-side synthetic weights are bit-identical to the oracle's numpy recipe -> identical output
    u2 = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), dtype, seed=wseed)
    out2 = u2.forward(x.cuda(), t.cuda(), context.cuda(), y.cuda()).cpu()
    print(f"  synthetic-vs-host max diff {max_abs(out2, outs[0]):.3e}; replay diffs {max_abs(outs[0], outs[1]):.3e} {max_abs(outs[1], outs[2]):.3e}")
    assert torch.equal(out2, outs[0]), "synthetic device weights differ from the oracle's"


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
def test_unet_forward_per_norm_eps(pkg, ctx, dtype):
    # eps is a per-module value in the reference's dumps (groupnorm/load.rs:19, layernorm/load.rs:17), not a global 1e-5:
    # low-variance inputs make the difference visible.  Every GroupNorm / LayerNorm (stand-alone kernels AND the folded
    # LayerNorm of the f16 mode) must honour the value that travels with the weights.
    ocfg = OC.tiny_config()
    W = unet_weights(ocfg)
    for k in W:
        if k.endswith(".eps"):
            W[k] = torch.tensor([3e-2 if ("norm1" in k or "norm_in" in k) else 1e-3 if "norm3" in k else 1e-6])
    x = torch.from_numpy(OC.arb_tensor(2, 4, 16, 16)) * 0.05          # small activations: eps matters
    context = torch.from_numpy(OC.arb_tensor(2, 5, ocfg.context_dim))
    y = torch.from_numpy(OC.arb_tensor(2, ocfg.adm_in_channels))
    t = torch.tensor([999, 1], dtype=torch.int32)
    ref = OM.unet_forward(ocfg, W, x, t.long(), context, y)
    W0 = {k: (torch.tensor([1e-5]) if k.endswith(".eps") else v) for k, v in W.items()}
    assert rel_err(OM.unet_forward(ocfg, W0, x, t.long(), context, y), ref) > 10 * FWD_TOL[dtype], "eps change is invisible"
    specs = pkg.unet_param_specs(to_pkg_cfg(pkg, ocfg))
    u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), dtype, weights=pkg.flatten_weights(specs, {k: v.numpy() for k, v in W.items()}))
    e = rel_err(u.forward(x.cuda(), t.cuda(), context.cuda(), y.cuda()).cpu(), ref)
    print(f"unet_forward per-norm eps dtype={dtype}: rel err {e:.3e}")
    assert e < EPS_TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_split_cfg_chains_equal_batched_pair(pkg, ctx, dtype):
    # sdxl_unet_set_split_cfg (per handle): the two entries of a batch-2 forward as two concurrent batch-1 chains (fork / join inside
    # the captured graph, second chain released after `split_offset` GEMMs) -- same bits as the batched pair, eager and replayed
    ocfg = OC.tiny_config()
    u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), dtype, seed=0)
    x = torch.from_numpy(OC.arb_tensor(2, 4, 16, 16)).cuda()
    context = torch.from_numpy(OC.arb_tensor(2, 5, ocfg.context_dim)).cuda()
    y = torch.from_numpy(OC.arb_tensor(2, ocfg.adm_in_channels)).cuda()
    t = torch.tensor([999, 1], dtype=torch.int32).cuda()
    ref = u.forward(x, t, context, y)
    try:
        for off in (0, 3, 10000):
            u.set_split_cfg(True, off)
            outs = [u.forward(x, t, context, y) for _ in range(3)]      # eager, capture, replay
            assert all(torch.equal(o, ref) for o in outs), off
        one = u.forward(x[:1], t[:1], context[:1], y[:1])                # batch 1 is untouched by the option
        assert torch.equal(one[0], ref[0])
    finally:
        u.set_split_cfg(False)
    assert torch.equal(u.forward(x, t, context, y), ref)


def test_gn_statistics_from_producer_matches_statistics_pass(pkg, ctx):
    # sdxl_unet_set_gn_from_producer (per handle, default on): the option's plumbing (scratch allocation, Act tagging, graph
    # capture) on the tiny net, both settings vs the oracle.  The tiny net's grids are single rounds, where the tile selection
    # prefers 96x128 tiles, so most norms keep their statistics pass here; the fused kernel path itself is covered by
    # test_conv2d_group_norm_statistics_from_producer and by the full-size oracle comparisons (test_gpu_baseline_parity.py).
    ocfg = OC.tiny_config()
    W = unet_weights(ocfg)
    x = torch.from_numpy(OC.arb_tensor(2, 4, 32, 32))
    context = torch.from_numpy(OC.arb_tensor(2, 7, ocfg.context_dim))
    y = torch.from_numpy(OC.arb_tensor(2, ocfg.adm_in_channels))
    t = torch.tensor([999, 1], dtype=torch.int32)
    ref = OM.unet_forward(ocfg, W, x, t.long(), context, y)
    outs = {}
    for on in (True, False):
        u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), 1, seed=0)
        u.set_gn_from_producer(on)
        o = [u.forward(x.cuda(), t.cuda(), context.cuda(), y.cuda()).cpu() for _ in range(3)]   # eager, capture, replay
        assert all(torch.equal(a, o[0]) for a in o)
        outs[on] = o[0]
    e_on, e_off, e_x = rel_err(outs[True], ref), rel_err(outs[False], ref), rel_err(outs[True], outs[False])
    print(f"GN statistics from producer: vs oracle {e_on:.3e}, statistics pass vs oracle {e_off:.3e}, between them {e_x:.3e}")
    assert e_on < FWD_TOL[1] and e_off < FWD_TOL[1] and e_x < 4e-3


def test_fused_cross_attention_matches_two_kernel_path(pkg, ctx):
    # sdxl_unet_set_fused_cross_attention (per handle, default on): attn2 inside the query projection's epilogue against the
    # projection + attention-kernel path of the same engine, and both against the oracle (unet/mod.rs:731-795)
    ocfg = OC.tiny_config()
    W = unet_weights(ocfg)
    u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), 1, seed=0)
    x = torch.from_numpy(OC.arb_tensor(2, 4, 32, 32))
    context = torch.from_numpy(OC.arb_tensor(2, 7, ocfg.context_dim))
    y = torch.from_numpy(OC.arb_tensor(2, ocfg.adm_in_channels))
    t = torch.tensor([999, 1], dtype=torch.int32)
    ref = OM.unet_forward(ocfg, W, x, t.long(), context, y)
    fused = [u.forward(x.cuda(), t.cuda(), context.cuda(), y.cuda()).cpu() for _ in range(3)]   # eager, capture, replay
    assert all(torch.equal(o, fused[0]) for o in fused)
    try:
        u.set_fused_cross_attention(False)
        plain = u.forward(x.cuda(), t.cuda(), context.cuda(), y.cuda()).cpu()
    finally:
        u.set_fused_cross_attention(True)
    assert torch.equal(u.forward(x.cuda(), t.cuda(), context.cuda(), y.cuda()).cpu(), fused[0])
    e_f, e_p, e_fp = rel_err(fused[0], ref), rel_err(plain, ref), rel_err(fused[0], plain)
    print(f"fused cross-attention: vs oracle {e_f:.3e}, two-kernel vs oracle {e_p:.3e}, fused vs two-kernel {e_fp:.3e}")
    assert e_f < FWD_TOL[1] and e_p < FWD_TOL[1] and e_fp < 4e-3


@pytest.mark.parametrize("dtype", [0, 1, 2, 3, 4, 5, 6, 7])
def test_unet_forward_batch_independence(pkg, ctx, dtype):
    # the engine batches the CFG pair; per-sample results must not depend on what else is in the batch
    ocfg = OC.tiny_config()
    u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), dtype, seed=weights_for(pkg, ocfg, dtype)[1])
    x, c, y = seeded(2, 4, 8, 8, seed=1), seeded(2, 5, ocfg.context_dim, seed=2), seeded(2, ocfg.adm_in_channels, seed=3)
    t = torch.tensor([500, 20], dtype=torch.int32)
    both = u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
    for i in range(2):
        one = u.forward(x[i:i + 1].cuda(), t[i:i + 1].cuda(), c[i:i + 1].cuda(), y[i:i + 1].cuda()).cpu()
        assert torch.equal(one[0], both[i])


@pytest.mark.parametrize("dtype", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("n,n_steps,cfg_scale", [(1, 4, 7.5), (2, 8, 1.0)])
def test_sample_latent(pkg, ctx, dtype, n, n_steps, cfg_scale):
    ocfg = OC.tiny_config()
    res = (64, 96)
    c, oc = _cond(ocfg, n, res)
    noise = seeded(n, 4, res[0] // 8, res[1] // 8, seed=40)
    trace = []
    W, wseed = weights_for(pkg, ocfg, dtype)
    ref = OP.Diffuser(ocfg, W, OC.alphas_cumprod()).sample_latent(oc, cfg_scale, n_steps, noise, trace)
    assert len(trace) == pkg.step_count(n_steps)
    d = pkg.Diffuser(ctx, to_pkg_cfg(pkg, ocfg), dtype, seed=wseed)
    out = d.sample_latent(_pkg_cond(pkg, c, res), cfg_scale, n_steps, noise.cuda()).cpu()
    e = max_abs(out, ref)
    print(f"sample_latent n={n} steps={n_steps} dtype={dtype}: latent max-abs err {e:.3e} (|latent| max {ref.abs().max():.2f})")
    assert np.isfinite(e) and e < lat_tol(dtype, ref)
    out2 = d.sample_latent(_pkg_cond(pkg, c, res), cfg_scale, n_steps, noise.cuda()).cpu()
    assert torch.equal(out, out2), "trajectory is not deterministic"


@pytest.mark.parametrize("dtype", [0, 1, 3, 4, 5, 6, 7])
def test_refine_latent(pkg, ctx, dtype):
    ocfg = OC.tiny_refiner_config()
    res = (64, 64)
    c, oc = _cond(ocfg, 1, res, refiner=True)
    latent, noise = seeded(1, 4, 8, 8, seed=41), seeded(1, 4, 8, 8, seed=42)
    W, wseed = weights_for(pkg, ocfg, dtype)
    ref = OP.Diffuser(ocfg, W, OC.alphas_cumprod()).refine_latent(latent, oc, 7.5, 800, 50, noise)
    d = pkg.Diffuser(ctx, to_pkg_cfg(pkg, ocfg), dtype, seed=wseed)
    out = d.refine_latent(latent.cuda(), _pkg_cond(pkg, c, res, True), 7.5, 800, 50, noise.cuda()).cpu()
    e = max_abs(out, ref)
    print(f"refine_latent dtype={dtype}: max-abs err {e:.3e}")
    assert e < lat_tol(dtype, ref)


@pytest.mark.parametrize("dtype", [0, 1, 3, 4, 5, 6, 7])
def test_sample_latent_with_inpainting(pkg, ctx, dtype):
    ocfg = OC.tiny_config()
    res = (64, 64)
    n_steps = 5
    iters = pkg.step_count(n_steps)
    c, oc = _cond(ocfg, 1, res)
    noise0, reference = seeded(1, 4, 8, 8, seed=43), seeded(1, 4, 8, 8, seed=44)
    step_noise = seeded(iters, 1, 4, 8, 8, seed=45)
    mask = torch.zeros(1, 4, 8, 8, dtype=torch.bool)
    mask[:, :, 0:3, :] = True      # crop rows in latent coords, broadcast to 4 channels (sample/main.rs:164-185)
    W, wseed = weights_for(pkg, ocfg, dtype)
    ref = OP.Diffuser(ocfg, W, OC.alphas_cumprod()).sample_latent_with_inpainting(
        oc, 7.5, n_steps, reference, mask, noise0, [step_noise[i] for i in range(iters)])
    d = pkg.Diffuser(ctx, to_pkg_cfg(pkg, ocfg), dtype, seed=wseed)
    out = d.sample_latent_with_inpainting(_pkg_cond(pkg, c, res), 7.5, n_steps, reference.cuda(), mask.cuda(),
                                          noise0.cuda(), step_noise.cuda()).cpu()
    e = max_abs(out, ref)
    print(f"inpainting dtype={dtype}: max-abs err {e:.3e}")
    assert e < lat_tol(dtype, ref)


@pytest.mark.parametrize("dtype", [0, 1, 3])
def test_vae_decode_and_image(pkg, ctx, dtype):
    v = OC.tiny_vae_config()
    Wd = OM.to_torch(OC.synth_weights(OC.vae_decoder_param_specs(v)))
    old = OP.LatentDecoder(v, Wd)
    latent = seeded(2, 4, 8, 8, seed=50) * 0.5
    ref = old.decode_latent(latent)
    ld = pkg.LatentDecoder(ctx, to_pkg_vcfg(pkg, v), dtype, seed=0)
    out = ld.decode_latent(latent.cuda()).cpu()
    e = rel_err(out, ref)
    print(f"vae decode dtype={dtype}: rel err {e:.3e}")
    assert e < (3.2e-3 if dtype == 1 else 5e-6)          # f16 measured 1.6e-3; exact fp32 1.6e-6; dtype 3 = split-operand fp32 class
    img = ld.latent_to_image(latent.cuda())
    assert (img.width, img.height) == (64, 64)
    ref8 = old.latent_to_image(latent)
    d8 = np.abs(img.buffer.cpu().numpy().astype(np.int32) - ref8.astype(np.int32))
    # truncating u8 cast: a float error of 1e-5 can flip a value sitting on an integer boundary
    assert d8.max() <= (8 if dtype == 1 else 1)
    assert (d8 > 0).mean() < (0.5 if dtype == 1 else 0.01)


@pytest.mark.parametrize("dtype", [0, 1, 3])
def test_vae_encode(pkg, ctx, dtype):
    v = OC.tiny_vae_config()
    We = OM.to_torch(OC.synth_weights(OC.vae_encoder_param_specs(v)))
    old = OP.LatentDecoder(v, We)
    img = (seeded(1, 32, 48, 3, seed=51).clamp(-2, 2) * 60 + 128).clamp(0, 255).to(torch.uint8)
    ref = old.image_to_latent(img.numpy())
    ld = pkg.LatentDecoder(ctx, to_pkg_vcfg(pkg, v), dtype, seed=0, with_encoder=True)
    out = ld.image_to_latent(pkg.RawImages(img.cuda(), 48, 32)).cpu()
    e = rel_err(out, ref)
    print(f"vae encode dtype={dtype}: rel err {e:.3e}")
    assert out.shape == ref.shape and e < (3.5e-3 if dtype == 1 else 5e-6)   # f16 measured 1.7e-3; fp32 classes 1.8e-6
    x = torch.from_numpy(img.numpy().astype(np.float32) / 255.0).permute(0, 3, 1, 2) * 2 - 1
    out2 = ld.encode_image(x.cuda()).cpu()
    assert rel_err(out2, ref) < (3.5e-3 if dtype == 1 else 5e-6)


@pytest.mark.parametrize("dtype", [0, 1])
def test_against_committed_golden_fixture(pkg, ctx, dtype):
    # tests/golden/tiny_unet_arb.npz (oracle/make_golden.py): reference probe inputs arb_tensor = sin(arange)
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_unet_arb.npz"))
    ocfg = OC.tiny_config()
    u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), dtype, seed=0)
    x = torch.from_numpy(OC.arb_tensor(1, 4, 8, 8)).cuda()
    c = torch.from_numpy(OC.arb_tensor(1, 1, ocfg.context_dim)).cuda()
    y = torch.from_numpy(OC.arb_tensor(1, ocfg.adm_in_channels)).cuda()
    out = u.forward(x, torch.tensor([1], dtype=torch.int32).cuda(), c, y).cpu()
    assert rel_err(out, torch.from_numpy(g["unet_out"])) < FWD_TOL[dtype]
    v = OC.tiny_vae_config()
    ld = pkg.LatentDecoder(ctx, to_pkg_vcfg(pkg, v), dtype, seed=0, with_encoder=True)
    # Decoder::forward alone (no 1/scale_factor, no post_quant) is not exported; check the trajectory fixture instead
    d = pkg.Diffuser(ctx, to_pkg_cfg(pkg, ocfg), dtype, seed=0)
    cond = pkg.Conditioning(context_full=c, channel_context=y, unconditional_context_full=c[0],
                            unconditional_channel_context=y[0] * 0.5, resolution=(64, 64))
    lat = d.sample_latent(cond, 1.0, 4, torch.from_numpy(OC.arb_tensor(1, 4, 8, 8)).cuda()).cpu()
    ref = torch.from_numpy(g["traj"][-1])
    assert max_abs(lat, ref) < lat_tol(dtype, ref)
    del ld


def test_f16w_mode_falls_back_on_parameters_that_are_not_f16_values(pkg, ctx):
    # SDXL_DTYPE_F32_SPLIT_MIX_F16W packs six more transformer classes as plain f16: on parameters that are not f16 values that would round the
    # WEIGHTS too and leave the mode's error bound.  The engine checks the tensors at create time and falls back to F32_SPLIT_MIX's two
    # classes: same bits as dtype 4, and sdxl_unet_mix_classes says so.  One perturbed weight is enough.
    ocfg = OC.tiny_config()
    cfg = to_pkg_cfg(pkg, ocfg)
    x = torch.from_numpy(OC.arb_tensor(2, 4, 16, 16)).cuda()
    c = torch.from_numpy(OC.arb_tensor(2, 5, ocfg.context_dim)).cuda()
    y = torch.from_numpy(OC.arb_tensor(2, ocfg.adm_in_channels)).cuda()
    t = torch.tensor([999, 1], dtype=torch.int32).cuda()
    u5, u4 = pkg.UNet(ctx, cfg, 5, seed=0), pkg.UNet(ctx, cfg, 4, seed=0)
    assert u5.mix_classes() == MIX_CLASSES == u4.mix_classes()
    assert torch.equal(u5.forward(x, t, c, y), u4.forward(x, t, c, y))
    assert pkg.UNet(ctx, cfg, 5, seed=pkg.SEED_F16_WEIGHTS).mix_classes() == F16W_CLASSES
    assert pkg.UNet(ctx, cfg, 6, seed=pkg.SEED_F16_WEIGHTS).mix_classes() == F16W_CLASSES | 2048 and pkg.UNet(ctx, cfg, 6, seed=0).mix_classes() == MIX_CLASSES
    assert pkg.UNet(ctx, cfg, 7, seed=pkg.SEED_F16_WEIGHTS).mix_classes() == 4096 | 512 | 256 and pkg.UNet(ctx, cfg, 7, seed=0).mix_classes() == 0
    specs = pkg.unet_param_specs(cfg)
    W16 = {k: (v if k.endswith(".eps") else v.half().float()).numpy().copy() for k, v in unet_weights(ocfg).items()}
    assert pkg.UNet(ctx, cfg, 5, weights=pkg.flatten_weights(specs, W16)).mix_classes() == F16W_CLASSES
    name = next(k for k in W16 if k.endswith(".mlp.lin.weight"))
    W16[name].flat[3] = np.float32(W16[name].flat[3]) * np.float32(1.0 + 2.0 ** -16)      # one value that is not an f16
    assert pkg.UNet(ctx, cfg, 5, weights=pkg.flatten_weights(specs, W16)).mix_classes() == MIX_CLASSES
    for dt in (0, 1, 2, 3):
        assert pkg.UNet(ctx, cfg, dt, seed=0).mix_classes() == 0


def test_host_weights_equal_synthetic(pkg, ctx):
    # sdxl_vae_create (flat host buffer in sdxl_vae_param_spec order) == sdxl_vae_create_synthetic
    v = OC.tiny_vae_config()
    vc = to_pkg_vcfg(pkg, v)
    W = OC.synth_weights(OC.vae_decoder_param_specs(v))
    flat = pkg.flatten_weights(pkg.vae_param_specs(vc, False), W)
    a = pkg.LatentDecoder(ctx, vc, 0, decoder_weights=flat)
    b = pkg.LatentDecoder(ctx, vc, 0, seed=0)
    latent = seeded(1, 4, 4, 4, seed=52).cuda()
    oa, ob, oa2 = a.decode_latent(latent), b.decode_latent(latent), a.decode_latent(latent)
    print(f"host-vs-synth diff {max_abs(oa, ob):.3e}, same-object rerun diff {max_abs(oa, oa2):.3e}")
    assert torch.equal(oa, ob)


def test_reference_npy_tree_loads_into_the_engine(pkg, ctx, tmp_path):
    # the reference's params/ tree (python/save.py format) -> importer -> sdxl_unet_create == synthetic weights of the same seed
    import importlib
    imp = importlib.import_module(pkg.__name__ + ".importer")
    ocfg = OC.tiny_config()
    specs = OC.unet_param_specs(ocfg)
    root = str(tmp_path / "diffuser_base")
    imp.export_tree(specs, OC.synth_weights(specs, 0), root, "unet")
    kind = {"Conv": 0, "Res": 1, "Down": 2, "ResT": 3, "ResTU": 4, "ResU": 5}
    inp, mid, out = OC.unet_block_plan(ocfg)
    blocks = lambda bl: [(kind[b["kind"]], b.get("depth", 0), b.get("n_head", 0)) for b in bl]   # noqa: E731
    imp.export_unet_structure(root, ocfg.model_channels, blocks(inp), blocks(out), mid["depth"], mid["n_head"])
    cfg, flat = imp.load_unet(pkg, root)
    x = torch.from_numpy(OC.arb_tensor(1, 4, 8, 8)).cuda()
    t = torch.tensor([7], dtype=torch.int32).cuda()
    c = torch.from_numpy(OC.arb_tensor(1, 5, ocfg.context_dim)).cuda()
    y = torch.from_numpy(OC.arb_tensor(1, ocfg.adm_in_channels)).cuda()
    a = pkg.UNet(ctx, cfg, 1, weights=flat).forward(x, t, c, y)
    b = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), 1, seed=0).forward(x, t, c, y)
    assert torch.equal(a, b)


def test_errors_are_reported_not_fatal(pkg, ctx):
    bad = pkg.UNetConfig(128, 48, [1, 2, 4], 64, [0, 1, 2], 128)     # 48 % 64 != 0 -> reference asserts (unet/mod.rs:73-76)
    with pytest.raises(pkg.EngineError):
        pkg.UNet(ctx, bad, 0, seed=0)
    ocfg = OC.tiny_config()
    u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), 0, seed=0)
    with pytest.raises(pkg.EngineError):   # height not divisible by 4
        u.forward(torch.zeros(1, 4, 6, 8).cuda(), torch.zeros(1, dtype=torch.int32).cuda(),
                  torch.zeros(1, 3, ocfg.context_dim).cuda(), torch.zeros(1, ocfg.adm_in_channels).cuda())


@pytest.mark.parametrize("dtype", [1, 0, 3])
def test_empty_replica_receives_weight_arena(pkg, ctx, dtype):
    """the multi-GPU replica path on one GPU: a model created `empty` (identical arena layout, no contents) must reproduce
    the source model bit for bit once the packed arena has been copied in through the zero-copy tensor view -- exactly what
    bench.py's RCCL broadcast does (folded-LayerNorm weights, column sums and biases all live in the arena)"""
    ocfg = OC.tiny_config()
    cfg = to_pkg_cfg(pkg, ocfg)
    res = (64, 64)
    c, _ = _cond(ocfg, 1, res)
    noise = seeded(1, 4, 8, 8, seed=46)
    src = pkg.Diffuser(ctx, cfg, dtype, seed=0)
    dst = pkg.Diffuser(ctx, cfg, dtype, seed=0, empty=True)
    a_src, a_dst = src.diffusion.weight_arena_tensor(), dst.diffusion.weight_arena_tensor()
    base, nbytes = src.diffusion.weight_arena()
    assert a_src.data_ptr() == base and a_src.numel() == nbytes and a_src.dtype == torch.uint8, "arena view is not zero-copy"
    assert a_dst.numel() == a_src.numel(), "replica arena layout differs from the source's"
    a_dst.copy_(a_src)
    torch.cuda.synchronize()
    ref = src.sample_latent(_pkg_cond(pkg, c, res), 7.5, 4, noise.cuda()).cpu()
    out = dst.sample_latent(_pkg_cond(pkg, c, res), 7.5, 4, noise.cuda()).cpu()
    assert torch.equal(out, ref)


def test_library_comm_single_rank(pkg, ctx):
    # the library's own RCCL communicator (csrc/comm.cpp, bound with dlopen): a world of one rank must initialise next to
    # torch's RCCL in the same process, and the weight broadcasts degenerate to no-ops that leave the arena untouched.
    # (N > 1 needs N GPUs: the schedule itself is executed over gloo in tests/test_cpu_distributed.py.)
    ocfg = OC.tiny_config()
    u = pkg.UNet(ctx, to_pkg_cfg(pkg, ocfg), 1, seed=0)
    before = u.weight_arena_tensor().clone()
    comm = pkg.Comm(0, 0, 1, pkg.Comm.unique_id())
    comm.bcast_unet(u)
    buf = torch.arange(1000, dtype=torch.float32, device="cuda")
    comm.bcast_buffer(buf)
    assert torch.equal(u.weight_arena_tensor(), before) and torch.equal(buf.cpu(), torch.arange(1000, dtype=torch.float32))
    assert pkg.bcast_plan(10_000_000, 8, 3) == (3 * 1249792, 1249792, 8 * 1249792, 10_000_000 - 8 * 1249792)


def test_vae_split_operand_token_count_not_a_multiple_of_8(pkg, ctx):
    # a 72 x 72 image: 9 x 9 latent, 81 tokens in the mid-block attention -- HL16 outputs are written as whole 8-key pieces, so the
    # split-operand V^T goes out as fp32 and is converted (was: a runtime error while f32 / f16 decoded the same latent)
    v = OC.tiny_vae_config()
    Wd = OM.to_torch(OC.synth_weights(OC.vae_decoder_param_specs(v)))
    latent = seeded(1, 4, 9, 9, seed=52) * 0.5
    ref = OP.LatentDecoder(v, Wd).decode_latent(latent)
    out = pkg.LatentDecoder(ctx, to_pkg_vcfg(pkg, v), 3, seed=0).decode_latent(latent.cuda()).cpu()
    e = rel_err(out, ref)
    print(f"vae decode 9x9 latent, split-operand: rel err {e:.3e}")
    assert e < 5e-6
