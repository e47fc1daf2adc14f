"""Parity at the BASELINE configurations (BASELINE.json configs[0] / configs[1]), full SDXL-base + SDXL-VAE architectures.

The targets are oracle outputs committed under tests/golden/fullsize_*.npz (oracle/make_golden_fullsize.py: the fp32 CPU
restatement of the reference graph run in the build container on the seeded synthetic weights the HIP fill kernel reproduces
bit for bit).  `SDXL_LIVE_ORACLE=1` additionally re-runs the oracle for config 1 on this box's host cores (1-2 minutes +
10 GB of fp32 weights) and checks the committed fixture against it.

What is compared, and to which bar:
  * SDXL_DTYPE_F32 (strict-parity mode, exact-fp32 MFMA) against the oracle:
      - config 1: SDXL-base, 512x512, n_steps=4, CFG 1.0 -- every per-step latent, the final latent, decode_latent and
        the u8 image (flow of src/bin/sample/main.rs:239-278 / stablediffusion/mod.rs:390-432);
      - one UNet::forward at 1024x1024 (unet/mod.rs:450-492);
      - one latent_to_image at 1024x1024 (stablediffusion/mod.rs:200-237);
      - config 2, the configuration bench.py times: 1024x1024, n_steps=30 (31 iterations), CFG 7.5 -- the oracle's own
        31-step trajectory (9 stored steps + final latent).
      - round 3: BASELINE configs[3] / configs[4] at full size -- one refiner UNet::forward, a 2-iteration refine_latent, one
        1024x1024 image_to_latent (Encoder::forward with its PaddedConv2d downsamples) and a 4-step inpainting trajectory
        (fixtures: oracle/make_golden_r3.py), plus one base forward with f16-representable weights (what a real record
        holds) that isolates activation rounding from weight rounding.
    Bar: north_star's 1e-3 on latents at SDXL's latent scale.  The synthetic-weight UNet is not a denoiser: its
    trajectories grow well past real SDXL latents (|x| <~ 4), so the absolute bound is scaled with the trajectory:
    1e-3 * max(1, max|latent_ref| / 4); both the raw absolute and the relative error are recorded.
  * SDXL_DTYPE_F16 / SDXL_DTYPE_F16_F32RES (the speed modes) against the now oracle-anchored F32 engine on the SAME 31-step
    1024x1024 CFG-7.5 trajectory: per-step max-abs / relative / rms error -> gpurun_out/r05_drift_*.json (committed under
    profiles/).  fp16 operands cannot meet 1e-3 absolute (one rounding is 4.9e-4 relative, CFG 7.5 multiplies the error of
    eps by up to 7.5 per step); the bound asserted here is the measured class with headroom, stated in DESIGN.md section 5.
Everything goes through the C ABI (ctypes); measured numbers are written to gpurun_out/r06_parity_baseline.json.
"""
import json
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
OUT_DIR = os.path.join(ROOT, "gpurun_out")
REPORT = {}
SUB = 5
CROPS = ((0, 0), (0, 960), (480, 480), (960, 0))

LAT_ABS = 1e-3               # north_star bar at SDXL's latent scale ...
LAT_SCALE_REF = 4.0          # ... |latent| <~ 4; synthetic trajectories are larger, the bound scales with them
F32_FWD_REL = 1e-4           # one UNet::forward, strict mode
# fp16-operand bounds = <= 2x the values measured on MI355X (profiles/r02_parity_baseline.json): a 2x regression fails
F16_FWD_REL = 2.7e-3         # one base UNet::forward, fp16 operands (measured 1.31e-3)
F16RES_FWD_REL = 2.0e-3      # ... with the fp32 residual stream (measured 9.6e-4)
F16_DECODE_REL = 2.5e-3      # latent_to_image, fp16 VAE (measured 1.1e-3)
F16_TRAJ_REL = {"f16": 3.8e-3, "f16_f32res": 2.3e-3}   # config 2, final latent after 31 CFG-7.5 steps (measured 1.85e-3 / 1.11e-3)
IMG_ABS_F32 = 2e-3           # decode_latent, strict mode: fp32 image in [-1, 1]-ish units


def seeded(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def checksum(*tensors):
    return np.array([float(t.double().sum()) for t in tensors] + [float(t.double().abs().sum()) for t in tensors])


def lat_bound(ref):
    return LAT_ABS * max(1.0, float(ref.abs().max()) / LAT_SCALE_REF)


def errs(out, ref):
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    d = (out - ref)
    return dict(max_abs=float(d.abs().max()), rel=float(d.abs().max() / ref.abs().max().clamp_min(1e-30)),
                rms_rel=float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30)), ref_max=float(ref.abs().max()))


@pytest.fixture(scope="module", autouse=True)
def _write_report():
    yield
    try:
        os.makedirs(OUT_DIR, exist_ok=True)
        path = os.path.join(OUT_DIR, "r06_parity_baseline.json")
        merged = {}
        if os.path.exists(path):          # a partial run (-k ...) updates its own entries only
            try:
                with open(path) as fh:
                    merged = json.load(fh)
            except ValueError:
                merged = {}
        merged.update(REPORT)
        with open(path, "w") as fh:
            json.dump(merged, fh, indent=1)
    except OSError:
        pass


def _cond(pkg, i, res):
    return pkg.Conditioning(context_full=i["ctx"].cuda(), channel_context=i["y"].cuda(),
                            unconditional_context_full=i["uctx"].cuda(), unconditional_channel_context=i["uy"].cuda(),
                            resolution=res)


def _inputs(cfg, base_seed, hw):
    return dict(noise=seeded(1, 4, hw, hw, seed=base_seed + 1), ctx=seeded(1, 77, cfg.context_dim, seed=base_seed + 2),
                uctx=seeded(77, cfg.context_dim, seed=base_seed + 3), y=seeded(1, cfg.adm_in_channels, seed=base_seed + 4),
                uy=seeded(cfg.adm_in_channels, seed=base_seed + 5))


def test_config1_f32_matches_oracle(pkg, ctx):
    """BASELINE configs[0] end to end: Diffuser::sample_latent (4 steps, CFG 1.0) -> LatentDecoder::latent_to_image"""
    g = np.load(os.path.join(GOLD, "fullsize_config1.npz"))
    cfg = pkg.sdxl_base_config()
    i = _inputs(cfg, 100, 64)
    assert np.allclose(checksum(*i.values()), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F32, seed=0)
    trace = torch.zeros(4, 1, 4, 64, 64, device="cuda")
    d.set_trace(trace)
    t0 = time.time()
    lat = d.sample_latent(_cond(pkg, i, (512, 512)), 1.0, 4, i["noise"].cuda())
    torch.cuda.synchronize()
    dt = time.time() - t0
    d.set_trace(None)
    ref_traj, ref = torch.from_numpy(g["traj"]), torch.from_numpy(g["latent"])
    steps = [errs(trace[k], ref_traj[k]) for k in range(4)]
    fin = errs(lat, ref)
    REPORT["config1_f32_vs_oracle"] = dict(per_step=steps, final=fin, bound=lat_bound(ref), engine_seconds=dt,
                                           oracle_seconds=g["oracle_seconds"].tolist(), oracle_threads=int(g["oracle_threads"][0]))
    print(f"config 1 F32 vs oracle: final latent max-abs {fin['max_abs']:.3e} (rel {fin['rel']:.3e}, |ref| {fin['ref_max']:.1f}); "
          f"per step {['%.2e' % s['max_abs'] for s in steps]}; engine {dt:.2f} s vs oracle {g['oracle_seconds'][0]:.1f} s")
    assert torch.equal(trace[3], lat), "trace of the last step differs from the returned latent"
    for k in range(4):
        assert steps[k]["max_abs"] <= lat_bound(ref_traj[k]), (k, steps[k])
    assert fin["max_abs"] <= lat_bound(ref), fin
    del d
    # decode: the oracle's latent through the strict-mode VAE (f32, as the reference runs it: sample/main.rs:121,271-278)
    ld = pkg.LatentDecoder(ctx, None, pkg.DTYPE_F32, seed=0)
    img = ld.decode_latent(ref.cuda()).cpu()
    ie = errs(img[:, :, ::SUB, ::SUB], torch.from_numpy(g["image_sub"]))
    u8 = ld.latent_to_image(ref.cuda()).buffer.cpu().numpy()
    d8 = np.abs(u8.astype(np.int32) - g["u8"].astype(np.int32))
    # and the engine's OWN latent through the engine's VAE: the whole config-1 job against the oracle's image
    u8e = ld.latent_to_image(lat).buffer.cpu().numpy()
    d8e = np.abs(u8e.astype(np.int32) - g["u8"].astype(np.int32))
    REPORT["config1_decode_f32_vs_oracle"] = dict(image=ie, u8_max_diff=int(d8.max()), u8_frac_diff=float((d8 > 0).mean()),
                                                  end_to_end_u8_max_diff=int(d8e.max()), end_to_end_u8_frac_diff=float((d8e > 0).mean()))
    print(f"config 1 decode F32 vs oracle: image max-abs {ie['max_abs']:.3e} (|ref| {ie['ref_max']:.2f}); u8 max diff {d8.max()} "
          f"({(d8 > 0).mean():.2e} of the bytes); end to end u8 max diff {d8e.max()} ({(d8e > 0).mean():.2e})")
    assert ie["max_abs"] <= IMG_ABS_F32 * max(1.0, ie["ref_max"]), ie
    assert d8.max() <= 1 and d8e.max() <= 1


@pytest.mark.skipif(os.environ.get("SDXL_LIVE_ORACLE") != "1", reason="set SDXL_LIVE_ORACLE=1: re-runs the oracle (1-2 min, 10 GB)")
def test_config1_fixture_matches_live_oracle():
    from oracle import make_golden_fullsize as MG, config as OC, pipeline as OP
    g = np.load(os.path.join(GOLD, "fullsize_config1.npz"))
    cfg, W = MG.base_weights()
    i = MG.config1_inputs(cfg)
    cond = OP.Conditioning(i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], None, (512, 512))
    t0 = time.time()
    lat = OP.Diffuser(cfg, W, OC.alphas_cumprod()).sample_latent(cond, 1.0, 4, i["noise"])
    REPORT["config1_live_oracle"] = dict(seconds=time.time() - t0, threads=torch.get_num_threads(), vs_fixture=errs(lat, torch.from_numpy(g["latent"])))
    assert errs(lat, torch.from_numpy(g["latent"]))["rel"] < 1e-4       # another box's BLAS: fp32 summation order only


def test_unet_forward_1024_matches_oracle(pkg, ctx):
    """one UNet::forward at the benchmarked resolution: strict mode against the oracle, speed modes against both"""
    g = np.load(os.path.join(GOLD, "fullsize_unet1024.npz"))
    cfg = pkg.sdxl_base_config()
    x, t = seeded(1, 4, 128, 128, seed=111), torch.tensor([500], dtype=torch.int32)
    c, y = seeded(1, 77, cfg.context_dim, seed=112), seeded(1, cfg.adm_in_channels, seed=113)
    assert np.allclose(checksum(x, c, y), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    ref = torch.from_numpy(g["out"])
    rep = {}
    outs = {}
    # f32_split_mix: the oracle's f16-operand model of its two f16 classes (attn 1.7e-5, geglu 1.9e-4 -> 2.0e-4 in quadrature), x 1.5
    for name, dt, tol in (("f32", pkg.DTYPE_F32, F32_FWD_REL), ("f32_split", pkg.DTYPE_F32_SPLIT, F32_FWD_REL), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX, 3.0e-4),
                          ("f16", pkg.DTYPE_F16, F16_FWD_REL), ("f16_f32res", pkg.DTYPE_F16_F32RES, F16RES_FWD_REL)):
        u = pkg.UNet(ctx, cfg, dt, seed=0)
        outs[name] = u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
        rep[name] = errs(outs[name], ref)
        del u
        print(f"UNet::forward 1024^2 {name} vs oracle: rel {rep[name]['rel']:.3e} rms-rel {rep[name]['rms_rel']:.3e} max-abs {rep[name]['max_abs']:.3e}")
        assert rep[name]["rel"] < tol, (name, rep[name])
    rep["f16_vs_f32_engine"] = errs(outs["f16"], outs["f32"])
    # round 5: the f16 bounds are anchored in the ORACLE, not in what the engine last measured.  (a) The reference's own GPU arithmetic is
    # LibTorch<f16> (src/bin/sample/main.rs:122: every op output an f16 tensor); the oracle in that arithmetic (oracle NUM "f16ref",
    # fixture fullsize_unet1024_f16ref.npz) is 6.7e-3 from the fp32 oracle on this forward -- the engine's f16 mode must stay well inside that
    # class (<= half).  (b) The oracle with only the GEMM / attention OPERANDS rounded to f16 (NUM "operands", fixture
    # fullsize_unet1024_classes.npz: 9.7e-4 with every class rounded) models the engine's arithmetic: f16_f32res <= 1.5x that, f16 (which also
    # rounds the residual stream and folds the LayerNorms) <= 2x.
    g16 = np.load(os.path.join(GOLD, "fullsize_unet1024_f16ref.npz"))
    env = errs(torch.from_numpy(g16["out"]), ref)
    gc = np.load(os.path.join(GOLD, "fullsize_unet1024_classes.npz"))
    model_all = float(gc["err_rel"][list(gc["names"]).index("qkv+attn+out+xattn+geglu+ff+conv")])
    rep["reference_f16_class"] = env
    rep["oracle_f16_operand_model_rel"] = model_all
    print(f"UNet::forward 1024^2: reference f16 class (oracle f16ref vs fp32 oracle) rel {env['rel']:.3e}; oracle f16-operand model rel {model_all:.3e}; "
          f"engine f16 {rep['f16']['rel']:.3e}, f16_f32res {rep['f16_f32res']['rel']:.3e}")
    assert abs(env["rel"] - float(g16["err_rel"][0])) < 1e-6
    assert rep["f16"]["rel"] <= 0.5 * env["rel"], "the engine's f16 mode left the reference's own f16 numerical class"
    assert rep["f16_f32res"]["rel"] <= 1.5 * model_all and rep["f16"]["rel"] <= 2.0 * model_all, (rep["f16"], rep["f16_f32res"], model_all)
    REPORT["unet_forward_1024_vs_oracle"] = rep


def test_hl_demote_instrument(pkg, ctx):
    """sdxl_debug_set("hl_demote", mask) -- the precision-frontier instrument (tools/precision_frontier.py): mask 0 is the split engine bit for bit,
    and with every class demoted the split engine reproduces the f16-operand arithmetic: its error against the fp32 oracle is the oracle's own
    f16-operand model (fixture fullsize_unet1024_classes.npz) and the F16_F32RES engine's, within 25 %."""
    g = np.load(os.path.join(GOLD, "fullsize_unet1024.npz"))
    gc = np.load(os.path.join(GOLD, "fullsize_unet1024_classes.npz"))
    names = list(gc["names"])
    cfg = pkg.sdxl_base_config()
    x, t = seeded(1, 4, 128, 128, seed=111), torch.tensor([500], dtype=torch.int32)
    c, y = seeded(1, 77, cfg.context_dim, seed=112), seeded(1, cfg.adm_in_channels, seed=113)
    ref = torch.from_numpy(g["out"])
    u = pkg.UNet(ctx, cfg, pkg.DTYPE_F32_SPLIT, seed=0)
    rep = {}
    try:
        base = u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
        pkg.debug_set("hl_demote", 0x7FF)
        alld = u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
        pkg.debug_set("hl_demote", 1 << 4)            # GEGLU alone
        geglu = u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
        pkg.debug_set("hl_demote", 0)
        again = u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
    finally:
        pkg.debug_set("hl_demote", 0)
    assert torch.equal(base, again), "mask 0 after a demoted run is not the split engine any more"
    rep["none"], rep["all"], rep["geglu"] = errs(base, ref), errs(alld, ref), errs(geglu, ref)
    m_all = float(gc["err_rel"][names.index("qkv+attn+out+xattn+geglu+ff+conv")]); m_geglu = float(gc["err_rel"][names.index("geglu")])
    print(f"hl_demote: none {rep['none']['rel']:.3e}, all {rep['all']['rel']:.3e} (oracle f16-operand model {m_all:.3e}), geglu alone {rep['geglu']['rel']:.3e} (model {m_geglu:.3e})")
    assert rep["none"]["rel"] < F32_FWD_REL
    assert 0.75 * m_all < rep["all"]["rel"] < 1.25 * m_all, (rep["all"], m_all)
    assert 0.6 * m_geglu < rep["geglu"]["rel"] < 1.6 * m_geglu, (rep["geglu"], m_geglu)
    REPORT["hl_demote_instrument"] = rep


def test_decode_1024_matches_oracle(pkg, ctx):
    """LatentDecoder::latent_to_image at 1024x1024 (Decoder::forward, autoencoder/mod.rs:203-216)"""
    g = np.load(os.path.join(GOLD, "fullsize_decode1024.npz"))
    latent = seeded(1, 4, 128, 128, seed=121)
    assert np.allclose(checksum(latent), g["in_checksum"], rtol=1e-9)
    rep = {}
    for name, dt in (("f32", pkg.DTYPE_F32), ("f32_split", pkg.DTYPE_F32_SPLIT), ("f16", pkg.DTYPE_F16)):
        ld = pkg.LatentDecoder(ctx, None, dt, seed=0)
        ld.decode_latent(latent.cuda())                                   # warm-up (plans, arena)
        torch.cuda.synchronize()
        t0 = time.time()
        img = ld.decode_latent(latent.cuda())
        torch.cuda.synchronize()
        ms = (time.time() - t0) * 1e3
        img = img.cpu()
        u8 = ld.latent_to_image(latent.cuda()).buffer.cpu().numpy()
        e_sub = errs(img[:, :, ::SUB, ::SUB], torch.from_numpy(g["image_sub"]))
        crops = torch.stack([img[0, :, r:r + 64, c:c + 64] for r, c in CROPS])
        e_crop = errs(crops, torch.from_numpy(g["crops"]))
        d8 = np.abs(u8[:, ::SUB, ::SUB].astype(np.int32) - g["u8_sub"].astype(np.int32))
        d8c = np.abs(np.stack([u8[0, r:r + 64, c:c + 64] for r, c in CROPS]).astype(np.int32) - g["crops_u8"].astype(np.int32))
        hist = np.bincount(u8.reshape(-1), minlength=256)
        rep[name] = dict(image_sub=e_sub, crops=e_crop, u8_max_diff=int(max(d8.max(), d8c.max())),
                         u8_frac_diff=float(((d8 > 0).sum() + (d8c > 0).sum()) / (d8.size + d8c.size)),
                         u8_hist_l1=float(np.abs(hist - g["u8_hist"]).sum() / hist.sum()), decode_ms=ms)
        print(f"decode 1024^2 {name} vs oracle: image max-abs {e_sub['max_abs']:.3e} / crops {e_crop['max_abs']:.3e} (|ref| {e_sub['ref_max']:.2f}); "
              f"u8 max diff {rep[name]['u8_max_diff']} ({rep[name]['u8_frac_diff']:.2e} of bytes); decode {ms:.1f} ms")
        del ld
    REPORT["decode_1024_vs_oracle"] = rep
    assert rep["f32"]["image_sub"]["max_abs"] <= IMG_ABS_F32 * max(1.0, rep["f32"]["image_sub"]["ref_max"])
    assert rep["f32"]["crops"]["max_abs"] <= IMG_ABS_F32 * max(1.0, rep["f32"]["crops"]["ref_max"])
    assert rep["f32"]["u8_max_diff"] <= 1
    # the split-operand mode (3 f16 MFMAs per product on (hi, lo) operand pairs) must sit in the exact-fp32 class
    assert rep["f32_split"]["image_sub"]["max_abs"] <= 1e-2 * IMG_ABS_F32 * max(1.0, rep["f32_split"]["image_sub"]["ref_max"]), rep["f32_split"]
    assert rep["f32_split"]["crops"]["max_abs"] <= 1e-2 * IMG_ABS_F32 * max(1.0, rep["f32_split"]["crops"]["ref_max"])
    assert rep["f32_split"]["u8_max_diff"] <= 1 and rep["f32_split"]["u8_frac_diff"] <= 2e-4
    assert rep["f16"]["image_sub"]["rel"] < F16_DECODE_REL


def test_decode_1024_f16_representable_weights(pkg, ctx):
    """latent_to_image at 1024x1024 with the VAE decoder's parameters rounded to f16 on both sides (the reference's decoder record is HalfPrecisionSettings too,
    src/bin/sample/main.rs:37-51; fixture oracle/make_golden_r6.py decode1024_f16w; bench.py --weights f16): the split-operand VAE then leaves out the w_lo
    MFMAs (two per product instead of three) -- same exact-fp32 class, faster decode."""
    gp = os.path.join(GOLD, "fullsize_decode1024_f16w.npz")
    if not os.path.exists(gp):
        pytest.skip("tests/golden/fullsize_decode1024_f16w.npz not generated (python -m oracle.make_golden_r6 decode1024_f16w, ~2 min)")
    g = np.load(gp)
    latent = seeded(1, 4, 128, 128, seed=121)
    assert np.allclose(checksum(latent), g["in_checksum"], rtol=1e-9)
    rep = {}
    for name, seed in (("f32_split", pkg.SEED_F16_WEIGHTS), ("f32_split_fp32_weights", 0)):
        ld = pkg.LatentDecoder(ctx, None, pkg.DTYPE_F32_SPLIT, seed=seed)
        ld.decode_latent(latent.cuda())
        torch.cuda.synchronize()
        t0 = time.time()
        img = ld.decode_latent(latent.cuda())
        torch.cuda.synchronize()
        ms = (time.time() - t0) * 1e3
        u8 = ld.latent_to_image(latent.cuda()).buffer.cpu().numpy()
        rep[name] = dict(decode_ms=ms)
        if seed:
            e = errs(img.cpu()[:, :, ::SUB, ::SUB], torch.from_numpy(g["image_sub"]))
            d8 = np.abs(u8[:, ::SUB, ::SUB].astype(np.int32) - g["u8_sub"].astype(np.int32))
            rep[name].update(image_sub=e, u8_max_diff=int(d8.max()), u8_frac_diff=float((d8 > 0).mean()))
        del ld
    print(f"decode 1024^2, f16-representable VAE weights, f32_split vs oracle: image max-abs {rep['f32_split']['image_sub']['max_abs']:.3e}; u8 max diff "
          f"{rep['f32_split']['u8_max_diff']} ({rep['f32_split']['u8_frac_diff']:.2e} of bytes); decode {rep['f32_split']['decode_ms']:.1f} ms "
          f"(fp32 weights, three MFMAs per product: {rep['f32_split_fp32_weights']['decode_ms']:.1f} ms)")
    REPORT["decode_1024_f16_weights_vs_oracle"] = rep
    e = rep["f32_split"]["image_sub"]
    assert e["max_abs"] <= 1e-2 * IMG_ABS_F32 * max(1.0, e["ref_max"]), e
    assert rep["f32_split"]["u8_max_diff"] <= 1 and rep["f32_split"]["u8_frac_diff"] <= 2e-4


def test_config2_trajectory_parity_and_drift(pkg, ctx):
    """the benchmarked configuration (BASELINE configs[1]): 1024x1024, n_steps=30 -> 31 CFG pairs, CFG 7.5.
    F32 engine against the ORACLE's trajectory, then the speed modes against the F32 engine step by step."""
    cfg = pkg.sdxl_base_config()
    i = _inputs(cfg, 130, 128)
    n_it = pkg.step_count(30)
    assert n_it == 31
    trajs, secs = {}, {}
    for name, dt in (("f32", pkg.DTYPE_F32), ("f32_split", pkg.DTYPE_F32_SPLIT), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX), ("f16", pkg.DTYPE_F16),
                     ("f16_f32res", pkg.DTYPE_F16_F32RES)):
        d = pkg.Diffuser(ctx, cfg, dt, seed=0)
        trace = torch.zeros(n_it, 1, 4, 128, 128, device="cuda")
        d.set_trace(trace)
        t0 = time.time()
        lat = d.sample_latent(_cond(pkg, i, (1024, 1024)), 7.5, 30, i["noise"].cuda())
        torch.cuda.synchronize()
        secs[name] = time.time() - t0
        d.set_trace(None)
        assert torch.isfinite(trace).all() and torch.equal(trace[-1], lat)
        trajs[name] = trace.cpu()
        del d
    rep = dict(engine_seconds=secs, ref_absmax=[float(trajs["f32"][k].abs().max()) for k in range(n_it)])
    gp = os.path.join(GOLD, "fullsize_config2.npz")
    if os.path.exists(gp):
        g = np.load(gp)
        assert np.allclose(checksum(*i.values()), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
        steps = [int(s) for s in g["steps"]]
        ref_traj = torch.from_numpy(g["traj"])
        rep["f32_vs_oracle"] = {str(s): errs(trajs["f32"][s], ref_traj[j]) for j, s in enumerate(steps)}
        rep["f32_vs_oracle"]["final"] = errs(trajs["f32"][-1], torch.from_numpy(g["latent"]))
        rep["oracle_seconds"] = float(g["oracle_seconds"][0])
        fin = rep["f32_vs_oracle"]["final"]
        print(f"config 2 (31 steps, CFG 7.5) F32 engine vs oracle: final max-abs {fin['max_abs']:.3e} rel {fin['rel']:.3e} (|ref| {fin['ref_max']:.1f}); "
              f"engine {secs['f32']:.1f} s vs oracle {rep['oracle_seconds']:.0f} s")
        for j, s in enumerate(steps):
            assert rep["f32_vs_oracle"][str(s)]["max_abs"] <= lat_bound(ref_traj[j]), (s, rep["f32_vs_oracle"][str(s)])
        # the split-operand mode (fp32 stream, (hi, lo) f16 GEMM operands, fp32 attention) against the SAME oracle trajectory and bar
        rep["f32_split_vs_oracle"] = {str(s): errs(trajs["f32_split"][s], ref_traj[j]) for j, s in enumerate(steps)}
        rep["f32_split_vs_oracle"]["final"] = errs(trajs["f32_split"][-1], torch.from_numpy(g["latent"]))
        fs = rep["f32_split_vs_oracle"]["final"]
        print(f"config 2 (31 steps, CFG 7.5) F32_SPLIT engine vs oracle: final max-abs {fs['max_abs']:.3e} rel {fs['rel']:.3e}; engine {secs['f32_split']:.1f} s")
        for j, s in enumerate(steps):
            assert rep["f32_split_vs_oracle"][str(s)]["max_abs"] <= lat_bound(ref_traj[j]), (s, rep["f32_split_vs_oracle"][str(s)])
        # round 5: the mixed mode (split engine, self-attention + GEGLU projection on f16 operands: the classes the measured precision frontier
        # affords, profiles/r05_precision_frontier.json) is held to the SAME bar on every recorded step -- and must be the faster engine
        rep["f32_split_mix_vs_oracle"] = {str(s): errs(trajs["f32_split_mix"][s], ref_traj[j]) for j, s in enumerate(steps)}
        rep["f32_split_mix_vs_oracle"]["final"] = errs(trajs["f32_split_mix"][-1], torch.from_numpy(g["latent"]))
        fm = rep["f32_split_mix_vs_oracle"]["final"]
        print(f"config 2 (31 steps, CFG 7.5) F32_SPLIT_MIX engine vs oracle: final max-abs {fm['max_abs']:.3e} rel {fm['rel']:.3e} (bound {lat_bound(torch.from_numpy(g['latent'])):.3e}); "
              f"engine {secs['f32_split_mix']:.1f} s (F32_SPLIT {secs['f32_split']:.1f} s)")
        for j, s in enumerate(steps):
            assert rep["f32_split_mix_vs_oracle"][str(s)]["max_abs"] <= lat_bound(ref_traj[j]), (s, rep["f32_split_mix_vs_oracle"][str(s)])
        assert fm["max_abs"] <= lat_bound(torch.from_numpy(g["latent"]))
        # the BENCHMARKED precision (and its fp32-stream twin) against the ORACLE's own trajectory, directly -- not only through
        # the F32 engine below: per recorded step and on the final latent, held to the same <= 2x-measured relative bars
        for name in ("f16", "f16_f32res"):
            vo = {str(s): errs(trajs[name][s], ref_traj[j]) for j, s in enumerate(steps)}
            vo["final"] = errs(trajs[name][-1], torch.from_numpy(g["latent"]))
            rep[name + "_vs_oracle"] = vo
            print(f"config 2 (31 steps, CFG 7.5) {name} engine vs oracle: final max-abs {vo['final']['max_abs']:.3e} rel {vo['final']['rel']:.3e}")
            assert vo["final"]["rel"] < F16_TRAJ_REL[name], (name, vo["final"])
            for s_ in steps:
                assert vo[str(s_)]["rel"] < F16_TRAJ_REL[name], (name, s_, vo[str(s_)])
        # round 5: the same trajectory in the REFERENCE's own GPU arithmetic (oracle NUM "f16ref": LibTorch<f16>, every op output an f16 tensor;
        # fixture fullsize_config2_f16ref.npz, 2721 s of oracle time): 1.03 absolute from the fp32 oracle at the end.  The engine's f16 modes must
        # stay inside HALF of that class at every recorded step -- a bound that does not move with the engine.
        g16p = os.path.join(GOLD, "fullsize_config2_f16ref.npz")
        if os.path.exists(g16p):
            g16 = np.load(g16p)
            env_steps = {int(s_): errs(torch.from_numpy(g16["traj"][j]), ref_traj[j]) for j, s_ in enumerate(g16["steps"])}
            env_final = errs(torch.from_numpy(g16["latent"]), torch.from_numpy(g["latent"]))
            rep["reference_f16_class"] = {"final": env_final, **{str(k): v for k, v in env_steps.items()}}
            print(f"config 2: reference f16 class (oracle f16ref vs fp32 oracle) final max-abs {env_final['max_abs']:.3e} rel {env_final['rel']:.3e}; "
                  f"engine f16 is {env_final['max_abs'] / rep['f16_vs_oracle']['final']['max_abs']:.1f}x inside it")
            for name in ("f16", "f16_f32res"):
                assert rep[name + "_vs_oracle"]["final"]["max_abs"] <= 0.5 * env_final["max_abs"], (name, rep[name + "_vs_oracle"]["final"], env_final)
                for s_ in steps:
                    if s_ in env_steps:
                        assert rep[name + "_vs_oracle"][str(s_)]["max_abs"] <= 0.5 * env_steps[s_]["max_abs"], (name, s_)
    for name in ("f16", "f16_f32res"):
        per = [errs(trajs[name][k], trajs["f32"][k]) for k in range(n_it)]
        rep[name + "_vs_f32"] = per
        print(f"config 2 drift {name} vs F32 engine: step 0 rel {per[0]['rel']:.2e}, step 15 {per[15]['rel']:.2e}, final {per[-1]['rel']:.2e} "
              f"(max-abs {per[-1]['max_abs']:.3e}, rms-rel {per[-1]['rms_rel']:.2e}); {secs[name]:.2f} s")
        try:
            os.makedirs(OUT_DIR, exist_ok=True)
            with open(os.path.join(OUT_DIR, f"r05_drift_{name}.json"), "w") as fh:
                json.dump(dict(config="SDXL-base 1024x1024, n_steps=30 (31 iterations), CFG 7.5, synthetic weights seed 0",
                               reference="SDXL_DTYPE_F32 engine trajectory (oracle-anchored: f32_vs_oracle in r05_parity_baseline.json)",
                               mode=name, per_step=per, ref_absmax=rep["ref_absmax"]), fh, indent=1)
        except OSError:
            pass
        assert per[-1]["rel"] < F16_TRAJ_REL[name], (name, per[-1])        # <= 2x the measured drift
    REPORT["config2_trajectory"] = rep


def test_split_operand_tile_selection_does_not_change_results(pkg, ctx):
    """round 5: the split-operand GEMMs take the f16 engine's extra tiles (sdxl_debug_set "hl_tile96": 96x128 linears, 4-wave 128x160, in-launch
    split-K for the long 3x3 convolutions).  Every tile sums k in the same order, so the tile bits (1 | 4 | 8) must not change one output bit;
    split-K (bit 16) re-associates 13 long contractions and stays inside the strict forward bar."""
    g = np.load(os.path.join(GOLD, "fullsize_unet1024.npz"))
    cfg = pkg.sdxl_base_config()
    x, t = seeded(1, 4, 128, 128, seed=111), torch.tensor([500], dtype=torch.int32)
    c, y = seeded(1, 77, cfg.context_dim, seed=112), seeded(1, cfg.adm_in_channels, seed=113)
    ref = torch.from_numpy(g["out"])
    outs = {}
    try:
        for knob in (0, 13, 29):
            pkg.debug_set("hl_tile96", knob)
            u = pkg.UNet(ctx, cfg, pkg.DTYPE_F32_SPLIT, seed=0)
            outs[knob] = u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
            del u
    finally:
        pkg.debug_set("hl_tile96", 29)
    assert torch.equal(outs[0], outs[13]), "a tile shape changed the k-summation order of a split-operand GEMM"
    e0, e29 = errs(outs[0], ref), errs(outs[29], ref)
    print(f"F32_SPLIT forward 1024^2 vs oracle: two-tile selection {e0['rel']:.3e}, with in-launch split-K {e29['rel']:.3e}")
    assert e0["rel"] < F32_FWD_REL and e29["rel"] < F32_FWD_REL
    REPORT["split_operand_tile_selection"] = {"tiles_256x128_128x128_only": e0, "default": e29}


def test_config2_second_prompt(pkg, ctx):
    """BASELINE configs[1] on a SECOND prompt / noise (seeds 231..235; fixture oracle/make_golden_r5.py config2b): one trajectory is not a distribution.
    The fp32-class engines are held to the bound on every recorded step as on the first prompt -- and so is the mixed mode (measured 0.60 ... 0.77 of the
    bound per step, 0.75 at the end: 0.0179 of 0.0238; first prompt: 0.88); the f16 mode is recorded against its relative bar."""
    gp = os.path.join(GOLD, "fullsize_config2b.npz")
    if not os.path.exists(gp):
        pytest.skip("tests/golden/fullsize_config2b.npz not generated (python -m oracle.make_golden_r5 config2b, ~25 min)")
    g = np.load(gp)
    cfg = pkg.sdxl_base_config()
    i = _inputs(cfg, 230, 128)
    assert np.allclose(checksum(*i.values()), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    steps = [int(s_) for s_ in g["steps"]]
    ref_traj, ref = torch.from_numpy(g["traj"]), torch.from_numpy(g["latent"])
    rep = {}
    for name, dt in (("f32_split", pkg.DTYPE_F32_SPLIT), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX), ("f16", pkg.DTYPE_F16)):
        d = pkg.Diffuser(ctx, cfg, dt, seed=0)
        trace = torch.zeros(31, 1, 4, 128, 128, device="cuda")
        d.set_trace(trace)
        lat = d.sample_latent(_cond(pkg, i, (1024, 1024)), 7.5, 30, i["noise"].cuda())
        torch.cuda.synchronize()
        d.set_trace(None)
        tr = trace.cpu()
        rep[name] = {str(s_): errs(tr[s_], ref_traj[j]) for j, s_ in enumerate(steps)}
        rep[name]["final"] = errs(lat.cpu(), ref)
        del d
        print(f"config 2, second prompt, {name} vs oracle: final max-abs {rep[name]['final']['max_abs']:.3e} rel {rep[name]['final']['rel']:.3e} "
              f"(|ref| {rep[name]['final']['ref_max']:.1f}, bound {lat_bound(ref):.3e})")
    REPORT["config2_second_prompt"] = rep
    for j, s_ in enumerate(steps):
        assert rep["f32_split"][str(s_)]["max_abs"] <= lat_bound(ref_traj[j]), (s_, rep["f32_split"][str(s_)])
        assert rep["f32_split_mix"][str(s_)]["max_abs"] <= lat_bound(ref_traj[j]), (s_, rep["f32_split_mix"][str(s_)])
    assert rep["f16"]["final"]["rel"] < F16_TRAJ_REL["f16"], rep["f16"]["final"]


def test_config2_second_prompt_f16_weights(pkg, ctx):
    """The second prompt on f16-representable weights (fixture oracle/make_golden_r5.py config2b_f16w): SDXL_DTYPE_F32_SPLIT_MIX_F16W's claim -- the
    config-2 latent inside the scaled bound on the weights the reference's records hold -- on a second trajectory, every recorded step.  `knob127`
    (the seventh class) is recorded beside it and held to the bound at the last step only."""
    gp = os.path.join(GOLD, "fullsize_config2b_f16w.npz")
    if not os.path.exists(gp):
        pytest.skip("tests/golden/fullsize_config2b_f16w.npz not generated (python -m oracle.make_golden_r5 config2b_f16w, ~30 min)")
    g = np.load(gp)
    cfg = pkg.sdxl_base_config()
    i = _inputs(cfg, 230, 128)
    assert np.allclose(checksum(*i.values()), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    steps = [int(s_) for s_ in g["steps"]]
    ref_traj, ref = torch.from_numpy(g["traj"]), torch.from_numpy(g["latent"])
    rep = {}
    for name, dt in (("f32_split", pkg.DTYPE_F32_SPLIT), ("f32_split_f16w", pkg.DTYPE_F32_SPLIT_F16W), ("f32_split_mix_f16w", pkg.DTYPE_F32_SPLIT_MIX_F16W),
                     ("f32_split_mix_f16w_geglu2", pkg.DTYPE_F32_SPLIT_MIX_F16W_GEGLU2), ("knob127", pkg.DTYPE_F32_SPLIT_MIX)):
        if name == "knob127":
            pkg.debug_set("mix_classes", 127)
        try:
            d = pkg.Diffuser(ctx, cfg, dt, seed=pkg.SEED_F16_WEIGHTS)
        finally:
            pkg.debug_set("mix_classes", -1)
        trace = torch.zeros(31, 1, 4, 128, 128, device="cuda")
        d.set_trace(trace)
        lat = d.sample_latent(_cond(pkg, i, (1024, 1024)), 7.5, 30, i["noise"].cuda())
        torch.cuda.synchronize()
        d.set_trace(None)
        tr = trace.cpu()
        rep[name] = {str(s_): errs(tr[s_], ref_traj[j]) for j, s_ in enumerate(steps)}
        rep[name]["final"] = errs(lat.cpu(), ref)
        del d
        print(f"config 2, second prompt, f16-representable weights, {name} vs oracle: final max-abs {rep[name]['final']['max_abs']:.3e} "
              f"(|ref| {rep[name]['final']['ref_max']:.1f}, bound {lat_bound(ref):.3e}); of the bound per recorded step: "
              + " ".join(f"{rep[name][str(s_)]['max_abs'] / lat_bound(ref_traj[j]):.2f}" for j, s_ in enumerate(steps)))
    REPORT["config2_second_prompt_f16_weights"] = rep
    for j, s_ in enumerate(steps):
        assert rep["f32_split"][str(s_)]["max_abs"] <= lat_bound(ref_traj[j]), (s_, rep["f32_split"][str(s_)])
        assert rep["f32_split_mix_f16w"][str(s_)]["max_abs"] <= lat_bound(ref_traj[j]), (s_, rep["f32_split_mix_f16w"][str(s_)])
    assert rep["f32_split"]["final"]["max_abs"] < 1e-3
    assert rep["f32_split_f16w"]["final"]["max_abs"] < 1e-3      # F32_SPLIT's arithmetic on the f16 kernels: the UNSCALED 1e-3, like F32_SPLIT
    assert rep["f32_split_mix_f16w"]["final"]["max_abs"] <= lat_bound(ref)
    assert rep["knob127"]["final"]["max_abs"] <= lat_bound(ref), rep["knob127"]["final"]


def test_config2_trajectory_f16_representable_weights(pkg, ctx):
    """The benchmarked trajectory with the weights a real SDXL record holds (every parameter an f16 value: HalfPrecisionSettings,
    src/bin/sample/main.rs:37) against the oracle's own 31-step trajectory on the same weights.  The split-operand engine then leaves
    out the w_lo x a_hi MFMAs (the packed lo halves are zero) -- same bar as the strict mode; the f16 engine's error on these
    weights is activation rounding alone."""
    gp = os.path.join(GOLD, "fullsize_config2_f16w.npz")
    if not os.path.exists(gp):
        pytest.skip("tests/golden/fullsize_config2_f16w.npz not generated (python -m oracle.make_golden_r3 config2_f16w, ~25 min)")
    g = np.load(gp)
    cfg = pkg.sdxl_base_config()
    i = _inputs(cfg, 130, 128)
    assert np.allclose(checksum(*i.values()), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    steps = [int(s_) for s_ in g["steps"]]
    ref_traj, ref = torch.from_numpy(g["traj"]), torch.from_numpy(g["latent"])
    rep = {}
    # "knob127": the F16W mode's six classes + the cross-attention itself on the f16 engine's fused launch (sdxl_debug_set "mix_classes" bit 64) -- 7 % faster,
    # 92 % of the bound at the last step: measured and recorded, NOT part of the mode
    for name, dt in (("f32_split", pkg.DTYPE_F32_SPLIT), ("f32_split_f16w", pkg.DTYPE_F32_SPLIT_F16W), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX), ("f32_split_mix_f16w", pkg.DTYPE_F32_SPLIT_MIX_F16W),
                     ("f32_split_mix_f16w_geglu2", pkg.DTYPE_F32_SPLIT_MIX_F16W_GEGLU2), ("knob127", pkg.DTYPE_F32_SPLIT_MIX), ("f16", pkg.DTYPE_F16)):
        if name == "knob127":
            pkg.debug_set("mix_classes", 127)
        try:
            d = pkg.Diffuser(ctx, cfg, dt, seed=pkg.SEED_F16_WEIGHTS)
        finally:
            pkg.debug_set("mix_classes", -1)
        trace = torch.zeros(31, 1, 4, 128, 128, device="cuda")
        d.set_trace(trace)
        t0 = time.time()
        lat = d.sample_latent(_cond(pkg, i, (1024, 1024)), 7.5, 30, i["noise"].cuda())
        torch.cuda.synchronize()
        dt_s = time.time() - t0
        d.set_trace(None)
        tr = trace.cpu()
        rep[name] = {str(s_): errs(tr[s_], ref_traj[j]) for j, s_ in enumerate(steps)}
        rep[name]["final"] = errs(lat.cpu(), ref)
        rep[name]["engine_seconds"] = dt_s
        del d
        print(f"config 2, f16-representable weights, {name} vs oracle: final max-abs {rep[name]['final']['max_abs']:.3e} rel {rep[name]['final']['rel']:.3e} "
              f"(|ref| {rep[name]['final']['ref_max']:.1f}); engine {dt_s:.2f} s")
    REPORT["config2_f16_weights_trajectory"] = rep
    for j, s_ in enumerate(steps):
        assert rep["f32_split"][str(s_)]["max_abs"] <= lat_bound(ref_traj[j]), (s_, rep["f32_split"][str(s_)])
        # SDXL_DTYPE_F32_SPLIT_F16W: F32_SPLIT's arithmetic on the f16 kernels -- the UNSCALED 1e-3 at every recorded step, like F32_SPLIT
        assert rep["f32_split_f16w"][str(s_)]["max_abs"] <= LAT_ABS, (s_, rep["f32_split_f16w"][str(s_)])
        assert rep["f32_split_mix"][str(s_)]["max_abs"] <= lat_bound(ref_traj[j]), (s_, rep["f32_split_mix"][str(s_)])
        # SDXL_DTYPE_F32_SPLIT_MIX_F16W (QKV projection and FF-out on f16 as well: the mode FOR these weights): same bar on every recorded step
        assert rep["f32_split_mix_f16w"][str(s_)]["max_abs"] <= lat_bound(ref_traj[j]), (s_, rep["f32_split_mix_f16w"][str(s_)])
        assert rep["f32_split_mix_f16w_geglu2"][str(s_)]["max_abs"] <= lat_bound(ref_traj[j]), (s_, rep["f32_split_mix_f16w_geglu2"][str(s_)])
    assert rep["f32_split_mix"]["final"]["max_abs"] <= lat_bound(ref)
    assert rep["f32_split_mix_f16w"]["final"]["max_abs"] <= lat_bound(ref)
    assert rep["f32_split_mix_f16w_geglu2"]["final"]["max_abs"] <= lat_bound(ref), rep["f32_split_mix_f16w_geglu2"]["final"]      # measured 0.48 / 0.72 of the bound in two builds (max-abs jitter)
    # (the seventh-class knob -- the f16 engine's fused cross-attention launch -- is inside the bound on this prompt too: 0.0195 in round 5, 0.0167 in round 6 after
    #  bit-level changes elsewhere; the max over 65 536 latent values of 31 steps of accumulated rounding moves by +-15 % under such perturbations, which is why
    #  a mode is given margin and the knob stays a knob)
    assert rep["knob127"]["final"]["max_abs"] <= lat_bound(ref), rep["knob127"]["final"]
    assert rep["f16"]["final"]["rel"] < F16_TRAJ_REL["f16"], rep["f16"]["final"]


# ------------------------------------------------------------------------------------------ round 3: configs[3] / configs[4]
def _refiner_cond(pkg, i, res=(1024, 1024)):
    return pkg.Conditioning(context_open_clip=i["ctx"].cuda(), channel_context_refiner=i["y"].cuda(),
                            unconditional_context_open_clip=i["uctx"].cuda(), unconditional_channel_context_refiner=i["uy"].cuda(),
                            resolution=res)


def test_refiner_forward_1024_matches_oracle(pkg, ctx):
    """BASELINE configs[3]: one forward of the 4-level refiner UNet (384/768/1536/1536 channels, depth 4, context 1280) at
    1024x1024 -- shapes (N = 1536, K = 13824, 24 LayerNorm slots) no base-model test reaches"""
    g = np.load(os.path.join(GOLD, "fullsize_refiner1024.npz"))
    cfg = pkg.sdxl_refiner_config()
    x, t = seeded(1, 4, 128, 128, seed=141), torch.tensor([150], dtype=torch.int32)
    c, y = seeded(1, 77, cfg.context_dim, seed=142), seeded(1, cfg.adm_in_channels, seed=143)
    assert np.allclose(checksum(x, c, y), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    ref = torch.from_numpy(g["out"])
    rep = {}
    for name, dt, tol in (("f32", pkg.DTYPE_F32, F32_FWD_REL), ("f32_split", pkg.DTYPE_F32_SPLIT, F32_FWD_REL), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX, 3.0e-4), ("f16", pkg.DTYPE_F16, F16_FWD_REL),
                          ("f16_f32res", pkg.DTYPE_F16_F32RES, F16RES_FWD_REL)):
        u = pkg.UNet(ctx, cfg, dt, seed=0)
        outs = [u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu() for _ in range(3)]     # eager, capture, replay
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), "hipGraph replay differs from the eager run"
        rep[name] = errs(outs[0], ref)
        del u
        print(f"refiner UNet::forward 1024^2 {name} vs oracle: rel {rep[name]['rel']:.3e} rms-rel {rep[name]['rms_rel']:.3e} max-abs {rep[name]['max_abs']:.3e}")
        assert rep[name]["rel"] < tol, (name, rep[name])       # measured 2.9e-6 / 1.33e-3 / 9.0e-4: the base model's bounds hold
    REPORT["refiner_forward_1024_vs_oracle"] = rep


def test_refine_latent_1024_matches_oracle(pkg, ctx):
    """Diffuser::refine_latent at full size (stablediffusion/mod.rs:355-376): step_start 800, n_steps 10 -> t = 199, 99"""
    g = np.load(os.path.join(GOLD, "fullsize_refine1024.npz"))
    cfg = pkg.sdxl_refiner_config()
    i = dict(latent=seeded(1, 4, 128, 128, seed=151), noise=seeded(1, 4, 128, 128, seed=152), ctx=seeded(1, 77, cfg.context_dim, seed=153),
             uctx=seeded(77, cfg.context_dim, seed=154), y=seeded(1, cfg.adm_in_channels, seed=155), uy=seeded(cfg.adm_in_channels, seed=156))
    assert np.allclose(checksum(*i.values()), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    assert pkg.step_count(10, 800) == 2
    ref_traj, ref = torch.from_numpy(g["traj"]), torch.from_numpy(g["latent"])
    rep = {}
    for name, dt in (("f32", pkg.DTYPE_F32), ("f32_split", pkg.DTYPE_F32_SPLIT), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX), ("f16", pkg.DTYPE_F16)):
        d = pkg.Diffuser(ctx, cfg, dt, seed=0)
        trace = torch.zeros(2, 1, 4, 128, 128, device="cuda")
        d.set_trace(trace)
        out = d.refine_latent(i["latent"].cuda(), _refiner_cond(pkg, i), 7.5, 800, 10, i["noise"].cuda())
        torch.cuda.synchronize()
        d.set_trace(None)
        assert torch.equal(trace[-1], out)
        rep[name] = dict(per_step=[errs(trace[k], ref_traj[k]) for k in range(2)], final=errs(out, ref))
        del d
        print(f"refine_latent 1024^2 {name} vs oracle: per step {['%.2e' % s['max_abs'] for s in rep[name]['per_step']]} "
              f"(rel {rep[name]['final']['rel']:.2e}, |ref| {rep[name]['final']['ref_max']:.2f})")
    REPORT["refine_latent_1024_vs_oracle"] = rep
    for k in range(2):
        for nm in ("f32", "f32_split", "f32_split_mix"):      # (the mixed mode is held to the strict modes' bar at every configuration)
            assert rep[nm]["per_step"][k]["max_abs"] <= lat_bound(ref_traj[k]), (nm, k, rep[nm]["per_step"][k])
    assert rep["f16"]["final"]["rel"] < 5.5e-4, rep["f16"]["final"]          # measured 2.6e-4 (1.4e-3 abs on |latent| 5.3)


def _encode_image():
    g = torch.Generator().manual_seed(161)
    yy, xx = torch.meshgrid(torch.arange(1024, dtype=torch.float32), torch.arange(1024, dtype=torch.float32), indexing="ij")
    base = torch.stack([torch.sin(xx / 37.0) * torch.cos(yy / 53.0), torch.sin((xx + yy) / 91.0), torch.cos(xx / 17.0 - yy / 29.0)], -1)
    img = (base * 0.35 + 0.5 + 0.08 * torch.randn(1024, 1024, 3, generator=g)).clamp(0, 1) * 255.0
    return img.to(torch.uint8)[None]


def test_encode_1024_matches_oracle(pkg, ctx):
    """BASELINE configs[4], first leg: LatentDecoder::image_to_latent of a 1024x1024 u8 image (stablediffusion/mod.rs:239-261,
    Encoder::forward autoencoder/mod.rs:131-144 with the PaddedConv2d downsamples :384-407)"""
    g = np.load(os.path.join(GOLD, "fullsize_encode1024.npz"))
    img = _encode_image()
    chk = np.array([float(img.numpy().astype(np.float64).sum()), float((img.numpy().astype(np.float64) ** 2).sum())])
    assert np.allclose(chk, g["in_checksum"], rtol=1e-12), "u8 test image differs from the fixture's: regenerate"
    ref = torch.from_numpy(g["latent"])
    rep = {}
    for name, dt in (("f32", pkg.DTYPE_F32), ("f32_split", pkg.DTYPE_F32_SPLIT), ("f16", pkg.DTYPE_F16)):
        ld = pkg.LatentDecoder(ctx, None, dt, seed=0, with_encoder=True)
        out = ld.image_to_latent(pkg.RawImages(img.cuda(), 1024, 1024)).cpu()
        rep[name] = errs(out, ref)
        del ld
        print(f"image_to_latent 1024^2 {name} vs oracle: max-abs {rep[name]['max_abs']:.3e} rel {rep[name]['rel']:.3e} (|ref| {rep[name]['ref_max']:.3f})")
    REPORT["encode_1024_vs_oracle"] = rep
    assert rep["f32"]["max_abs"] <= LAT_ABS and rep["f32"]["rel"] < 1e-4, rep["f32"]
    assert rep["f32_split"]["rel"] < 2e-5, rep["f32_split"]
    assert rep["f16"]["rel"] < 3.9e-3, rep["f16"]                            # measured 1.9e-3


def test_inpainting_1024_matches_oracle(pkg, ctx):
    """BASELINE configs[4]: Diffuser::sample_latent_with_inpainting at 1024x1024, 4 CFG-7.5 steps, mask = latent rows 0..24
    generated (the 200 px crop), reference latent = the oracle's encode, explicit per-step re-noise (stablediffusion/mod.rs:434-483)"""
    g = np.load(os.path.join(GOLD, "fullsize_inpaint1024.npz"))
    cfg = pkg.sdxl_base_config()
    i = dict(noise=seeded(1, 4, 128, 128, seed=171), ctx=seeded(1, 77, cfg.context_dim, seed=172), uctx=seeded(77, cfg.context_dim, seed=173),
             y=seeded(1, cfg.adm_in_channels, seed=174), uy=seeded(cfg.adm_in_channels, seed=175), step_noise=seeded(4, 1, 4, 128, 128, seed=176))
    assert np.allclose(checksum(*i.values()), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    reference = torch.from_numpy(g["reference"])
    mask = torch.zeros(1, 4, 128, 128, dtype=torch.bool)
    mask[:, :, 0:25, :] = True
    ref_traj, ref = torch.from_numpy(g["traj"]).clone(), torch.from_numpy(g["latent"])
    # The engine's per-step trace is taken AFTER its fused DDIM kernel, which already holds the blend for the NEXT iteration
    # (mask ? latent : reference * sqrt(a_next) + step_noise[next] * sqrt(1 - a_next), stablediffusion/mod.rs:463-465 at the top of
    # the next loop pass); the oracle's trace is the latent at the bottom of the pass.  Apply the same blend to the oracle's
    # per-step latents (host f64 scalars as the reference computes them) so whole tensors are compared; the last one has no next.
    alphas = pkg.default_alphas_cumprod()
    ts = [999, 749, 499, 249]
    for k in range(3):
        a_n = float(alphas[ts[k + 1]])
        ref_traj[k] = torch.where(mask, ref_traj[k], reference * (a_n ** 0.5) + i["step_noise"][k + 1] * ((1.0 - a_n) ** 0.5))
    rep = {}
    for name, dt in (("f32", pkg.DTYPE_F32), ("f32_split", pkg.DTYPE_F32_SPLIT), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX), ("f16", pkg.DTYPE_F16)):
        d = pkg.Diffuser(ctx, cfg, dt, seed=0)
        trace = torch.zeros(4, 1, 4, 128, 128, device="cuda")
        d.set_trace(trace)
        out = d.sample_latent_with_inpainting(_cond(pkg, i, (1024, 1024)), 7.5, 4, reference.cuda(), mask.cuda(), i["noise"].cuda(),
                                              i["step_noise"].cuda())
        torch.cuda.synchronize()
        d.set_trace(None)
        assert torch.equal(trace[-1], out)
        rep[name] = dict(per_step=[errs(trace[k], ref_traj[k]) for k in range(4)], final=errs(out, ref))
        del d
        print(f"inpainting 1024^2 {name} vs oracle: per step {['%.2e' % s['max_abs'] for s in rep[name]['per_step']]} "
              f"(final rel {rep[name]['final']['rel']:.2e}, |ref| {rep[name]['final']['ref_max']:.1f})")
    REPORT["inpainting_1024_vs_oracle"] = rep
    for k in range(4):
        for nm in ("f32", "f32_split"):
            assert rep[nm]["per_step"][k]["max_abs"] <= lat_bound(ref_traj[k]), (nm, k, rep[nm]["per_step"][k])
        # The mixed mode (f16 self-attention + f16 GEGLU operands) is 2.2e-4 of max|latent| on config 2 -- inside the scaled bound (2.5e-4) -- but
        # these four 250-step jumps amplify a forward's error more: measured 3.6e-4 / 3.1e-4 of max|latent| = 1.43x / 1.26x the bound at the first /
        # last step.  Recorded, and held to 2x the bound: the mode is a precision point between F32_SPLIT and F16, compliant at the benchmarked
        # configuration only.
        assert rep["f32_split_mix"]["per_step"][k]["max_abs"] <= 2.0 * lat_bound(ref_traj[k]), (k, rep["f32_split_mix"]["per_step"][k])
    assert rep["f16"]["final"]["rel"] < F16_TRAJ_REL["f16"], rep["f16"]["final"]


def test_inpainting_1024_f16_representable_weights(pkg, ctx):
    """The inpainting fixture on f16-representable UNet weights (oracle/make_golden_r5.py inpaint1024_f16w): where the mixed modes stand on the configuration
    whose four 250-step jumps amplify a forward's error most.  F32_SPLIT is held to the bound at every step; SDXL_DTYPE_F32_SPLIT_MIX / _F16W are recorded
    and held to 2x the bound like the mixed mode on fp32 weights (test_inpainting_1024_matches_oracle): precision points, compliant at the benchmarked
    configuration."""
    gp = os.path.join(GOLD, "fullsize_inpaint1024_f16w.npz")
    if not os.path.exists(gp):
        pytest.skip("tests/golden/fullsize_inpaint1024_f16w.npz not generated (python -m oracle.make_golden_r5 inpaint1024_f16w, ~10 min)")
    g = np.load(gp)
    cfg = pkg.sdxl_base_config()
    i = dict(noise=seeded(1, 4, 128, 128, seed=171), ctx=seeded(1, 77, cfg.context_dim, seed=172), uctx=seeded(77, cfg.context_dim, seed=173),
             y=seeded(1, cfg.adm_in_channels, seed=174), uy=seeded(cfg.adm_in_channels, seed=175), step_noise=seeded(4, 1, 4, 128, 128, seed=176))
    assert np.allclose(checksum(*i.values()), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    reference = torch.from_numpy(np.load(os.path.join(GOLD, "fullsize_inpaint1024.npz"))["reference"])
    mask = torch.zeros(1, 4, 128, 128, dtype=torch.bool)
    mask[:, :, 0:25, :] = True
    ref_traj, ref = torch.from_numpy(g["traj"]).clone(), torch.from_numpy(g["latent"])
    alphas = pkg.default_alphas_cumprod()
    ts = [999, 749, 499, 249]
    for k in range(3):       # (the engine's trace already holds the blend for the next iteration: see test_inpainting_1024_matches_oracle)
        a_n = float(alphas[ts[k + 1]])
        ref_traj[k] = torch.where(mask, ref_traj[k], reference * (a_n ** 0.5) + i["step_noise"][k + 1] * ((1.0 - a_n) ** 0.5))
    rep = {}
    for name, dt in (("f32_split", pkg.DTYPE_F32_SPLIT), ("f32_split_f16w", pkg.DTYPE_F32_SPLIT_F16W), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX),
                     ("f32_split_mix_f16w", pkg.DTYPE_F32_SPLIT_MIX_F16W), ("f32_split_mix_f16w_geglu2", pkg.DTYPE_F32_SPLIT_MIX_F16W_GEGLU2)):
        d = pkg.Diffuser(ctx, cfg, dt, seed=pkg.SEED_F16_WEIGHTS)
        trace = torch.zeros(4, 1, 4, 128, 128, device="cuda")
        d.set_trace(trace)
        out = d.sample_latent_with_inpainting(_cond(pkg, i, (1024, 1024)), 7.5, 4, reference.cuda(), mask.cuda(), i["noise"].cuda(),
                                              i["step_noise"].cuda())
        torch.cuda.synchronize()
        d.set_trace(None)
        rep[name] = dict(per_step=[errs(trace[k], ref_traj[k]) for k in range(4)], final=errs(out, ref))
        del d
        print(f"inpainting 1024^2, f16-representable weights, {name} vs oracle: per step {['%.2e' % s_['max_abs'] for s_ in rep[name]['per_step']]} = "
              + " ".join(f"{rep[name]['per_step'][k]['max_abs'] / lat_bound(ref_traj[k]):.2f}" for k in range(4)) + " of the bound")
    REPORT["inpainting_1024_f16_weights_vs_oracle"] = rep
    for k in range(4):
        assert rep["f32_split"]["per_step"][k]["max_abs"] <= lat_bound(ref_traj[k]), (k, rep["f32_split"]["per_step"][k])
        assert rep["f32_split_f16w"]["per_step"][k]["max_abs"] <= lat_bound(ref_traj[k]), (k, rep["f32_split_f16w"]["per_step"][k])      # (the same arithmetic on the f16 kernels)
        for nm in ("f32_split_mix", "f32_split_mix_f16w"):
            assert rep[nm]["per_step"][k]["max_abs"] <= 2.0 * lat_bound(ref_traj[k]), (nm, k, rep[nm]["per_step"][k])
        # SDXL_DTYPE_F32_SPLIT_MIX_F16W_GEGLU2 (round 6): the class that takes the mixed modes over the bound on this stress fixture -- the GEGLU projection, every
        # map that contains it on f16 operands is over, every map without it is under (profiles/r06_mix_classes_inpaint4.txt) -- with its activations as (hi, lo)
        # pairs: held at 1x, measured 0.83-0.92
        assert rep["f32_split_mix_f16w_geglu2"]["per_step"][k]["max_abs"] <= lat_bound(ref_traj[k]), (k, rep["f32_split_mix_f16w_geglu2"]["per_step"][k])


def test_refiner_1024_f16_representable_weights(pkg, ctx):
    """BASELINE configs[3] on the weights the reference's records hold (every parameter an f16 value, src/bin/sample/main.rs:37; fixtures
    oracle/make_golden_r6.py refiner1024_f16w / refine1024_f16w): one refiner UNet::forward at 1024^2 and Diffuser::refine_latent (2 iterations), the
    split engines and SDXL_DTYPE_F32_SPLIT_MIX_F16W -- the mode FOR these weights -- against the oracle on the same weights, the latter at 1x the bound."""
    gf, gr = os.path.join(GOLD, "fullsize_refiner1024_f16w.npz"), os.path.join(GOLD, "fullsize_refine1024_f16w.npz")
    if not (os.path.exists(gf) and os.path.exists(gr)):
        pytest.skip("tests/golden/fullsize_refine{r,}1024_f16w.npz not generated (python -m oracle.make_golden_r6, ~5 min)")
    cfg = pkg.sdxl_refiner_config()
    g = np.load(gf)
    x, t = seeded(1, 4, 128, 128, seed=141), torch.tensor([150], dtype=torch.int32)
    c, y = seeded(1, 77, cfg.context_dim, seed=142), seeded(1, cfg.adm_in_channels, seed=143)
    assert np.allclose(checksum(x, c, y), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    ref = torch.from_numpy(g["out"])
    rep = {"forward": {}, "refine_latent": {}}
    for name, dt, tol in (("f32_split", pkg.DTYPE_F32_SPLIT, F32_FWD_REL), ("f32_split_f16w", pkg.DTYPE_F32_SPLIT_F16W, F32_FWD_REL), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX, 3.0e-4),
                          ("f32_split_mix_f16w", pkg.DTYPE_F32_SPLIT_MIX_F16W, 4.0e-4)):
        u = pkg.UNet(ctx, cfg, dt, seed=pkg.SEED_F16_WEIGHTS | 0)
        if name in ("f32_split_mix_f16w", "f32_split_f16w"):
            assert u.mix_classes() != 0 and (name != "f32_split_mix_f16w" or u.mix_classes() & 4), "the F16W mode fell back on f16-representable weights"
        outs = [u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu() for _ in range(3)]     # eager, capture, replay
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), "hipGraph replay differs from the eager run"
        rep["forward"][name] = errs(outs[0], ref)
        del u
        print(f"refiner UNet::forward 1024^2, f16-representable weights, {name} vs oracle: rel {rep['forward'][name]['rel']:.3e} max-abs {rep['forward'][name]['max_abs']:.3e}")
        assert rep["forward"][name]["rel"] < tol, (name, rep["forward"][name])
    g = np.load(gr)
    i = dict(latent=seeded(1, 4, 128, 128, seed=151), noise=seeded(1, 4, 128, 128, seed=152), ctx=seeded(1, 77, cfg.context_dim, seed=153),
             uctx=seeded(77, cfg.context_dim, seed=154), y=seeded(1, cfg.adm_in_channels, seed=155), uy=seeded(cfg.adm_in_channels, seed=156))
    assert np.allclose(checksum(*i.values()), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    ref_traj, ref = torch.from_numpy(g["traj"]), torch.from_numpy(g["latent"])
    for name, dt in (("f32_split", pkg.DTYPE_F32_SPLIT), ("f32_split_f16w", pkg.DTYPE_F32_SPLIT_F16W), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX), ("f32_split_mix_f16w", pkg.DTYPE_F32_SPLIT_MIX_F16W)):
        d = pkg.Diffuser(ctx, cfg, dt, seed=pkg.SEED_F16_WEIGHTS | 0)
        trace = torch.zeros(2, 1, 4, 128, 128, device="cuda")
        d.set_trace(trace)
        out = d.refine_latent(i["latent"].cuda(), _refiner_cond(pkg, i), 7.5, 800, 10, i["noise"].cuda())
        torch.cuda.synchronize()
        d.set_trace(None)
        rep["refine_latent"][name] = dict(per_step=[errs(trace[k], ref_traj[k]) for k in range(2)], final=errs(out, ref))
        del d
        print(f"refine_latent 1024^2, f16-representable weights, {name} vs oracle: per step {['%.2e' % s_['max_abs'] for s_ in rep['refine_latent'][name]['per_step']]} = "
              + " ".join(f"{rep['refine_latent'][name]['per_step'][k]['max_abs'] / lat_bound(ref_traj[k]):.2f}" for k in range(2)) + " of the bound")
        for k in range(2):       # every mode at 1x: the strict modes' bar
            assert rep["refine_latent"][name]["per_step"][k]["max_abs"] <= lat_bound(ref_traj[k]), (name, k, rep["refine_latent"][name]["per_step"][k])
    REPORT["refiner_1024_f16_weights_vs_oracle"] = rep


CONFIG5_KEEP = tuple(range(9, 100, 10))


@pytest.mark.parametrize("weights", ["f16w", "fp32"])
def test_config5_inpainting_100_steps(pkg, ctx, weights):
    """BASELINE configs[4] AT ITS OWN STEP COUNT (oracle/make_golden_r6.py config5[_f16w], ~80 min of oracle time each): Diffuser::sample_latent_with_inpainting at
    1024x1024, n_steps = 100 (100 CFG-7.5 pairs, t = 999, 989 ... 9), mask = latent rows 0..24 generated (the 200 px crop), reference latent = the oracle's
    encode, explicit per-step re-noise (stablediffusion/mod.rs:434-483).  Every 10th latent and the final one against the oracle, EVERY mode at 1x the scaled bound:
    F32_SPLIT, F32_SPLIT_MIX (fp32 weights) / F32_SPLIT_MIX_F16W (f16-representable weights).  The 4-step fixtures of rounds 3 / 5 (test_inpainting_1024_*) take
    250-step jumps, which multiply one forward's error by ~2.3 per step before CFG; the 10-step jumps of the configuration BASELINE names multiply it by ~0.2."""
    gp = os.path.join(GOLD, "fullsize_config5_f16w.npz" if weights == "f16w" else "fullsize_config5.npz")
    if not os.path.exists(gp):
        pytest.skip(f"{os.path.basename(gp)} not generated (python -m oracle.make_golden_r6 config5{'_f16w' if weights == 'f16w' else ''}, ~80 min)")
    g = np.load(gp)
    cfg = pkg.sdxl_base_config()
    i = dict(noise=seeded(1, 4, 128, 128, seed=181), ctx=seeded(1, 77, cfg.context_dim, seed=182), uctx=seeded(77, cfg.context_dim, seed=183),
             y=seeded(1, cfg.adm_in_channels, seed=184), uy=seeded(cfg.adm_in_channels, seed=185), step_noise=seeded(100, 1, 4, 128, 128, seed=186))
    assert np.allclose(checksum(*i.values()), g["in_checksum"], rtol=1e-9), "torch CPU generator changed: regenerate the fixtures"
    assert tuple(int(k) for k in g["steps"]) == CONFIG5_KEEP and pkg.step_count(100) == 100
    reference = torch.from_numpy(np.load(os.path.join(GOLD, "fullsize_inpaint1024.npz"))["reference"])
    mask = torch.zeros(1, 4, 128, 128, dtype=torch.bool)
    mask[:, :, 0:25, :] = True
    ref_traj, ref = torch.from_numpy(g["traj"]).clone(), torch.from_numpy(g["latent"])
    alphas = pkg.default_alphas_cumprod()
    ts = list(range(999, -1, -10))
    for j, k in enumerate(CONFIG5_KEEP):      # (the engine's trace holds the blend for the NEXT iteration: see test_inpainting_1024_matches_oracle)
        if k + 1 < 100:
            a_n = float(alphas[ts[k + 1]])
            ref_traj[j] = torch.where(mask, ref_traj[j], reference * (a_n ** 0.5) + i["step_noise"][k + 1] * ((1.0 - a_n) ** 0.5))
    seed = pkg.SEED_F16_WEIGHTS if weights == "f16w" else 0
    modes = (("f32_split", pkg.DTYPE_F32_SPLIT), ("f32_split_f16w", pkg.DTYPE_F32_SPLIT_F16W), ("f32_split_mix_f16w", pkg.DTYPE_F32_SPLIT_MIX_F16W),
             ("f32_split_mix_f16w_geglu2", pkg.DTYPE_F32_SPLIT_MIX_F16W_GEGLU2)) if weights == "f16w" else \
            (("f32_split", pkg.DTYPE_F32_SPLIT), ("f32_split_mix", pkg.DTYPE_F32_SPLIT_MIX), ("f16", pkg.DTYPE_F16))
    rep = {}
    for name, dt in modes:
        d = pkg.Diffuser(ctx, cfg, dt, seed=seed)
        trace = torch.zeros(100, 1, 4, 128, 128, device="cuda")
        d.set_trace(trace)
        out = d.sample_latent_with_inpainting(_cond(pkg, i, (1024, 1024)), 7.5, 100, reference.cuda(), mask.cuda(), i["noise"].cuda(), i["step_noise"].cuda())
        torch.cuda.synchronize()
        d.set_trace(None)
        assert torch.equal(trace[-1], out)
        rep[name] = dict(per_step=[errs(trace[k], ref_traj[j]) for j, k in enumerate(CONFIG5_KEEP)], final=errs(out, ref))
        del d
        print(f"config 5 (100-step inpainting), {weights} weights, {name} vs oracle: final max-abs {rep[name]['final']['max_abs']:.3e} (|ref| {rep[name]['final']['ref_max']:.1f}, bound "
              f"{lat_bound(ref):.3e}); of the bound per kept step: " + " ".join(f"{rep[name]['per_step'][j]['max_abs'] / lat_bound(ref_traj[j]):.2f}" for j in range(len(CONFIG5_KEEP))))
    REPORT[f"config5_100_steps_{weights}_vs_oracle"] = rep
    for name, _ in modes:
        if name == "f16":
            assert rep[name]["final"]["rel"] < F16_TRAJ_REL["f16"], rep[name]["final"]
            continue
        for j in range(len(CONFIG5_KEEP)):
            assert rep[name]["per_step"][j]["max_abs"] <= lat_bound(ref_traj[j]), (name, CONFIG5_KEEP[j], rep[name]["per_step"][j])
        assert rep[name]["final"]["max_abs"] <= lat_bound(ref), (name, rep[name]["final"])


def test_unet_forward_1024_f16_representable_weights(pkg, ctx):
    """Real SDXL records hold f16 parameters (HalfPrecisionSettings, src/bin/sample/main.rs:37): with such weights the engine's
    f16 weight rounding is exact.  Same forward as test_unet_forward_1024_matches_oracle with every parameter rounded to f16 on
    both sides (oracle fixture / SDXL_SEED_F16_WEIGHTS): what is left in the f16 modes is ACTIVATION rounding alone."""
    g = np.load(os.path.join(GOLD, "fullsize_unet1024_f16w.npz"))
    cfg = pkg.sdxl_base_config()
    x, t = seeded(1, 4, 128, 128, seed=111), torch.tensor([500], dtype=torch.int32)
    c, y = seeded(1, 77, cfg.context_dim, seed=112), seeded(1, cfg.adm_in_channels, seed=113)
    assert np.allclose(checksum(x, c, y), g["in_checksum"], rtol=1e-9)
    ref = torch.from_numpy(g["out"])
    rep = {}
    for name, dt, tol in (("f32", pkg.DTYPE_F32, F32_FWD_REL), ("f32_split", pkg.DTYPE_F32_SPLIT, F32_FWD_REL), ("f32_split_f16w", pkg.DTYPE_F32_SPLIT_F16W, F32_FWD_REL), ("f16", pkg.DTYPE_F16, 2.4e-3),
                          ("f16_f32res", pkg.DTYPE_F16_F32RES, 1.3e-3)):   # measured 3.1e-6 / (split: two MFMAs per product on these weights) / 1.17e-3 / 6.4e-4
        u = pkg.UNet(ctx, cfg, dt, seed=pkg.SEED_F16_WEIGHTS | 0)
        rep[name] = errs(u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu(), ref)
        del u
        print(f"UNet::forward 1024^2, f16-representable weights, {name} vs oracle: rel {rep[name]['rel']:.3e} rms-rel {rep[name]['rms_rel']:.3e}")
        assert rep[name]["rel"] < tol, (name, rep[name])
    REPORT["unet_forward_1024_f16_weights_vs_oracle"] = rep
