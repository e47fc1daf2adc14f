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
    Bar: north_star's 1e-3 on latents at SDXL's latent scale.  The synthetic-weight UNet is not a denoiser: its
    trajectories grow well past real SDXL latents (|x| <~ 4), so the absolute bound is scaled with the trajectory:
    1e-3 * max(1, max|latent_ref| / 4); both the raw absolute and the relative error are recorded.
  * SDXL_DTYPE_F16 / SDXL_DTYPE_F16_F32RES (the speed modes) against the now oracle-anchored F32 engine on the SAME 31-step
    1024x1024 CFG-7.5 trajectory: per-step max-abs / relative / rms error -> gpurun_out/r02_drift_*.json (committed under
    profiles/).  fp16 operands cannot meet 1e-3 absolute (one rounding is 4.9e-4 relative, CFG 7.5 multiplies the error of
    eps by up to 7.5 per step); the bound asserted here is the measured class with headroom, stated in DESIGN.md section 5.
Everything goes through the C ABI (ctypes); measured numbers are written to gpurun_out/r02_parity_baseline.json.
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
F16_FWD_REL = 3e-2           # one UNet::forward, fp16 operands (measured value is recorded)
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
        path = os.path.join(OUT_DIR, "r02_parity_baseline.json")
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
    for name, dt, tol in (("f32", pkg.DTYPE_F32, F32_FWD_REL), ("f16", pkg.DTYPE_F16, F16_FWD_REL), ("f16_f32res", pkg.DTYPE_F16_F32RES, F16_FWD_REL)):
        u = pkg.UNet(ctx, cfg, dt, seed=0)
        outs[name] = u.forward(x.cuda(), t.cuda(), c.cuda(), y.cuda()).cpu()
        rep[name] = errs(outs[name], ref)
        del u
        print(f"UNet::forward 1024^2 {name} vs oracle: rel {rep[name]['rel']:.3e} rms-rel {rep[name]['rms_rel']:.3e} max-abs {rep[name]['max_abs']:.3e}")
        assert rep[name]["rel"] < tol, (name, rep[name])
    rep["f16_vs_f32_engine"] = errs(outs["f16"], outs["f32"])
    REPORT["unet_forward_1024_vs_oracle"] = rep


def test_decode_1024_matches_oracle(pkg, ctx):
    """LatentDecoder::latent_to_image at 1024x1024 (Decoder::forward, autoencoder/mod.rs:203-216)"""
    g = np.load(os.path.join(GOLD, "fullsize_decode1024.npz"))
    latent = seeded(1, 4, 128, 128, seed=121)
    assert np.allclose(checksum(latent), g["in_checksum"], rtol=1e-9)
    rep = {}
    for name, dt in (("f32", pkg.DTYPE_F32), ("f16", pkg.DTYPE_F16)):
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
    assert rep["f16"]["image_sub"]["rel"] < 3e-2


def test_config2_trajectory_parity_and_drift(pkg, ctx):
    """the benchmarked configuration (BASELINE configs[1]): 1024x1024, n_steps=30 -> 31 CFG pairs, CFG 7.5.
    F32 engine against the ORACLE's trajectory, then the speed modes against the F32 engine step by step."""
    cfg = pkg.sdxl_base_config()
    i = _inputs(cfg, 130, 128)
    n_it = pkg.step_count(30)
    assert n_it == 31
    trajs, secs = {}, {}
    for name, dt in (("f32", pkg.DTYPE_F32), ("f16", pkg.DTYPE_F16), ("f16_f32res", pkg.DTYPE_F16_F32RES)):
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
    for name in ("f16", "f16_f32res"):
        per = [errs(trajs[name][k], trajs["f32"][k]) for k in range(n_it)]
        rep[name + "_vs_f32"] = per
        print(f"config 2 drift {name} vs F32 engine: step 0 rel {per[0]['rel']:.2e}, step 15 {per[15]['rel']:.2e}, final {per[-1]['rel']:.2e} "
              f"(max-abs {per[-1]['max_abs']:.3e}, rms-rel {per[-1]['rms_rel']:.2e}); {secs[name]:.2f} s")
        try:
            os.makedirs(OUT_DIR, exist_ok=True)
            with open(os.path.join(OUT_DIR, f"r02_drift_{name}.json"), "w") as fh:
                json.dump(dict(config="SDXL-base 1024x1024, n_steps=30 (31 iterations), CFG 7.5, synthetic weights seed 0",
                               reference="SDXL_DTYPE_F32 engine trajectory (oracle-anchored: f32_vs_oracle in r02_parity_baseline.json)",
                               mode=name, per_step=per, ref_absmax=rep["ref_absmax"]), fh, indent=1)
        except OSError:
            pass
        assert per[-1]["rel"] < 0.25, (name, per[-1])        # measured class recorded in DESIGN.md section 5; fp16 operands
    REPORT["config2_trajectory"] = rep
