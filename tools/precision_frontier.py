"""Precision frontier of the UNet on BASELINE configs[1] (1024^2, 31 CFG-7.5 steps): which GEMM classes carry the f16 mode's error, and is
there a per-class precision map that meets north_star's two numbers at once (latents within lat_bound AND >= 1.0 img/s)?

    python tools/precision_frontier.py [out.json]                (GPU box; ~3 minutes)

Instrument: the split-operand engine (SDXL_DTYPE_F32_SPLIT: fp32 stream, (hi, lo) f16 operand pairs) with sdxl_debug_set("hl_demote", mask)
-- the GEMMs of the classes in `mask` run on operands whose lo halves are zero, i.e. with exactly the f16 engine's operand rounding
(f16 x f16 products, fp32 accumulation), every other class keeps fp32-class operands (csrc/engine.h DemoteClass, csrc/unet.cpp).  Each point
is the full config-2 trajectory against the committed oracle fixture (tests/golden/fullsize_config2.npz).  Points:
  * none (= the split engine) and all (= f16 operands everywhere on an fp32 stream: the F16_F32RES arithmetic),
  * every class alone demoted from split to f16 (11 points: six transformer classes + five kinds of convolution),
  * every class alone promoted from f16 to split, i.e. all others demoted (11 points),
  * the best maps the per-class figures predict (quadrature model), measured directly.
Step time of a map is MODELLED from measured per-class times (the instrument runs the split kernels whatever the mask): the eager per-launch
profiles of the split engine and of the F16_F32RES engine (same fp32 stream, unfused LayerNorms) are classified by launch shape, and a map's
step = split step - sum over demoted classes (split class ms - f16 class ms); images/s = 1 / (31 steps + the measured f32_split decode).
"""
import csv
import itertools
import json
import os
import statistics
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

CLASSES = ["qkv", "attn", "out", "xattn", "geglu", "ff", "conv_res", "conv_skip", "conv_io", "conv_updown", "conv_proj"]       # bit i of the mask (engine.h DemoteClass)
BIT = {c: 1 << i for i, c in enumerate(CLASSES)}
ALL = (1 << len(CLASSES)) - 1
BITNAME = {v: k for k, v in BIT.items()}


def seeded(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def label_rows(rows):
    """per-launch profile rows (class, M, N, K, ksize, ms, tag) -> frontier class of every launch (+ 'norm' / 'other'); tag = the DemoteClass bit
    the UNet driver attaches to every GEMM / attention launch"""
    labels = []
    for r in rows:
        cls, tag = int(r["class"]), int(r["tag"])
        if cls in (2, 3):
            k = "norm"
        elif tag and tag in BITNAME:
            k = BITNAME[tag]
        else:
            k = "other"
        labels.append(k)
    return labels


def class_profile(pkg, ctx, cfg, dt, cond, noise, seed=0):
    d = pkg.Diffuser(ctx, cfg, dt, seed=seed)
    d.sample_latent(cond, 7.5, 2, noise)
    with tempfile.NamedTemporaryFile(suffix=".csv", delete=False) as f:
        path = f.name
    os.environ["SDXL_PROFILE_DUMP"] = path
    d.diffusion.profile(2, 128, 128)
    del os.environ["SDXL_PROFILE_DUMP"]
    rows = list(csv.DictReader(open(path)))
    os.unlink(path)
    labels = label_rows(rows)
    cls = {c: 0.0 for c in CLASSES + ["norm", "other"]}
    cnt = {c: 0 for c in cls}
    for r, k in zip(rows, labels):
        cls[k] += float(r["ms"]); cnt[k] += 1
    # graph-replayed step of the same engine: the eager per-launch times carry the event overhead -> subtracted per launch
    d.enable_step_timing(True)
    d.sample_latent(cond, 7.5, 30, noise)
    step = statistics.median(d.step_times_ms())
    tot = sum(cls.values())
    over = max(tot - step, 0.0) / max(len(rows), 1)
    adj = {c: max(cls[c] - cnt[c] * over, 0.0) for c in cls}
    del d
    return {"class_ms": {c: round(v, 3) for c, v in adj.items()}, "launches": cnt, "step_ms_p50": round(step, 3),
            "eager_sum_ms": round(tot, 3), "event_overhead_us_per_launch": round(1e3 * over, 3)}


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_precision_frontier.json")
    pkg = ge.load_package()
    ctx = pkg.Context(0)
    cfg = pkg.sdxl_base_config()
    g = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_config2.npz"))
    ref = torch.from_numpy(g["latent"])
    steps = [int(s) for s in g["steps"]]
    traj = torch.from_numpy(g["traj"])
    i = dict(noise=seeded(1, 4, 128, 128, seed=131), ctx=seeded(1, 77, cfg.context_dim, seed=132), uctx=seeded(77, cfg.context_dim, seed=133),
             y=seeded(1, cfg.adm_in_channels, seed=134), uy=seeded(cfg.adm_in_channels, seed=135))

    def cond():
        return pkg.Conditioning(context_full=i["ctx"].cuda(), channel_context=i["y"].cuda(), unconditional_context_full=i["uctx"].cuda(),
                                unconditional_channel_context=i["uy"].cuda(), resolution=(1024, 1024))
    noise = i["noise"].cuda()
    lat_bound = 1e-3 * max(1.0, float(ref.abs().max()) / 4.0)

    res = {"workload": "BASELINE configs[1]: SDXL-base 1024x1024, n_steps=30 (31 CFG-7.5 step pairs), oracle fixture tests/golden/fullsize_config2.npz",
           "lat_bound_abs": lat_bound, "ref_absmax": float(ref.abs().max()), "classes": CLASSES,
           "instrument": "SDXL_DTYPE_F32_SPLIT engine + sdxl_debug_set('hl_demote', mask): demoted classes multiply f16-rounded operands (lo halves zero)"}

    # ---- per-class times of the two engines a map is made of
    res["times_split"] = class_profile(pkg, ctx, cfg, pkg.DTYPE_F32_SPLIT, cond(), noise)
    res["times_f16_f32res"] = class_profile(pkg, ctx, cfg, pkg.DTYPE_F16_F32RES, cond(), noise)
    res["times_f16"] = class_profile(pkg, ctx, cfg, pkg.DTYPE_F16, cond(), noise)
    dec = pkg.LatentDecoder(ctx, None, pkg.DTYPE_F32_SPLIT, seed=0)
    lat0 = torch.randn(1, 4, 128, 128, device="cuda")
    dec.latent_to_image(lat0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dec.latent_to_image(lat0)
    e1.record()
    torch.cuda.synchronize()
    decode_ms = e0.elapsed_time(e1) / 3
    res["decode_ms_f32_split"] = round(decode_ms, 2)
    del dec
    ts, tf = res["times_split"], res["times_f16_f32res"]

    def model(mask):
        step = ts["step_ms_p50"] - sum(ts["class_ms"][c] - tf["class_ms"][c] for c in CLASSES if mask & BIT[c])
        return round(step, 3), round(1e3 / (31 * step + decode_ms + 8.0), 4)     # (+ 8 ms: set_context, DDIM kernels, copies -- the f16 line's own residue)

    # ---- measured points
    d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F32_SPLIT, seed=0)
    trace = torch.zeros(31, 1, 4, 128, 128, device="cuda")
    d.set_trace(trace)
    d.enable_step_timing(True)

    def point(mask, label):
        pkg.debug_set("hl_demote", mask)
        lat = d.sample_latent(cond(), 7.5, 30, noise)
        torch.cuda.synchronize()
        dd = (lat.cpu() - ref).abs().max().item()
        per = [float((trace[k].cpu() - traj[j]).abs().max()) for j, k in enumerate(steps)]
        step, ips = model(mask)
        p = {"label": label, "mask": mask, "demoted": [c for c in CLASSES if mask & BIT[c]], "final_abs": dd, "final_rel": dd / float(ref.abs().max()),
             "per_kept_step_abs": per, "finite": bool(torch.isfinite(lat).all()), "meets_lat_bound": bool(dd <= lat_bound),
             "instrument_step_ms": round(statistics.median(d.step_times_ms()), 2), "modelled_step_ms": step, "modelled_images_per_sec": ips}
        print(f"[frontier] {label:28s} mask {mask:3d} final abs {dd:.4e} rel {p['final_rel']:.3e}  modelled {step:6.2f} ms/step {ips:.3f} img/s", flush=True)
        return p

    pts = [point(0, "split (no class demoted)"), point(ALL, "all classes f16 operands")]
    for c in CLASSES:
        pts.append(point(BIT[c], f"only {c} demoted"))
    for c in CLASSES:
        pts.append(point(ALL & ~BIT[c], f"only {c} promoted"))
    res["points"] = pts

    # ---- quadrature model from the single-class demotions: err(map)^2 = e0^2 + sum_c (e_c^2 - e0^2); check it on the measured maps
    e0 = pts[0]["final_abs"]
    e1c = {c: pts[2 + k]["final_abs"] for k, c in enumerate(CLASSES)}
    var = {c: max(e1c[c] ** 2 - e0 ** 2, 0.0) for c in CLASSES}

    def predict(mask):
        return (e0 ** 2 + sum(var[c] for c in CLASSES if mask & BIT[c])) ** 0.5
    res["quadrature_check"] = [{"label": p["label"], "measured": p["final_abs"], "predicted": predict(p["mask"])} for p in pts]
    maps = []
    for mask in range(ALL + 1):
        step, ips = model(mask)
        maps.append({"mask": mask, "demoted": [c for c in CLASSES if mask & BIT[c]], "predicted_abs": predict(mask), "modelled_step_ms": step,
                     "modelled_images_per_sec": ips})
    feas = [m for m in maps if m["predicted_abs"] <= lat_bound and m["modelled_images_per_sec"] >= 1.0]
    res["maps_meeting_both_predicted"] = sorted(feas, key=lambda m: -m["modelled_images_per_sec"])[:8]
    # the fastest maps inside lat_bound and the most accurate maps at >= 1.0 img/s, measured directly
    inside = sorted([m for m in maps if m["predicted_abs"] <= lat_bound], key=lambda m: -m["modelled_images_per_sec"])[:3]
    fast = sorted([m for m in maps if m["modelled_images_per_sec"] >= 1.0], key=lambda m: m["predicted_abs"])[:3]
    res["fastest_inside_lat_bound"] = [dict(m, measured=point(m["mask"], "fastest inside lat_bound (predicted)")) for m in inside if m["mask"] not in (0,)]
    res["most_accurate_at_1_img_s"] = [dict(m, measured=point(m["mask"], "most accurate at >= 1 img/s (modelled)")) for m in fast]
    pkg.debug_set("hl_demote", 0)
    meets = [m for m in res["fastest_inside_lat_bound"] + res["most_accurate_at_1_img_s"]
             if m["measured"]["meets_lat_bound"] and m["measured"]["modelled_images_per_sec"] >= 1.0]
    res["verdict"] = ("a per-class map meets both: " + json.dumps(meets[0]["demoted"])) if meets else \
        "no per-class precision map meets lat_bound and >= 1.0 img/s together (measured; see points / maps)"
    print("[frontier]", res["verdict"], flush=True)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
