"""Round-5 fixtures: the oracle in the REFERENCE'S OWN f16 arithmetic.  TEST INFRASTRUCTURE.

    python -m oracle.make_golden_r5 [unet1024_f16ref] [config2_f16ref] [classes]

The reference benchmarks / ships ``LibTorch<f16>`` for the UNet (src/bin/sample/main.rs:122): every tensor -- the record's f16
parameters (``HalfPrecisionSettings``, :37) and the output of every burn op -- is an f16 value.  ``oracle.model.NUM`` (mode
"f16ref") emulates exactly that on the fp32 restatement: parameters rounded to f16, the output of every op of the graph rounded
to f16 at the granularity the reference issues them, fp32 inside an op (libtorch's half kernels accumulate in fp32).  The distance
of THAT run from the fp32 oracle is the numerical class of the reference's own GPU path; the engine's f16 mode is asserted to stay
inside it (tests/test_gpu_baseline_parity.py), so the f16 bounds stop being "2x whatever the engine measured".

  fullsize_unet1024_f16ref.npz   one UNet::forward at 1024^2, inputs of fullsize_unet1024.npz: the f16ref output, its error against
                                 the fp32 oracle output (max-abs / rel), |out|max
  fullsize_config2_f16ref.npz    the 31-step CFG-7.5 trajectory of fullsize_config2.npz in f16ref arithmetic (~30 min on 8 cores):
                                 final latent + the kept steps, error of each against the fp32 trajectory
  fullsize_unet1024_classes.npz  the same forward in "operands" emulation (the ENGINE's f16 arithmetic seen from the oracle: only
                                 the GEMM operands of one class rounded to f16): error per class alone and for all classes together
"""
import os
import sys
import time

import numpy as np
import torch

from . import config as OC, model as OM, pipeline as OP
from .make_golden_fullsize import OUT, CONFIG2_KEEP, base_weights, checksum, config2_inputs, unet1024_inputs

CLASSES = ("qkv", "attn", "out", "xattn", "geglu", "ff", "conv_res", "conv_skip", "conv_io", "conv_updown", "conv_proj")


def _err(a, ref):
    d = float((a - ref).abs().max())
    return d, d / float(ref.abs().max())


def run_unet1024_f16ref(cfg, W):
    i = unet1024_inputs(cfg)
    ref = torch.from_numpy(np.load(os.path.join(OUT, "fullsize_unet1024.npz"))["out"])
    OM.NUM.set("f16ref")
    t0 = time.time()
    out = OM.unet_forward(cfg, W, i["x"], i["t"], i["ctx"], i["y"])
    OM.NUM.set(None)
    a, r = _err(out, ref)
    print(f"[golden r5] UNet::forward 1024^2 in f16ref arithmetic: {time.time() - t0:.1f} s, vs fp32 oracle max-abs {a:.4e} rel {r:.4e}, "
          f"finite {bool(torch.isfinite(out).all())}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_unet1024_f16ref.npz"), out=out.numpy(), err_abs=np.array([a]), err_rel=np.array([r]),
                        in_checksum=checksum(i["x"], i["ctx"], i["y"]))


def run_classes(cfg, W):
    i = unet1024_inputs(cfg)
    ref = torch.from_numpy(np.load(os.path.join(OUT, "fullsize_unet1024.npz"))["out"])
    res = {}
    only = [c for c in sys.argv[1:] if c in CLASSES]
    for cl in ([(c,) for c in only] if only else [(c,) for c in CLASSES] + [("conv",), CLASSES]):
        OM.NUM.set("operands", cl)
        out = OM.unet_forward(cfg, W, i["x"], i["t"], i["ctx"], i["y"])
        OM.NUM.set(None)
        res["+".join(cl)] = _err(out, ref)
        print(f"[golden r5] operands {'+'.join(cl):40s} max-abs {res['+'.join(cl)][0]:.4e} rel {res['+'.join(cl)][1]:.4e}", flush=True)
    if only:
        return
    np.savez_compressed(os.path.join(OUT, "fullsize_unet1024_classes.npz"), names=np.array(list(res)), err_abs=np.array([v[0] for v in res.values()]),
                        err_rel=np.array([v[1] for v in res.values()]))


def run_config2_f16ref(cfg, W):
    i = config2_inputs(cfg)
    g = np.load(os.path.join(OUT, "fullsize_config2.npz"))
    cond = OP.Conditioning(i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], None, (1024, 1024))
    trace = []
    OM.NUM.set("f16ref")
    t0 = time.time()
    lat = OP.Diffuser(cfg, W, OC.alphas_cumprod()).sample_latent(cond, 7.5, 30, i["noise"], trace)
    OM.NUM.set(None)
    dt = time.time() - t0
    ref = torch.from_numpy(g["latent"])
    a, r = _err(lat, ref)
    steps = [int(k) for k in g["steps"]]
    per = np.array([_err(trace[k], torch.from_numpy(g["traj"][j])) for j, k in enumerate(steps)])
    print(f"[golden r5] config 2 in f16ref arithmetic: {dt:.1f} s, final latent vs fp32 oracle max-abs {a:.4e} rel {r:.4e}; per kept step "
          f"{np.array2string(per[:, 0], precision=3)}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_config2_f16ref.npz"), steps=np.array(CONFIG2_KEEP), latent=lat.numpy(),
                        traj=np.stack([trace[k].numpy() for k in CONFIG2_KEEP]), err_abs=np.array([a]), err_rel=np.array([r]),
                        per_step_err=per, in_checksum=checksum(*i.values()), oracle_seconds=np.array([dt]))


def main():
    what = set(sys.argv[1:]) or {"unet1024_f16ref"}
    if what <= {"config2b", "config2b_f16w", "inpaint1024_f16w"}:
        return          # (handled at the end of the module)
    cfg, W = base_weights()
    if "unet1024_f16ref" in what:
        run_unet1024_f16ref(cfg, W)
    if "classes" in what:
        run_classes(cfg, W)
    if "config2_f16ref" in what:
        run_config2_f16ref(cfg, W)


if __name__ == "__main__":
    main()


def config2b_inputs(cfg):
    """a SECOND prompt / noise for BASELINE configs[1] (seeds 231..235): the mixed mode sits 1.13x under the scaled bound on the first one -- one
    trajectory is not a distribution (tests/test_gpu_baseline_parity.py::test_config2_second_prompt)"""
    from .make_golden_fullsize import seeded
    return dict(noise=seeded(1, 4, 128, 128, seed=231), ctx=seeded(1, 77, cfg.context_dim, seed=232), uctx=seeded(77, cfg.context_dim, seed=233),
                y=seeded(1, cfg.adm_in_channels, seed=234), uy=seeded(cfg.adm_in_channels, seed=235))


def run_config2b(cfg, W):
    i = config2b_inputs(cfg)
    cond = OP.Conditioning(i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], None, (1024, 1024))
    trace = []
    t0 = time.time()
    lat = OP.Diffuser(cfg, W, OC.alphas_cumprod()).sample_latent(cond, 7.5, 30, i["noise"], trace)
    dt = time.time() - t0
    print(f"[golden r5] config 2, second prompt: {dt:.1f} s, |latent|max {float(lat.abs().max()):.2f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_config2b.npz"), steps=np.array(CONFIG2_KEEP), latent=lat.numpy(),
                        traj=np.stack([trace[k].numpy() for k in CONFIG2_KEEP]), in_checksum=checksum(*i.values()), oracle_seconds=np.array([dt]))


def run_config2b_f16w(cfg, W):
    """the second prompt on f16-representable weights (every parameter rounded to IEEE f16 first, as oracle.make_golden_r3.run_config2_f16w): the weights
    SDXL_DTYPE_F32_SPLIT_MIX_F16W is for -- its 'inside the bound' claim rested on one prompt (tests/test_gpu_baseline_parity.py::test_config2_second_prompt_f16_weights)"""
    i = config2b_inputs(cfg)
    W16 = {k: (v if k.endswith(".eps") else v.half().float()) for k, v in W.items()}
    cond = OP.Conditioning(i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], None, (1024, 1024))
    trace = []
    t0 = time.time()
    lat = OP.Diffuser(cfg, W16, OC.alphas_cumprod()).sample_latent(cond, 7.5, 30, i["noise"], trace)
    dt = time.time() - t0
    print(f"[golden r5] config 2, second prompt, f16-representable weights: {dt:.1f} s, |latent|max {float(lat.abs().max()):.2f}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_config2b_f16w.npz"), steps=np.array(CONFIG2_KEEP), latent=lat.numpy(),
                        traj=np.stack([trace[k].numpy() for k in CONFIG2_KEEP]), in_checksum=checksum(*i.values()), oracle_seconds=np.array([dt]))


def run_inpaint1024_f16w(cfg, W):
    """the inpainting fixture of oracle.make_golden_r3.run_inpaint1024 (4 CFG-7.5 steps, same inputs and reference latent) on f16-representable UNet weights:
    where SDXL_DTYPE_F32_SPLIT_MIX_F16W stands on the configuration whose 250-step jumps amplify a forward's error most"""
    from .make_golden_r3 import inpaint1024_inputs, inpaint_mask
    i = inpaint1024_inputs(cfg)
    ref_latent = torch.from_numpy(np.load(os.path.join(OUT, "fullsize_inpaint1024.npz"))["reference"])
    W16 = {k: (v if k.endswith(".eps") else v.half().float()) for k, v in W.items()}
    cond = OP.Conditioning(i["uctx"], None, i["ctx"], None, i["uy"], None, i["y"], None, (1024, 1024))
    trace = []
    t0 = time.time()
    out = OP.Diffuser(cfg, W16, OC.alphas_cumprod()).sample_latent_with_inpainting(
        cond, 7.5, 4, ref_latent, inpaint_mask(), i["noise"], [i["step_noise"][k] for k in range(4)], trace)
    dt = time.time() - t0
    print(f"[golden r5] inpainting 1024^2, f16-representable weights: {dt:.1f} s, |latent|max per step {[round(float(t.abs().max()), 2) for t in trace]}", flush=True)
    np.savez_compressed(os.path.join(OUT, "fullsize_inpaint1024_f16w.npz"), traj=np.stack([t.numpy() for t in trace]), latent=out.numpy(),
                        in_checksum=checksum(*i.values()), oracle_seconds=np.array([dt]))


if __name__ == "__main__" and "inpaint1024_f16w" in sys.argv:
    run_inpaint1024_f16w(*base_weights())

if __name__ == "__main__" and ("config2b" in sys.argv or "config2b_f16w" in sys.argv):
    _cfg, _W = base_weights()
    if "config2b" in sys.argv:
        run_config2b(_cfg, _W)
    if "config2b_f16w" in sys.argv:
        run_config2b_f16w(_cfg, _W)
