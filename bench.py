#!/usr/bin/env python
"""bench.py -- images/sec of the SDXL-base sampling hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input per GPU: ONE 1024x1024 image =
`Diffuser::sample_latent` with n_steps=30 (which the reference executes as 31 CFG UNet step pairs, stablediffusion/mod.rs:
400-406) at CFG 7.5, followed by `LatentDecoder::latent_to_image` -- BASELINE.json configs[1] (429.7 TFLOP/image, SURVEY
section 8d).  Inputs (conditioning, noise) are synthetic seeded tensors already resident in HBM when the timed region
starts; weights are the seeded synthetic SDXL-base architecture (no checkpoint can be downloaded here).  With N>1 every
rank renders its own prompt (weak scaling, no collective in the loop) after ONE RCCL broadcast of the packed weight arena
from rank 0 over xGMI; value = total images of all ranks / max-over-ranks wall time.

Extra objects on the JSON line:
  roofline      live hipEvent measurement of the dominant kernel (the implicit-GEMM conv/linear kernel, which carries
                ~94% of the algorithmic FLOPs): achieved = sum of its algorithmic FLOPs (2*M*N*K per launch, unpadded) /
                sum of its launch durations over one UNet forward of the timed configuration; peak = 2500 TFLOP/s
                (dense fp16 MFMA, MI355X_MICROARCH.md).
  cpu_baseline  the oracle (torch-CPU fp32 restatement of the reference graph; the reference's burn-ndarray path cannot
                be built here) timed on this box's host cores on a bounded sample, rank 0 / N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TFLOP_PER_UNET_FWD_1024 = 6.761      # SURVEY section 8(d), base UNet, latent 128x128, B=1
TFLOP_VAE_DECODE_1024 = 10.470
TFLOP_PER_UNET_FWD_512 = 1.589
PEAK_F16_TFLOPS = 2500.0             # dense fp16 MFMA peak, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


# ------------------------------------------------------------------------------------------ distributed helpers
def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_dist(backend: str):
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def broadcast_arena(arena_u8, src: int = 0, chunk_bytes: int = 1 << 30):
    """one-time weight broadcast (RCCL over xGMI on GPUs, gloo in the CPU tests), in <=1 GiB pieces"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    n = arena_u8.numel()
    for o in range(0, n, chunk_bytes):
        dist.broadcast(arena_u8[o:min(n, o + chunk_bytes)], src=src)


def max_over_ranks(seconds: float, device) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def prompt_seed(rank: int, step: int) -> int:
    """prompt i -> GPU i mod N: every (rank, step) pair is an independent synthetic prompt"""
    return 1000 + 97 * rank + step


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(res: int, max_seconds: float = 40.0):
    """oracle timed on the host cores on a bounded sample: ONE UNet::forward (fp32 torch-CPU) at 512^2, extrapolated to
    an image by FLOP ratio.  Weights are uniform random of the real shapes (values do not affect the timing)."""
    import torch
    from oracle import config as OC, model as OM
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 0
    cfg = OC.sdxl_base_config()
    specs = OC.unet_param_specs(cfg)
    need = sum(p.numel for p in specs) * 4 * 1.15
    if avail and avail < need:
        return {"value": None, "unit": "images/sec", "cores": cores, "kind": "port",
                "sample": f"skipped: host has {avail / 1e9:.0f} GB free, fp32 SDXL-base weights need {need / 1e9:.0f} GB"}
    t00 = time.time()
    g = torch.Generator().manual_seed(0)
    W = {}
    for p in specs:   # constant fill of the real shapes: values do not affect CPU GEMM/conv timing, and this is fast
        W[p.name] = torch.full(p.shape, 0.01 if p.kind < 2 else float(p.mean) + 0.01, dtype=torch.float32)
    print(f"[cpu_baseline] weights ready in {time.time() - t00:.1f}s, {cores} threads", file=sys.stderr, flush=True)
    lat = 64
    x = torch.randn(1, 4, lat, lat, generator=g)
    ctx = torch.randn(1, 77, cfg.context_dim, generator=g)
    y = torch.randn(1, cfg.adm_in_channels, generator=g)
    with torch.no_grad():
        t0 = time.time()
        OM.unet_forward(cfg, W, x, torch.tensor([500]), ctx, y)
        t_fwd = time.time() - t0
    print(f"[cpu_baseline] UNet::forward @512^2: {t_fwd:.1f}s", file=sys.stderr, flush=True)
    tf_per_s = TFLOP_PER_UNET_FWD_512 / t_fwd
    tflop_image = 62 * TFLOP_PER_UNET_FWD_1024 + TFLOP_VAE_DECODE_1024
    return {"value": tf_per_s / tflop_image, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": (f"1 UNet::forward @512x512 (1.589 TFLOP) in {t_fwd:.1f}s on {cores} threads = {tf_per_s:.3f} TFLOP/s; "
                       f"images/sec extrapolated by FLOPs to the {tflop_image:.1f} TFLOP of one 1024x1024 31-step CFG image"),
            "tflops": tf_per_s}


# ------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed images per GPU")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--n-steps", type=int, default=30, help="--n-diffusion-steps of the reference CLI (30 -> 31 iterations)")
    ap.add_argument("--cfg", type=float, default=7.5)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32", "f16_f32res"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    rank, local_rank, world = init_dist("nccl")
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    dt = {"f16": pkg.DTYPE_F16, "f32": pkg.DTYPE_F32, "f16_f32res": pkg.DTYPE_F16_F32RES}[args.dtype]

    ctx = pkg.Context(local_rank)
    cfg = pkg.sdxl_base_config()
    # rank 0 builds the weights; replicas allocate the identical arena and receive it over RCCL / xGMI
    t0 = time.time()
    diffuser = pkg.Diffuser(ctx, cfg, dt, seed=0, empty=(rank != 0))
    decoder = pkg.LatentDecoder(ctx, None, dt, seed=0, empty=(rank != 0))
    ctx.synchronize()
    t_build = time.time() - t0
    t0 = time.time()
    if world > 1:
        broadcast_arena(diffuser.diffusion.weight_arena_tensor())
        broadcast_arena(decoder.weight_arena_tensor())
        torch.cuda.synchronize()
    t_bcast = time.time() - t0
    if args.no_graph:
        diffuser.diffusion.set_graph(False)
    diffuser.enable_step_timing(True)

    res = args.res
    lat = res // 8

    def make_prompt(seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        r = lambda *s: torch.randn(*s, device=dev, generator=g)   # noqa: E731
        cond = pkg.Conditioning(context_full=r(1, 77, cfg.context_dim), channel_context=r(1, cfg.adm_in_channels),
                                unconditional_context_full=r(77, cfg.context_dim),
                                unconditional_channel_context=r(cfg.adm_in_channels), resolution=(res, res))
        return cond, r(1, 4, lat, lat)

    prompts_ready = [make_prompt(prompt_seed(rank, s)) for s in range(-args.warmup, args.steps)]   # resident before timing
    step_ms = []
    for w in range(args.warmup):
        cond, noise = prompts_ready[w]
        decoder.latent_to_image(diffuser.sample_latent(cond, args.cfg, args.n_steps, noise))
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for s in range(args.steps):
        cond, noise = prompts_ready[args.warmup + s]
        latent = diffuser.sample_latent(cond, args.cfg, args.n_steps, noise)
        last = decoder.latent_to_image(latent)
        step_ms += diffuser.step_times_ms()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = max_over_ranks(time.perf_counter() - t0, dev)
    finite = bool(torch.isfinite(latent).all().item())
    n_images = sum_over_ranks(float(args.steps), dev)

    iters = pkg.step_count(args.n_steps)
    tflop_image = iters * 2 * TFLOP_PER_UNET_FWD_1024 * (res / 1024.0) ** 2 + TFLOP_VAE_DECODE_1024 * (res / 1024.0) ** 2
    value = n_images / elapsed

    # --- roofline of the dominant kernel, measured live with hipEvents on the timed configuration (B=2 CFG pair)
    prof = diffuser.diffusion.profile(2, lat, lat)
    ig_ms, ig_n, ig_fl = prof["igemm"]
    peak = PEAK_F32_TFLOPS if args.dtype == "f32" else PEAK_F16_TFLOPS
    achieved = ig_fl / 1e12 / (ig_ms / 1e3) if ig_ms > 0 else 0.0
    roofline = {"bound": "mfma", "kernel": "igemm_kernel (NHWC implicit-GEMM conv3x3/1x1/linear)",
                "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": None,
                "launches_per_unet_step": ig_n, "avg_launch_us": round(1e3 * ig_ms / max(ig_n, 1), 2),
                "algorithmic_tflop_per_unet_step": round(ig_fl / 1e12, 3),
                "class_ms_per_unet_step": {k: round(v[0], 3) for k, v in prof.items()},
                "whole_job_tflops": round(tflop_image * value, 1),
                "whole_job_frac_of_peak": round(tflop_image * value / (peak * max(world, 1)), 4)}

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(res)
        p50 = statistics.median(step_ms) if step_ms else None
        out = {
            "metric": "images/sec SDXL-base 1024x1024 30-step CFG7.5 (whole job); UNet step ms p50",
            "value": round(value, 4), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if args.dtype != "f32" else "f32", "data": "synthetic",
            "config": {"workload": f"SDXL-base {res}x{res}, n_steps={args.n_steps} ({iters} CFG UNet step pairs), CFG {args.cfg}, "
                                   f"batch 1 prompt/GPU + VAE decode to u8 (BASELINE configs[1])",
                       "precision": args.dtype, "weights": "synthetic seeded (random-init SDXL-base architecture)",
                       "parallelism": f"replica x{world}, 1 prompt per GPU, weights broadcast once over RCCL",
                       "hipgraph": not args.no_graph},
            "images_per_sec_per_gpu": round(value / world, 4),
            "unet_step_ms_p50": None if p50 is None else round(p50, 3),
            "tflop_per_image": round(tflop_image, 1),
            "outputs_finite": finite,
            "setup_s": {"build_weights": round(t_build, 2), "broadcast": round(t_bcast, 2)},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
