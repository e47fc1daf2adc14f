"""Two-rank RCCL test of the multi-GPU replica path (row (e) of the hot-path scope): one process per GPU, the library's own
communicator (csrc/comm.cpp: scatter + all-gather + tail, the schedule tests/test_cpu_distributed.py runs over gloo) delivers
rank 0's packed weight arena to an `empty` replica, both ranks then sample the same prompt and must agree bit for bit; and
`bench.py --gpus 2` launched WITHOUT a torchrun environment must spawn its two ranks itself and report them on the JSON line.
Skipped on a box with fewer than two GPUs (the 1-GPU boxes this suite normally runs on)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs")]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import bench
    import __graft_entry__ as ge
    from oracle import config as OC
    from util import to_pkg_cfg, seeded
    r, lr, w = bench.init_dist("nccl")
    torch.cuda.set_device(lr)
    pkg = ge.load_package()
    ctx = pkg.Context(lr)
    ocfg = OC.tiny_config()
    cfg = to_pkg_cfg(pkg, ocfg)
    comm = bench.make_comm(pkg, lr)
    d = pkg.Diffuser(ctx, cfg, pkg.DTYPE_F16, seed=0, empty=(rank != 0))      # replicas start without contents
    before = d.diffusion.weight_arena_tensor().clone()
    comm.bcast_unet(d.diffusion)
    torch.cuda.synchronize()
    arena = d.diffusion.weight_arena_tensor()
    changed = bool((arena != before).any().item())
    # every rank's arena equals rank 0's: checksum agreement over the torch group
    cs = torch.tensor([float(arena.to(torch.float64).sum().item()), float(arena[::97].to(torch.float64).sum().item())], device="cuda", dtype=torch.float64)
    all_cs = [torch.zeros_like(cs) for _ in range(world)]
    dist.all_gather(all_cs, cs)
    same_arena = all(bool(torch.equal(c, all_cs[0])) for c in all_cs)
    # a buffer broadcast through the same communicator
    buf = (torch.arange(4099, dtype=torch.float32, device="cuda") if rank == 0 else torch.zeros(4099, dtype=torch.float32, device="cuda"))
    comm.bcast_buffer(buf)
    buf_ok = bool(torch.equal(buf.cpu(), torch.arange(4099, dtype=torch.float32)))
    # identical prompt on every rank -> identical latent (independent replicas, no collective in the loop)
    B, res = 1, (64, 64)
    ctx_t = seeded(B, 77, ocfg.context_dim, seed=47).cuda()
    y = seeded(B, ocfg.adm_in_channels, seed=48).cuda()
    noise = seeded(B, 4, 8, 8, seed=46).cuda()
    cond = pkg.Conditioning(context_full=ctx_t, channel_context=y, unconditional_context_full=torch.zeros_like(ctx_t),
                            unconditional_channel_context=torch.zeros_like(y), resolution=res)
    lat = d.sample_latent(cond, 7.5, 3, noise)
    lats = [torch.zeros_like(lat) for _ in range(world)]
    dist.all_gather(lats, lat.contiguous())
    same_out = all(bool(torch.equal(t, lats[0])) for t in lats) and bool(torch.isfinite(lat).all().item())
    q.put((rank, changed, same_arena, buf_ok, same_out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_rccl_weight_broadcast_and_replicas():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert not res[0][1], "rank 0 is the source: its arena must be untouched"
    assert res[1][1], "the empty replica's arena did not change"
    assert all(r[2] for r in res), "arenas differ after the RCCL broadcast"
    assert all(r[3] for r in res), "buffer broadcast did not deliver rank 0's data"
    assert all(r[4] for r in res), "replicas disagree on the same prompt"


def test_bench_gpus_2_spawns_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "1",
                        "--no-cpu-baseline", "--no-live-parity"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak" and d["value"] > 0
