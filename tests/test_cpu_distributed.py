"""world_size-2 gloo test (CPU) of the multi-GPU orchestration: the library's weight-broadcast schedule (sdxl_bcast_plan:
scatter + all-gather + tail, csrc/comm.cpp) executed over gloo on a real parameter byte image, the torch.distributed fallback
broadcast, independent prompt sharding (no collective in the loop), max-over-ranks timing and whole-job aggregation."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import bench
    r, lr, w = bench.init_dist("gloo")
    assert (r, w) == (rank, world)
    # rank 0 owns the weights, replicas start with garbage ("empty" arena) and receive them in chunks
    n = 3_000_001
    arena = (torch.arange(n, dtype=torch.int64) % 251).to(torch.uint8) if rank == 0 else torch.full((n,), 7, dtype=torch.uint8)
    bench.broadcast_arena(arena, src=0, chunk_bytes=1 << 20)
    ok = bool(torch.equal(arena, (torch.arange(n, dtype=torch.int64) % 251).to(torch.uint8)))
    # the LIBRARY's schedule (sdxl_bcast_plan: scatter + all-gather + tail) on a real weight-arena byte image: the flat fp32
    # parameters of the tiny UNet in sdxl_unet_param_spec order (what rank 0 holds; 48 odd bytes appended -> a ragged tail)
    import numpy as np
    import __graft_entry__ as ge
    from oracle import config as OC
    pkg = ge.load_package()
    W = OC.synth_weights(OC.unet_param_specs(OC.tiny_config()), 5)
    image = np.concatenate([W[p.name].reshape(-1) for p in OC.unet_param_specs(OC.tiny_config())]).view(np.uint8)
    image = torch.from_numpy(np.concatenate([image, np.arange(48, dtype=np.uint8)]))
    arena2 = image.clone() if rank == 0 else torch.full_like(image, 0xAB)
    bench.scatter_allgather_arena(arena2, pkg.bcast_plan, src=0)
    ok = ok and bool(torch.equal(arena2, image))
    plan = pkg.bcast_plan(image.numel(), world, rank)
    ok = ok and plan[1] % 256 == 0 and plan[0] == rank * plan[1] and plan[2] == world * plan[1] and plan[2] + plan[3] == image.numel()
    bench.barrier()
    t = bench.max_over_ranks(1.0 + rank, torch.device("cpu"))
    total = bench.sum_over_ranks(3.0, torch.device("cpu"))
    seeds = [bench.prompt_seed(rank, s) for s in range(4)]
    q.put((rank, ok, t, total, seeds))
    dist.destroy_process_group()


def test_two_rank_gloo_orchestration():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "broadcast did not deliver rank 0's arena"
    assert all(abs(r[2] - 2.0) < 1e-9 for r in res), "timing must be the max over ranks"
    assert all(abs(r[3] - 6.0) < 1e-9 for r in res), "value aggregates the images of all ranks"
    all_seeds = [s for r in res for s in r[4]]
    assert len(set(all_seeds)) == len(all_seeds), "every (rank, step) must be an independent prompt"


def test_single_process_helpers_are_noops():
    sys.path.insert(0, ROOT)
    import bench
    a = torch.ones(10, dtype=torch.uint8)
    bench.broadcast_arena(a)
    assert bench.max_over_ranks(1.5, torch.device("cpu")) == 1.5 and bench.sum_over_ranks(2.0, torch.device("cpu")) == 2.0
