"""CPU suite: the N>1 path of bench.py (pair-per-rank sharding, barrier, max-over-ranks) on gloo, world size 2.
No GPU: the per-rank step is the oracle on a tiny pair; what is covered is the sharding / timing protocol."""
import os
import socket
import sys

import numpy as np
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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import oracle as orc
    from tests import refpath as rp

    orc.set_num_threads(1)
    f0, f1, _, _ = rp.synth_pair(40, 48, C=3, seed=rank, max_flow=3)   # one seeded pair per rank, as bench.py
    dist.barrier()
    res = rp.dense_flow_oracle(f0, f1, 9, 9, 7, 7)
    elapsed = torch.tensor([0.01 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    chk = torch.tensor([float(res["idx"].sum())], dtype=torch.float64)
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    if rank == 0:
        q.put((float(elapsed.item()), [float(g.item()) for g in gathered]))
    dist.destroy_process_group()


def test_pair_per_rank_sharding_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    tmax, sums = q.get(timeout=10)
    assert abs(tmax - 0.02) < 1e-12          # max over ranks, not rank 0's own time
    assert len(sums) == 2 and sums[0] != sums[1]   # each rank processed its own pair
