"""CPU suite: the N>1 code of bench.py itself -- shard_pairs, timed_region (barrier + max-over-ranks), gather_results -- run on
gloo, world size 2, with an injected step (the oracle on a tiny pair: the step is not what is tested, the protocol is), and
the self-launch of `python bench.py --gpus N` (command line and exit-code relay)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time

    import bench
    from tests import oracle as orc
    from tests import refpath as rp

    orc.set_num_threads(1)
    (pair,) = bench.shard_pairs(world, world, rank)              # a batch of `world` pairs: one per rank
    f0, f1, _, _ = rp.synth_pair(40, 48, C=3, seed=pair, max_flow=3)
    calls, res = [0], [None]

    def step():
        calls[0] += 1
        if res[0] is None:
            res[0] = rp.dense_flow_oracle(f0, f1, 9, 9, 7, 7)
        if rank == 1:
            time.sleep(0.004)                                   # rank 1 is the slow one: the reported time must be ITS time

    marks = []
    el = bench.timed_region(step, steps=5, warmup=2, world=world, dist=dist, device=torch.device("cpu"), sync=lambda: None,
                            spin_s=0.0, before_timed=lambda: marks.append(calls[0]))
    idx = torch.from_numpy(res[0]["idx"]).to(torch.int16)
    best = torch.from_numpy(res[0]["best"])
    gathered, g_s, g_bytes = bench.gather_results([idx, best], world, rank, dist, lambda: None)
    q.put((rank, pair, el, calls[0], marks, int(idx.sum()), None if gathered is None else [[int(t.sum()) for t in r] for r in gathered], g_bytes))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_protocol_gloo_world2():
    import bench

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = bench.free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, pair0, el0, calls0, marks0, s0, g0, b0), (r1, pair1, el1, calls1, marks1, s1, g1, b1) = got
    assert (r0, pair0, r1, pair1) == (0, 0, 1, 1)               # pair p on rank p
    assert el0 == el1 and el0 >= 5 * 0.004                      # max over ranks (rank 1 sleeps), the same number on every rank
    assert calls0 == calls1 == 7 and marks0 == marks1 == [2]    # W warm-up steps, the hook, then exactly K timed steps
    assert s0 != s1                                             # each rank processed its own pair
    assert g1 is None and g0 == [[s0, g0[0][1]], [s1, g0[1][1]]]  # rank 0 holds every rank's results, in rank order
    assert b0 == b1 > 0


def test_shard_pairs_covers_batch_once():
    import bench

    for n, w in ((8, 8), (8, 4), (5, 2), (1, 1), (3, 8)):
        shards = [bench.shard_pairs(n, w, r) for r in range(w)]
        assert sorted(p for s in shards for p in s) == list(range(n))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_bare_gpus_invocation_becomes_launcher(monkeypatch):
    """`python bench.py --gpus 4` without WORLD_SIZE must start torch.distributed.run with 4 ranks as a CHILD process (before
    anything touches the GPU) and relay its exit code."""
    import subprocess

    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    try:
        bench.main()
        raise AssertionError("main() returned instead of exiting with the launcher's code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and os.path.samefile(cmd[-7], os.path.join(ROOT, "bench.py"))
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch.cuda" not in " ".join(cmd)
