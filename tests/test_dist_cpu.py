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
    # the extra keys of a multi-rank line: a second timed region on another frame size (`scale_1080p`) and what the process group is
    leg = bench.scale_leg(step, 3, 1, world, dist, torch.device("cpu"), lambda: None, 1080, 1920, spin_s=0.0)
    info = bench.rccl_info(dist, world)
    q.put((rank, pair, el, calls[0], marks, int(idx.sum()), None if gathered is None else [[int(t.sum()) for t in r] for r in gathered], g_bytes, leg, info))
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
    (r0, pair0, el0, calls0, marks0, s0, g0, b0, leg0, info0), (r1, pair1, el1, calls1, marks1, s1, g1, b1, leg1, info1) = got
    assert (r0, pair0, r1, pair1) == (0, 0, 1, 1)               # pair p on rank p
    assert el0 == el1 and el0 >= 5 * 0.004                      # max over ranks (rank 1 sleeps), the same number on every rank
    assert calls0 == calls1 == 7 + 4 and marks0 == marks1 == [2]    # W warm-up steps, the hook, then exactly K timed steps (+ 1 + 3 of the extra leg)
    assert leg0 == leg1 and leg0["pairs_per_step"] == 2 and leg0["steps"] == 3 and leg0["ms_per_step"] >= 4.0    # the slow rank's time again
    assert abs(leg0["value"] - 2 * 1080 * 1920 / (leg0["ms_per_step"] * 1e3)) / leg0["value"] < 1e-3              # whole-job Mpixels/s
    assert info0 == info1 == {"world": 2, "backend": "gloo"}
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


def _band_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np

    import bench

    H, W, Cc = 52, 24, 3
    rng = np.random.default_rng(5)
    frames = torch.from_numpy(rng.integers(0, 256, (2, Cc, H, W)).astype(np.uint8)) if rank == 0 else torch.zeros((2, Cc, H, W), dtype=torch.uint8)
    halo = 3

    def compute(sub, a, b, r0):
        # a stand-in with the band pipeline's locality: every output row depends on the input rows within `halo` of it (a vertical box
        # sum, zero outside the FRAME) -- rows nearer than halo to an artificial band edge would be wrong, which is what the halo is for
        x = sub.to(torch.float32).sum(1)                           # [2][rows][W]
        rows = x.shape[1]
        pad = torch.zeros((2, rows + 2 * halo, W))
        pad[:, halo : halo + rows] = x
        box = sum(pad[:, i : i + rows] for i in range(2 * halo + 1))
        return [box[:, a:b].to(torch.int16), (box[0, a:b] - box[1, a:b] + r0 * 0).contiguous()]

    compute.out_shapes = lambda h: [(2, h, W), (h, W)]
    compute.out_dtypes = [torch.int16, torch.float32]
    res = {}
    for nb in (2, 3, 5):                                            # one band per rank; more bands than ranks (two rounds); an odd count
        plan = bench.band_plan(H, nb, 4, halo)
        t = []
        full = bench.run_banded_step(frames, plan, world, rank, dist, compute, torch.device("cpu"), lambda: None, t)
        res[nb] = (plan, None if full is None else [f.clone() for f in full], t)
    whole = compute(frames, 0, H, 0) if rank == 0 else None
    q.put((rank, {nb: (p, None if f is None else [x.numpy() for x in f]) for nb, (p, f, _) in res.items()}, None if whole is None else [x.numpy() for x in whole],
           frames.numpy().sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_band_split_protocol_gloo_world2():
    """ONE frame pair over the ranks (SURVEY 8(e) "single huge frame"): the uint8 frames are broadcast from rank 0, every rank computes
    the row bands it owns on the rows it needs (owned rows + halo), the owned rows are gathered and stitched on rank 0 -- and the
    result equals the whole frame computed in one piece, for one band per rank, for more bands than ranks and for ragged bands."""
    import numpy as np

    import bench

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = bench.free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, res0, whole, sum0), (r1, res1, whole1, sum1) = got
    assert (r0, r1) == (0, 1) and whole1 is None and sum0 == sum1       # the broadcast delivered rank 0's frames
    for nb, (plan, full) in res0.items():
        assert len(plan) == nb and plan[0][0] == 0 and plan[-1][1] == 52
        for (o0, o1, a, b), nxt in zip(plan, plan[1:] + [None]):
            assert o0 % 4 == 0 and a == max(0, o0 - 3) and b == min(52, o1 + 3) and (nxt is None or nxt[0] == o1)
        assert full is not None and res1[nb][1] is None                 # only rank 0 holds the stitched frame
        for f, w in zip(full, whole):
            assert f.shape == w.shape and f.dtype == w.dtype and np.array_equal(f, w)


def test_band_plan_properties():
    import bench

    for H, n, align, halo in [(2160, 8, 1, 19), (2160, 8, 16, 128), (1080, 3, 8, 64), (40, 8, 16, 16), (17, 4, 1, 2)]:
        plan = bench.band_plan(H, n, align, halo)
        assert 1 <= len(plan) <= n and plan[0][0] == 0 and plan[-1][1] == H
        for (o0, o1, r0, r1), nxt in zip(plan, plan[1:] + [None]):
            assert o0 < o1 and o0 % align == 0 and r0 == max(0, o0 - halo) and r1 == min(H, o1 + halo)
            assert nxt is None or nxt[0] == o1
        sizes = [o1 - o0 for o0, o1, _, _ in plan]
        assert max(sizes) - min(sizes) <= 2 * align
