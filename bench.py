#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the dense flow+depth hot path on synthetic frame pairs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload vga|720p|1080p]

One *step* = one pass of the hot path over one frame pair per GPU, frames already resident in HBM:
dfe_flow_depth_pair_f32 = SSD cost-volume build (materialised once, reference layout) with the per-chunk
arg-min fused into its epilogue -> finalize (centre tie-break / extractOutput / decode) -> border + flow->depth.  N > 1 is launched by
torch.distributed.run, one rank per GPU; every rank processes its own pair (pairs are independent:
weak scaling, no data-path collective; RCCL is used for the barrier and the max-over-ranks only).

The JSON line carries `roofline` for the dominant kernel (the cost-volume build: algorithmic bytes
B_alg = 2*C*H*W*4 + Ho*Wo*hWin*wWin*4 per launch over its HIP-event time, measured live on the
kernel's stream inside the timed region) and `cpu_baseline` (the CPU oracle, i.e. a port of the
reference's loop nest, timed on this host's cores on a bounded band of output rows, rank 0 / N=1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (H, W, C, k, hWin, wWin)  -- 7x7 patch, +-16 search (33x33 window), fp32
    "vga": (480, 640, 3, 7, 33, 33),
    "720p": (720, 1280, 3, 7, 33, 33),
    "1080p": (1080, 1920, 3, 7, 33, 33),
    "vga-luma": (480, 640, 1, 7, 33, 33),   # luminance frames (C = 1)
}
# fp16 cost volume (BASELINE configs[4]; SURVEY 8(d): 3840x2160 / 33x33 fp16 = 17 770.8 MB per pair): the same step through
# dfe_flow_depth_pair_f16 -- volume stored as half(cost * 2^-8), arg-min before the down-convert
F16_WORKLOADS = {
    "4k-f16": (2160, 3840, 3, 7, 33, 33),
    "vga-f16": (480, 640, 3, 7, 33, 33),
    "1080p-f16": (1080, 1920, 3, 7, 33, 33),
}
# BASELINE.json configs[1] literally: 640x480, 3-level pyramid {1,2,4}, 7x7 patch, 8x8 window per scale (= +-16 at the
# coarsest scale), through dfe_multiscale_flow_pair_f32.  Not the default: the north-star roofline target is stated for
# the single-scale cost-volume build, which is what `vga` measures.
PYRAMIDS = {
    "vga-pyramid": (480, 640, 3, 7, 8, 8, (1, 2, 4)),
    "720p-pyramid": (720, 1280, 3, 7, 8, 8, (1, 2, 4, 8)),
    "1080p-pyramid": (1080, 1920, 3, 7, 8, 8, (1, 2, 4, 8)),
    # BASELINE.json configs[4]: 3840x2160, 5-level pyramid, fp16 cost volumes (dfe_multiscale_flow_pair_f16, scale 1: frames / 64);
    # the *-f16 variants of the smaller frames are there to compare with their fp32 lines
    "4k-pyramid-f16": (2160, 3840, 3, 7, 8, 8, (1, 2, 4, 8, 16)),
    "4k-pyramid": (2160, 3840, 3, 7, 8, 8, (1, 2, 4, 8, 16)),
    "1080p-pyramid-f16": (1080, 1920, 3, 7, 8, 8, (1, 2, 4, 8)),
    "vga-pyramid-f16": (480, 640, 3, 7, 8, 8, (1, 2, 4)),
    # the same matcher with LEARNED patch filters in front of every scale (getModelMultiscale + getFilter, shared parameters):
    # layers {3,5,5,4},{4,5,5,4},{4,5,5,10} of tests/time_matching.lua:13 (13 x 13 receptive field, K = 10 features), random-init
    "vga-pyramid-learned": (480, 640, 3, 13, 8, 8, (1, 2, 4)),
    "1080p-pyramid-learned": (1080, 1920, 3, 13, 8, 8, (1, 2, 4, 8)),
}
LEARNED_LAYERS = [(3, 5, 5, 4), (4, 5, 5, 4), (4, 5, 5, 10)]
# BASELINE.json configs[2]: 1280x720 radial (polar-warped) flow: C2P warp of both frames around the epipole, the default
# separable filter stack {{3,1,17,5},{5,17,1,10}}, SpatialRadialMatching(15), arg-min, P2C warp, flow2depth, through
# dfe_radial_flow_depth_pair_f32; polar image = frame size ("bench at 720x1280 polar").
RADIALS = {
    # name: (hImg, wImg, C, hInput, wInput, hWin, layers)
    "720p-radial": (720, 1280, 3, 720, 1280, 15, [[3, 1, 17, 5], [5, 17, 1, 10]]),
    "vga-radial": (480, 640, 3, 480, 640, 15, [[3, 1, 17, 5], [5, 17, 1, 10]]),
}
# The trained single-scale models (verdict r3 items 3 / 4):
#   version2-*     version2/network.lua:5-39 + version2/test.lua:40-53 through dfe_version2_flow_pair_f32 (ONE call: contrastive
#                  normalisation k = 17, 17 x 17 x 32 convolution, SpatialMatching(17, 17), first-min decode), random-init weights
#   time-matching  tests/time_matching.lua:5-47 as written: getFilter({3,5,5,4},{4,5,5,4},{4,5,5,10}) on two randn 3 x 180 x 320 frames,
#                  prepareInput's narrow, nn.SpatialMatching(16, 16), and the script's `output:min(1)` of the Reshape(256, 293, 153) view
LEARNED_SINGLE = {
    # name: (H, W, normalization_k, layers {nIn, kW, kH, nOut}, window)
    "version2-vga": (480, 640, 17, [(3, 17, 17, 32)], 17),
    "version2-180p": (180, 320, 17, [(3, 17, 17, 32)], 17),      # version2/test.lua's own datap: 320 x 180
    # the same model with both opt-in matrix-core forms (dfe_set_option conv_mfma = 1, fm_mfma = 1): fused multiply-adds in the convolution,
    # |a|^2 + |b|^2 - 2 a.b in the matcher -- results within the tolerances of include/dfe.h, not bit-identical to the exact line above
    "version2-vga-mfma": (480, 640, 17, [(3, 17, 17, 32)], 17),
    "time-matching": (180, 320, 0, [(3, 5, 5, 4), (4, 5, 5, 4), (4, 5, 5, 10)], 16),
    # getModel + processOutput as depth_estimation_opticalflow.lua:103-116 runs them for a single-scale model, in ONE call
    # (dfe_flow_pair_filtered_f32): the stack of time_matching.lua:13 on both frames, prepareInput's narrow, SpatialMatching(16, 16),
    # Minus / SoftMax, arg-max with the centre tie-break (nk = -1) or extractOutput + threshold 0.11 (nk = -2), decode, centre paste
    "vga-learned": (480, 640, -1, [(3, 5, 5, 4), (4, 5, 5, 4), (4, 5, 5, 10)], 16),
    "vga-learned-thr": (480, 640, -2, [(3, 5, 5, 4), (4, 5, 5, 4), (4, 5, 5, 10)], 16),
    "720p-learned": (720, 1280, -1, [(3, 5, 5, 4), (4, 5, 5, 4), (4, 5, 5, 10)], 16),
}
VALU_RATE_PER_CU = 1.6   # wave-instructions per cycle and CU for plain fp32 register ops (tools/ubench/valu2.hip), 256 CUs at 2.4 GHz
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(H, W, Cc, k, hWin, wWin, elem=4):
    Ho, Wo = H - k + 1 - hWin + 1, W - k + 1 - wWin + 1
    return 2 * Cc * H * W * 4 + Ho * Wo * hWin * wWin * elem


# ------------------------------------------------------------------------------------------------------------------
# multi-GPU protocol: pairs are independent -> pair p runs on rank p mod world, no data-path collective;
# the only collectives are the barrier / max-over-ranks of the timing and the optional gather of the results to rank 0.
# These functions take the process group, the device and the synchronise callable from the caller, so that
# tests/test_dist_cpu.py drives exactly this code on gloo / CPU with an injected step.
# ------------------------------------------------------------------------------------------------------------------
def shard_pairs(n_pairs, world, rank):
    """Pair ids this rank processes: p -> rank p mod world (north_star: one pair per GPU when a batch splits naturally)."""
    return [p for p in range(n_pairs) if p % world == rank]


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def timed_region(step, steps, warmup, world, dist, device, sync, spin_s=0.08, before_timed=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + sync on both sides; returns the MAX over ranks of
    the elapsed seconds (every rank gets it).  `spin_s` of untimed steps first: the GPU needs ~50 ms of work to reach its
    sustained clocks (after 20 steps alone the same run reads 2.5 % low)."""
    import torch

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < spin_s:
        for _ in range(20):
            step()
        sync()
    for _ in range(warmup):
        step()
    barrier()
    if before_timed:
        before_timed()
        barrier()
    t_start = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    el = torch.tensor([time.perf_counter() - t_start], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item())


def scale_leg(step, steps, warmup, world, dist, device, sync, H, W, spin_s=0.04):
    """A second timed region of the same protocol on another frame size, run only when world > 1: the >= 6x-at-8-GPUs target of the
    north star is stated for batched 1080p pairs, the driver's scaling run uses the default (VGA) workload -- so the VGA line carries
    the 1080p number of the same launch as an extra key (`scale_1080p`).  Returns {value, ms_per_step, pairs_per_step}."""
    el = timed_region(step, steps, warmup, world, dist, device, sync, spin_s=spin_s)
    return {"value": round(world * steps * H * W / el / 1e6, 3), "unit": "Mpixels/s", "ms_per_step": round(el / steps * 1e3, 4), "pairs_per_step": world,
            "steps": steps, "warmup": warmup}


def rccl_info(dist, world):
    """What the live process group is: {"world", "backend"} (backend "nccl" IS RCCL on ROCm; gloo in the CPU tests)."""
    if world <= 1 or not dist.is_initialized():
        return {"world": 1, "backend": None}
    return {"world": dist.get_world_size(), "backend": dist.get_backend()}


def gather_results(tensors, world, rank, dist, sync):
    """Optional result gather: every rank's per-pair outputs -- flow as int16 x 2, confidence and depth fp32 --
    to rank 0 over the process group (RCCL gather over xGMI on the GPU box), timed on its own, outside the timed region.
    Returns (list of per-rank tensor lists on rank 0 / None elsewhere, seconds, bytes received by rank 0)."""
    import torch

    if world == 1:
        return [tensors], 0.0, 0
    dist.barrier()
    sync()
    t = time.perf_counter()
    out, nbytes = [], 0
    for x in tensors:
        wire = x.contiguous().view(torch.uint8) if x.dtype == torch.int16 else x   # (neither RCCL nor gloo has an int16 type: bytes)
        dst = [torch.empty_like(wire) for _ in range(world)] if rank == 0 else None
        dist.gather(wire, dst, dst=0)
        out.append([t.view(x.dtype) for t in dst] if rank == 0 else None)
        nbytes += x.numel() * x.element_size() * (world - 1)
    sync()
    dist.barrier()
    secs = time.perf_counter() - t
    if rank != 0:
        return None, secs, nbytes
    return [[out[i][r] for i in range(len(tensors))] for r in range(world)], secs, nbytes


# ------------------------------------------------------------------------------------------------------------------
# ONE frame pair over N GPUs (SURVEY 8(e) "single huge frame", BASELINE configs[4]: a 4K pair on 8 GPUs): row bands of the
# output with a read halo.  Rank 0 holds the uint8 frames and broadcasts them (RCCL over xGMI: 2 x 24.9 MB at 4K); every rank
# computes the bands it owns on the rows it needs; the owned rows are gathered to rank 0.  Strong scaling: the work of a step is
# fixed, the ranks split it.
# ------------------------------------------------------------------------------------------------------------------
def band_plan(H, n_bands, align, halo):
    """Row bands of a frame of H rows: band b OWNS frame rows [o0, o1) (o0 a multiple of `align`; equal shares up to `align`) and
    READS rows [r0, r1) = the owned rows widened by `halo` and clipped to the frame.  -> list of (o0, o1, r0, r1); bands may be fewer
    than n_bands when the frame is short."""
    units = -(-H // align)
    n = max(1, min(n_bands, units))
    out = []
    for b in range(n):
        o0, o1 = (b * units // n) * align, min(H, ((b + 1) * units // n) * align)
        out.append((o0, o1, max(0, o0 - halo), min(H, o1 + halo)))
    return out


def band_owner(b, world):
    return b % world


def run_banded_step(frames_u8, plan, world, rank, dist, compute_band, device, sync, timing=None):
    """One step of the band-split pipeline.  frames_u8: on rank 0 the pair as ONE uint8 tensor [2][C][H][W], a same-shaped buffer on
    the other ranks.  compute_band(sub_u8 [2][C][rows][W], o0 - r0, o1 - r0, r0) -> list of tensors whose FIRST dimension after
    any leading plane dimension is the owned rows (each [.., o1-o0, W]).  Returns on rank 0 the stitched full-frame tensors
    (list), None elsewhere.  Collectives: one broadcast of the frames, one gather per output tensor and band round."""
    import torch

    t0 = time.perf_counter()
    if world > 1:
        dist.broadcast(frames_u8, src=0)
        if timing is not None:
            sync()          # (the collective is asynchronous to the host: without this its time would be charged to `compute`)
    t1 = time.perf_counter()
    mine = {}
    for b, (o0, o1, r0, r1) in enumerate(plan):
        if band_owner(b, world) != rank:
            continue
        mine[b] = compute_band(frames_u8[:, :, r0:r1], o0 - r0, o1 - r0, r0)
    sync()
    t2 = time.perf_counter()
    full = None
    H = plan[-1][1]
    nrounds = -(-len(plan) // world)
    hmax = max(o1 - o0 for o0, o1, _, _ in plan)
    for rnd in range(nrounds):
        b = rnd * world + rank
        outs = mine.get(b)
        if world == 1:
            parts = [(b, outs)] if outs is not None else []
        else:
            # every rank sends tensors of ONE shape (its band padded to the tallest band): gather needs equal sizes
            like = next(iter(mine.values())) if mine else None
            shapes = compute_band.out_shapes(hmax) if like is None else [tuple(t.shape[:-2]) + (hmax, t.shape[-1]) for t in like]
            dtypes = compute_band.out_dtypes if like is None else [t.dtype for t in like]
            parts = []
            got = []
            for i, (shp, dt) in enumerate(zip(shapes, dtypes)):
                send = torch.zeros(shp, dtype=dt, device=device)
                if outs is not None:
                    send[..., : outs[i].shape[-2], :] = outs[i]
                wire = send.view(torch.uint8) if dt == torch.int16 else send
                dst = [torch.empty_like(wire) for _ in range(world)] if rank == 0 else None
                dist.gather(wire, dst, dst=0)
                got.append([t.view(dt) for t in dst] if rank == 0 else None)
            if rank == 0:
                for r in range(world):
                    bb = rnd * world + r
                    if bb < len(plan):
                        parts.append((bb, [got[i][r] for i in range(len(shapes))]))
        if rank == 0:
            for bb, ts in parts:
                o0, o1 = plan[bb][0], plan[bb][1]
                if full is None:
                    full = [torch.empty(tuple(t.shape[:-2]) + (H, t.shape[-1]), dtype=t.dtype, device=t.device) for t in ts]
                for f, t in zip(full, ts):
                    f[..., o0:o1, :] = t[..., : o1 - o0, :]
    sync()
    if timing is not None:
        timing.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
    return full


def launch_ranks(gpus, argv):
    """`python bench.py --gpus N` from a bare shell: start N fresh rank processes under torch.distributed.run (one per GPU,
    RCCL) and relay their output and exit code.  Called before this process touches torch.cuda or libdfe -- the ranks are
    children, never a re-exec of a process that has initialised the GPU."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    # the pool's host driver supports dmabuf IPC only: without this RCCL's (and torch's) cross-process buffer sharing fails with
    # `hipIpcGetMemHandle: invalid argument` at init_process_group.  Already exported on the driver's boxes; kept for a bare shell.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(f0, f1, k, hWin, wWin, cx, cy, budget_s=12.0):
    """The oracle (a port of the reference's CPU loop nest: 5-deep SSD loop with OpenMP over output
    rows + min/tie-break + extractOutput + decode + depth), timed on a band of output rows sized
    for ~budget_s of work.  Mpix/s is scaled by the band's share of the pair."""
    import numpy as np
    from tests import oracle as orc
    from tests import refpath as rp

    Cc, H, W = f0.shape
    Ho, Wo = H - k + 1 - hWin + 1, W - k + 1 - wWin + 1
    halo = k - 1 + hWin - 1

    def run(rows):
        r0 = (Ho - rows) // 2
        t = time.perf_counter()
        band = orc.ssd_cost_volume(f0[:, r0 : r0 + rows + halo], f1[:, r0 : r0 + rows + halo], k, k, hWin, wWin)
        vol = band.reshape(rows, Wo, hWin * wWin)
        idx, _ = orc.argbest_center(vol, rp.middle_index(hWin, wWin), False)
        y, x = orc.x2yx(idx, hWin, wWin)
        sc, im = np.zeros((rows, Wo), np.float32), np.zeros((rows, Wo), np.int64)
        orc.extract_output(vol, 0.21, im, sc)
        flow = np.stack([y, x]).astype(np.float32)
        orc.flow_to_depth_cartesian(flow, cx, cy)
        return time.perf_counter() - t

    ncores = orc.max_threads()
    run(1)  # start the OpenMP team
    t8 = run(8)
    rows = int(max(8, min(Ho, round(8 * budget_s / max(t8, 1e-6)))))
    t = run(rows)
    mpix = (H * W * rows / Ho) / t / 1e6
    out = {
        "value": round(mpix, 4),
        "unit": "Mpixels/s",
        "cores": ncores,
        "kind": "port",
        "sample": "%d of %d output rows of one %dx%d pair (cost volume + arg-min/extract/decode/depth), %.1f s" % (rows, Ho, W, H, t),
    }
    # the reference's own default is 2 OpenMP threads (depth_estimation_opticalflow.lua:16,40): same port, 2 threads
    orc.set_num_threads(2)
    try:
        t2 = run(2)
        rows2 = int(max(2, min(Ho, round(2 * 5.0 / max(t2, 1e-6)))))
        t2 = run(rows2)
        out["value_2_threads"] = round((H * W * rows2 / Ho) / t2 / 1e6, 5)
        out["sample_2_threads"] = "%d rows, %.1f s" % (rows2, t2)
    finally:
        orc.set_num_threads(ncores)
    return out


def main_pyramid(args, world, rank, local_rank, dev, torch, dist, d, rp):
    """Multiscale workload: step = dfe_multiscale_flow_pair_f32 on one pair per GPU (per scale: down-sample + pad, raw-patch
    SSD volume, softmin; then the fused cascade / ring / arg-max / decode).  Several small kernels, so `roofline` prices
    the WHOLE step against the algorithmic bytes of SURVEY 8(d) (frames read once + every scale's native volume once)."""
    H, W, Cc, k, maxh, maxw, ratios = PYRAMIDS[args.workload]
    if os.environ.get("DFE_GRAPHS") == "1":
        # libdfe enqueues on torch's current stream; on a real stream (not the legacy default one) it can replay this
        # launch-bound step -- 4 or 5 kernels of 9-33 us -- as a hipGraph (measured slower than the direct launches: 0.0857
        # against 0.0803 ms at VGA, so off by default)
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    rmax = ratios[-1]
    Hp, Wp = -(-H // rmax) * rmax, -(-W // rmax) * rmax            # the reference pads to a multiple of the coarsest ratio
    f0, f1, _, _ = rp.synth_pair(H, W, C=Cc, seed=rank, max_flow=12)
    import numpy as np
    p0, p1 = np.zeros((Cc, Hp, Wp), np.float32), np.zeros((Cc, Hp, Wp), np.float32)
    p0[:, :H, :W], p1[:, :H, :W] = f0 / 64.0, f1 / 64.0
    t0, t1 = torch.from_numpy(p0).to(dev), torch.from_numpy(p1).to(dev)
    flow = torch.empty((2, Hp, Wp), device=dev)
    ctx = d.get_ctx(local_rank)
    lib = d.lib()
    rr = (C.c_int32 * len(ratios))(*ratios)

    f16 = args.workload.endswith("-f16")
    learned = args.workload.endswith("-learned")
    if learned:
        from depth_estimation_amd.multiscale import filter_layers_array

        filt = d.getFilter(dict(layers=LEARNED_LAYERS), device=dev, generator=torch.Generator().manual_seed(0))   # random-init weights of the architecture
        larr, nl, _keep = filter_layers_array([filt])

    def step():
        if learned:
            ctx.check(lib.dfe_multiscale_flow_pair_filtered_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, Hp, Wp, maxh, maxw, rr, len(ratios), larr, nl, 1, 0.0,
                                                               flow.data_ptr(), None))
            return
        if f16:
            ctx.check(lib.dfe_multiscale_flow_pair_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, Hp, Wp, k, maxh, maxw, rr, len(ratios), 1.0,
                                                      flow.data_ptr(), None))
            return
        ctx.check(lib.dfe_multiscale_flow_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, Hp, Wp, k, maxh, maxw, rr, len(ratios),
                                                  flow.data_ptr(), None))

    elapsed = timed_region(step, args.steps, args.warmup, world, dist, dev, torch.cuda.synchronize)
    last_kernel = ctx.last_kernel()
    if rank == 0:
        balg = 2 * Cc * Hp * Wp * 4 + sum((Hp // r) * (Wp // r) * maxh * maxw * (2 if f16 else 4) for r in ratios)
        step_s = elapsed / args.steps
        print(json.dumps({
            "metric": "Mpixels/s dense flow, %dx%d pair, %d-level pyramid, %s, %dx%d window per scale" % (W, H, len(ratios), "13x13 learned filters" if learned else "7x7 patch", maxh, maxw),
            "value": round(world * args.steps * H * W / elapsed / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 sums, f16 volumes" if f16 else "f32", "data": "synthetic",
            "config": {"workload": "%dx%d C=%d multiscale matcher ratios %s (per scale: box down-sample, zero-pad, %s over %dx%d%s, "
                                   "softmin) + cascade / ring / arg-max / decode, one pair per GPU per step" % (
                                       W, H, Cc, list(ratios), ("learned filter stack %s (shared, random-init) + SpatialMatching" % LEARNED_LAYERS) if learned else "7x7 raw-patch SSD",
                                       maxh, maxw, " stored as fp16" if f16 else ""),
                       "pairs_per_step": world, "sharding": "pair-per-gpu" if world > 1 else "single"},
            # (`achieved` prices the step against the ALGORITHMIC bytes -- every scale's volume once; where the library consumes the finest
            #  scales inside the volume kernel (last kernel *_fine_kernel: frames from 720p up, learned filters from VGA up) those bytes
            #  never reach HBM and the step is VALU-issue-bound)
            "roofline": {"bound": "hbm", "kernel": "whole step (prep, volumes, one cascade launch per scale; last kernel: %s)" % last_kernel, "achieved": round(balg / step_s / 1e9, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(balg / step_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": balg},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


BAND_WORKLOADS = {
    # one pair split into row bands over the ranks (BASELINE configs[4] on N GPUs; SURVEY 8(e) "single huge frame")
    # name: (kind, H, W, C, k, window, ratios)
    "4k-f16-bands": ("single", 2160, 3840, 3, 7, 33, None),
    "4k-pyramid-f16-bands": ("pyramid", 2160, 3840, 3, 7, 8, (1, 2, 4, 8, 16)),
    "1080p-f16-bands": ("single", 1080, 1920, 3, 7, 33, None),
}


def make_band_compute(kind, d, ctx, dev, Cc, W, k, win, ratios, cx, cy):
    """compute_band for run_banded_step: the fp16-volume pipelines on a row band of the pair (uint8 rows in, owned rows out)."""
    import torch

    lib = d.lib()
    if kind == "single":
        def compute(sub_u8, a, b, r0):
            f = sub_u8.to(torch.float32)                                   # [2][C][rows][W], contiguous
            rows = f.shape[2]
            flow = torch.empty((2, rows, W), device=dev)
            depth, conf = torch.empty((rows, W), device=dev), torch.empty((rows, W), device=dev)
            ctx.check(lib.dfe_flow_depth_pair_f16(ctx.handle, f[0].data_ptr(), f[1].data_ptr(), Cc, rows, W, k, win, win, cx, cy - r0, 2.0 ** -8, None, None,
                                                  flow.data_ptr(), depth.data_ptr(), conf.data_ptr()))
            return [flow[:, a:b].to(torch.int16), depth[a:b].contiguous()]
        compute.out_shapes = lambda h: [(2, h, W), (h, W)]
        compute.out_dtypes = [torch.int16, torch.float32]
        compute.halo, compute.align = (win - 1) // 2 + (k - 1) // 2, 1     # the centre-paste offset: rows nearer than this to the band edge are border
    else:
        rr = (C.c_int32 * len(ratios))(*ratios)

        def compute(sub_u8, a, b, r0):
            f = sub_u8.to(torch.float32) * (1.0 / 64.0)
            rows = f.shape[2]
            flow = torch.empty((2, rows, W), device=dev)
            ctx.check(lib.dfe_multiscale_flow_pair_f16(ctx.handle, f[0].data_ptr(), f[1].data_ptr(), Cc, rows, W, k, win, win, rr, len(ratios), 1.0,
                                                       flow.data_ptr(), None))
            return [flow[:, a:b].to(torch.int16)]
        compute.out_shapes = lambda h: [(2, h, W)]
        compute.out_dtypes = [torch.int16]
        # scale r reads down-sampled rows ys - floor(hp/2) .. ys + ceil(hp/2) with hp = (win-1)+(k-1): at the coarsest ratio that is
        # ceil(hp/2) = 7 coarse rows below the last owned one for the 8 x 8 / 7 x 7 geometry (6 above the first); bands and halo are
        # multiples of rmax so that every scale's parent pixels stay aligned
        hp = win - 1 + k - 1
        compute.halo, compute.align = ratios[-1] * (hp - hp // 2), ratios[-1]
    return compute


def main_bands(args, world, rank, local_rank, dev, torch, dist, d, rp):
    """One pair, row bands over the ranks: step = broadcast of the uint8 frames from rank 0 + every rank's band(s) + gather of the
    owned rows to rank 0.  `--bands B` forces at least B bands (band b on rank b mod world), which is how the stitching is
    exercised on one GPU."""
    import numpy as np

    kind, H, W, Cc, k, win, ratios = BAND_WORKLOADS[args.workload]
    f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=Cc, seed=0, max_flow=12)
    frames = torch.from_numpy(np.stack([f0, f1]).astype(np.uint8)).to(dev) if rank == 0 else torch.empty((2, Cc, H, W), dtype=torch.uint8, device=dev)
    ctx = d.get_ctx(local_rank)
    compute = make_band_compute(kind, d, ctx, dev, Cc, W, k, win, ratios, cx, cy)
    plan = band_plan(H, max(world, args.bands), compute.align, compute.halo)
    timing = []
    out = [None]

    def step():
        out[0] = run_banded_step(frames, plan, world, rank, dist, compute, dev, torch.cuda.synchronize, timing)

    elapsed = timed_region(step, args.steps, args.warmup, world, dist, dev, torch.cuda.synchronize, spin_s=0.0)
    if rank == 0:
        tm = np.array(timing[-args.steps:])
        step_s = elapsed / args.steps
        vol = (H - k - win + 2) * (W - k - win + 2) * win * win * 2 if kind == "single" else sum((H // r) * (W // r) * win * win * 2 for r in ratios)
        balg = 2 * Cc * H * W * 4 + vol
        print(json.dumps({
            "metric": "Mpixels/s dense flow%s, ONE %dx%d pair split into row bands over the GPUs" % ("+depth" if kind == "single" else "", W, H),
            "value": round(args.steps * H * W / elapsed / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 sums, f16 volume%s" % ("" if kind == "single" else "s"), "data": "synthetic",
            "config": {"workload": "%dx%d C=%d %s, fp16 volume(s); uint8 frames broadcast from rank 0, %d row bands (halo %d rows), owned rows gathered to rank 0"
                                   % (W, H, Cc, "single-scale 33x33" if kind == "single" else "pyramid %s 8x8" % list(ratios), len(plan), compute.halo),
                       "pairs_per_step": 1, "sharding": "row-bands", "bands": len(plan),
                       # rows computed per row owned, summed over the bands: what the halo recompute costs the strong-scaling efficiency
                       "halo_overhead": round(sum(r1 - r0 for _, _, r0, r1 in plan) / float(sum(o1 - o0 for o0, o1, _, _ in plan)), 4)},
            "roofline": {"bound": "hbm", "kernel": "whole step (broadcast, band pipelines, gather)", "achieved": round(balg / step_s / 1e9, 2),
                         "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(balg / step_s / 1e9 / (HBM_PEAK_GBS * world), 4), "traffic": None,
                         "algorithmic_bytes_per_launch": balg},
            "step_breakdown_ms": {"broadcast": round(float(tm[:, 0].mean()) * 1e3, 4), "compute": round(float(tm[:, 1].mean()) * 1e3, 4),
                                  "gather_and_stitch": round(float(tm[:, 2].mean()) * 1e3, 4)},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_radial(args, world, rank, local_rank, dev, torch, dist, d, rp):
    """Radial workload: step = dfe_radial_flow_depth_pair_f32 on one pair per GPU.  `roofline` is the matcher + arg-min kernel
    (A1r) against its algorithmic bytes (both feature maps read once + the hWin-cell volume written once), timed
    with HIP events inside the timed region; the step also runs the polar warps, the two filter launches per frame and the
    P2C / depth pass."""
    hImg, wImg, Cc, hIn, wIn, hWin, layers = RADIALS[args.workload]
    networkp = dict(hImg=hImg, wImg=wImg, hInput=hIn, wInput=wIn, hWin=hWin, layers=layers)
    (pair_id,) = shard_pairs(world, world, rank)
    f0, f1, _, (cx, cy) = rp.synth_pair(hImg, wImg, C=Cc, seed=pair_id, max_flow=12)
    t0, t1 = torch.from_numpy(f0 / 255.0).to(dev, torch.float32), torch.from_numpy(f1 / 255.0).to(dev, torch.float32)
    net = d.getTesterNetwork(networkp, device=dev, generator=torch.Generator().manual_seed(0))   # random-init weights of the architecture
    ctx = d.get_ctx(local_rank)
    lib = d.lib()
    from depth_estimation_amd._lib import RadialParams
    from depth_estimation_amd.radial import _separable_weights

    w1, b1, w2, b2, th = _separable_weights(net, networkp)
    prm = RadialParams(Cc, hImg, wImg, hIn, wIn, hWin, w1.shape[0], w1.shape[3], w2.shape[0], w2.shape[2], int(th), 1.0, 0.65)
    hm, hOut, wOut = d.radial_out_shape(networkp)
    vol = torch.empty((hm, wIn, hWin), device=dev)
    pf = torch.empty((hm, wIn), device=dev)
    cart, depth, conf = (torch.empty((hOut, wOut), device=dev) for _ in range(3))

    def step():
        ctx.check(lib.dfe_radial_flow_depth_pair_f32(ctx.handle, C.byref(prm), t0.data_ptr(), t1.data_ptr(), cx, cy, w1.data_ptr(), b1.data_ptr(),
                                                     w2.data_ptr(), b2.data_ptr(), vol.data_ptr(), pf.data_ptr(), cart.data_ptr(), depth.data_ptr(),
                                                     conf.data_ptr()))

    elapsed = timed_region(step, args.steps, args.warmup, world, dist, dev, torch.cuda.synchronize,
                           before_timed=lambda: ctx.check(lib.dfe_profile_enable(ctx.handle, 1)))
    ms, n = C.c_double(), C.c_int()
    ctx.check(lib.dfe_profile_read(ctx.handle, C.byref(ms), C.byref(n)))
    ctx.check(lib.dfe_profile_enable(ctx.handle, 0))
    if rank == 0:
        K = w2.shape[0]
        hf2 = hIn - (w2.shape[2] - 1)
        balg = K * hm * wIn * 4 + K * hf2 * wIn * 4 + hm * wIn * hWin * 4        # in1 (the rows the matcher reads) + in2 + volume
        kern_s = ms.value / 1e3 / max(n.value, 1)
        step_s = elapsed / args.steps
        print(json.dumps({
            "metric": "Mpixels/s dense radial flow+depth, %dx%d pair, polar %dx%d, 17x17 separable filter, radial window %d" % (wImg, hImg, wIn, hIn, hWin),
            "value": round(world * args.steps * hImg * wImg / elapsed / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d C=%d radial path: C2P warp of both frames (polar %dx%d + 16 wrap columns), filter %s (random-init), "
                                   "SpatialRadialMatching(%d) + arg-min, P2C warp, flow2depth; one pair per GPU per step" % (wImg, hImg, Cc, wIn, hIn, layers, hWin),
                       "pairs_per_step": world, "sharding": "pair-per-gpu" if world > 1 else "single"},
            "roofline": {"bound": "hbm", "kernel": "radial_match_kernel", "achieved": round(balg / kern_s / 1e9, 2) if kern_s > 0 else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(balg / kern_s / 1e9 / HBM_PEAK_GBS, 4) if kern_s > 0 else None,
                         "traffic": None, "algorithmic_bytes_per_launch": balg, "kernel_ms": round(kern_s * 1e3, 5), "launches_timed": n.value},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_learned_single(args, world, rank, local_rank, dev, torch, dist, d, rp):
    """version2-* / time-matching: one pair per GPU per step.  `roofline` is the matcher (the dominant kernel): its algorithmic bytes
    (both feature maps + the H1 x W1 x hWin x wWin volume) against HBM, and next to it the vector-ALU bound of its 3 K lane-operations
    per output (subtract, multiply, add: the separately rounded k-ordered sum is the contract) -- the larger of the two is the bound."""
    import numpy as np

    H, W, nk, layers, win = LEARNED_SINGLE[args.workload]
    lib = d.lib()
    ctx = d.get_ctx(local_rank)
    g = torch.Generator().manual_seed(rank)
    tm = args.workload == "time-matching"
    hk = 1 + sum(l[2] - 1 for l in layers)
    wk = 1 + sum(l[1] - 1 for l in layers)
    K = layers[-1][3]
    if tm:
        geometry = dict(maxh=win, maxw=win, layers=[list(l) for l in layers], multiscale=False, prefilter=True)
        filt = d.getFilter(geometry, device=dev, generator=g)
        matcher = d.nn.SpatialMatching(win, win, False)
        im1 = torch.randn((3, H, W), generator=g).to(dev)
        im2 = torch.randn((3, H, W), generator=g).to(dev)
        H1, W1 = H - (hk - 1) - (win - 1), W - (wk - 1) - (win - 1)
        mn = torch.empty((W1 * H1,), device=dev)
        mi = torch.empty((W1 * H1,), dtype=torch.int64, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        stage = {"filter": 0.0, "match": 0.0, "min": 0.0, "n": 0}

        def step(timed=False):
            # tests/time_matching.lua:30-45 (the script filters im1 and im2 with the same module; prepareInput narrows the first)
            if timed: ev[0].record()
            f1 = filt.forward(im1)          # (a module's forward returns a fresh tensor here: no clone needed to keep im1's features)
            f2 = filt.forward(im2)
            p1, p2 = d.prepareInput(geometry, f1, f2)
            if timed: ev[1].record()
            out = matcher.forward([p1, p2])
            if timed: ev[2].record()
            # matcher:add(nn.Reshape(wsize*wsize, w-wsize+1-12, h-wsize+1-12)); output:min(1): the minimum over the leading dimension of
            # the [256][293 * 153] VIEW of the volume's memory (a reshape, not a transpose -- as the script has it)
            ctx.check(lib.dfe_min_dim0_f32(ctx.handle, out.data_ptr(), win * win, W1 * H1, mn.data_ptr(), mi.data_ptr()))
            if timed:
                ev[3].record()
                torch.cuda.synchronize()
                stage["filter"] += ev[0].elapsed_time(ev[1]); stage["match"] += ev[1].elapsed_time(ev[2]); stage["min"] += ev[2].elapsed_time(ev[3]); stage["n"] += 1
    elif nk < 0:
        geometry = dict(maxh=win, maxw=win, layers=[list(l) for l in layers], multiscale=False, output_extraction_method="max", hImg=H, wImg=W)
        model = d.getModel(geometry, True, False, device=dev, generator=g)
        f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=rank, max_flow=6)
        prev, cur = torch.from_numpy(f0 / np.float32(255)).to(dev), torch.from_numpy(f1 / np.float32(255)).to(dev)
        from depth_estimation_amd.multiscale import filter_layers_array

        arr, nl, keep = filter_layers_array([model.modules[0].modules[0]])
        H1, W1 = H - (hk - 1) - (win - 1), W - (wk - 1) - (win - 1)
        full, fconf = torch.empty((2, H, W), device=dev), torch.empty((H, W), device=dev)
        use_thr, thr = (1, 0.11) if nk == -2 else (0, 0.0)

        def step(timed=False):
            ctx.check(lib.dfe_flow_pair_filtered_f32(ctx.handle, prev.data_ptr(), cur.data_ptr(), 3, H, W, arr, nl, win, win, use_thr, thr, H, W,
                                                     full.data_ptr(), fconf.data_ptr(), None, None))
    else:
        if args.workload.endswith("-mfma"):
            ctx.set_option("conv_mfma", 1)
            ctx.set_option("fm_mfma", 1)
        datap = d.version2.defaultDatap(wImg=W, hImg=H, normalization_k=nk, layers=layers, wWin=win, hWin=win)
        net = d.version2.getNetwork(datap, device=dev, generator=g)
        f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=rank, max_flow=6)
        prev, cur = torch.from_numpy(f0 / np.float32(255)).to(dev), torch.from_numpy(f1 / np.float32(255)).to(dev)
        from depth_estimation_amd.multiscale import filter_layers_array

        stack = d.network.Sequential()
        for m in net.modules[0].modules[0].modules[2:]:
            stack.add(m)
        arr, nl, keep = filter_layers_array([stack])
        scn = net.modules[0].modules[0].modules[0]
        kp = scn.kernel.numpy().ctypes.data_as(C.POINTER(C.c_float))
        H1, W1 = H - (hk - 1) - (win - 1), W - (wk - 1) - (win - 1)
        xf, yf = torch.empty((H1, W1), device=dev), torch.empty((H1, W1), device=dev)

        def step(timed=False):
            ctx.check(lib.dfe_version2_flow_pair_f32(ctx.handle, prev.data_ptr(), cur.data_ptr(), 3, H, W, kp, scn.kernel.numel(), scn.threshold, scn.thresval,
                                                     arr, nl, win, win, xf.data_ptr(), yf.data_ptr(), None, None))

    elapsed = timed_region(step, args.steps, args.warmup, world, dist, dev, torch.cuda.synchronize,
                           before_timed=lambda: ctx.check(lib.dfe_profile_enable(ctx.handle, 1)))
    ms, n = C.c_double(), C.c_int()
    ctx.check(lib.dfe_profile_read(ctx.handle, C.byref(ms), C.byref(n)))
    ctx.check(lib.dfe_profile_enable(ctx.handle, 0))
    kernel = ctx.last_kernel()
    stages = None
    if tm:
        for _ in range(10):
            step(timed=True)
        stages = {k: round(stage[k] / stage["n"], 4) for k in ("filter", "match", "min")}
    else:
        ctx.check(lib.dfe_stage_timers_enable(ctx.handle, 1))
        for _ in range(10):
            step()
        sm, sr = (C.c_double * 4)(), (C.c_int * 4)()
        ctx.check(lib.dfe_stage_timers_read(ctx.handle, sm, sr))
        ctx.check(lib.dfe_stage_timers_enable(ctx.handle, 0))
        stages = {nm: round(sm[i] / 10, 4) for i, nm in enumerate(("load", "filter", "match", "extract"))}
    if rank == 0:
        out_elems = H1 * W1 * win * win
        fused_tail = nk < 0 or (nk > 0 and kernel.endswith("+argmin"))        # no volume leaves the matcher: its HBM bytes are the two feature maps
        balg = (K * H1 * W1 + K * (H1 + win - 1) * (W1 + win - 1) + (0 if fused_tail else out_elems) + (4 * H1 * W1 if fused_tail else 0)) * 4
        kern_s = ms.value / 1e3 / max(n.value, 1)
        step_s = elapsed / args.steps
        laneops = 3.0 * K * out_elems
        valu_s = laneops / 64.0 / (VALU_RATE_PER_CU * 256 * 2.4e9)
        hbm_s = balg / (HBM_PEAK_GBS * 1e9)
        bound = "valu" if valu_s > hbm_s else "hbm"
        mfma = None
        if args.workload.endswith("-mfma"):
            # both matrix-core kernels of a step together: the convolution's implicit GEMM (32 planes x 896 padded taps per output pixel of both
            # frames) and the matcher's banded GEMM (per 16 pixels and window row two 16 x 16 tiles over K + 4 taps); executed MFMA flops,
            # padding and the band's unused half included, against the dense f32 matrix-core peak
            pxa, pxb = (H - 16 - 16) * (W - 16 - 16), (H - 16) * (W - 16)
            conv_fl = 2.0 * 32 * 896 * (pxa + pxb)
            fm_fl = 2.0 * (-(-H1 // 8) * 8) * (-(-W1 // 16) * 16) * win * 32 * (K + 4)
            both_s = ms.value / 1e3 / args.steps                        # (profile scopes: one per matrix-core kernel, two per step)
            mfma = {"bound": "mfma", "kernel": "conv_mfma_res_kernel + fmm_kernel+argmin", "kernel_ms": round(both_s * 1e3, 5), "launches_timed": n.value,
                    "achieved": round((conv_fl + fm_fl) / both_s / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round((conv_fl + fm_fl) / both_s / 157.3e12, 4),
                    "traffic": None, "flops_per_step": conv_fl + fm_fl,
                    "note": "executed f32 MFMA flops (tap padding and the band's unused tile halves included) of both kernels over their summed time; the pipe "
                            "by itself delivers 155 TFLOP/s (tools/ubench/mfma_rate.hip), but every vector instruction of a wave -- operand addresses, the "
                            "arg-min epilogue -- adds its issue time: the f32 matrix pipe does not run beside the vector ALU (DESIGN 4.17)"}
        print(json.dumps({
            "metric": "Mpixels/s dense flow, %dx%d pair, learned single-scale model (%s), %dx%d window" % (
                W, H, "tests/time_matching.lua" if tm else "opticalflow_model.lua getModel + processOutput" if nk < 0 else "version2/network.lua", win, win),
            "value": round(world * args.steps * H * W / elapsed / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("tests/time_matching.lua:5-47: getFilter %s (tanh between, random-init) on two randn 3x%dx%d frames, prepareInput narrow, "
                                    "nn.SpatialMatching(%d,%d) on %d planes %dx%d, min over the script's Reshape(256, %d, %d) view" % (layers, H, W, win, win, K, W1, H1, W1, H1)) if tm else
                                   ("depth_estimation_opticalflow.lua:103-116 for a single-scale model in one call (dfe_flow_pair_filtered_f32): getFilter %s (tanh between, "
                                    "random-init) on both frames, prepareInput narrow, SpatialMatching(%d,%d) on %d planes %dx%d -> Minus -> SoftMax, processOutput(geometry, out, "
                                    "true, %s), no volume in HBM" % (layers, win, win, K, W1, H1, "0.11" if nk == -2 else "nil")) if nk < 0 else
                                   ("version2/test.lua:40-53 in one call (dfe_version2_flow_pair_f32): SpatialContrastiveNormalization(3, gaussian1D(%d)) on both frames, crop, "
                                    "SpatialConvolution %s (shared, random-init), SpatialMatching(%d,%d) on %d planes %dx%d, first-min decode" % (nk, layers, win, win, K, W1, H1)),
                       "pairs_per_step": world, "sharding": "pair-per-gpu" if world > 1 else "single", "stage_ms": stages,
                       "arith": "mfma (v_mfma_f32_16x16x4_f32: fused multiply-adds in the convolution, |a|^2+|b|^2-2ab in the matcher; tolerances in include/dfe.h)"
                                if args.workload.endswith("-mfma") else "exact (separately rounded k-ordered sums: bit-identical to the CPU restatement)"},
            "roofline": mfma if mfma else {"bound": bound, "kernel": kernel, "kernel_ms": round(kern_s * 1e3, 5), "launches_timed": n.value,
                         "achieved": round(balg / kern_s / 1e9, 2) if bound == "hbm" else round(laneops / kern_s / 1e12, 3),
                         "peak": HBM_PEAK_GBS if bound == "hbm" else round(64 * VALU_RATE_PER_CU * 256 * 2.4e9 / 1e12, 2),
                         "unit": "GB/s" if bound == "hbm" else "Tlane-op/s",
                         "frac": round((hbm_s if bound == "hbm" else valu_s) / kern_s, 4) if kern_s > 0 else None,
                         "traffic": None, "algorithmic_bytes_per_launch": balg,
                         "hbm_bound_ms": round(hbm_s * 1e3, 5), "valu_bound_ms": round(valu_s * 1e3, 5),
                         "note": "both bounds stated: bytes / 8 TB/s and 3 K lane-ops per output at %.1f plain fp32 wave-instructions per cycle and CU" % VALU_RATE_PER_CU},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="vga", choices=sorted(WORKLOADS) + sorted(PYRAMIDS) + sorted(RADIALS) + sorted(F16_WORKLOADS) + sorted(BAND_WORKLOADS) + sorted(LEARNED_SINGLE))
    ap.add_argument("--bands", type=int, default=1, help="*-bands workloads: at least this many row bands (default: one per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the (untimed) gather of the results to rank 0")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher (nothing below has touched the GPU yet)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist

    import depth_estimation_amd as d
    from tests import refpath as rp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libdfe has no CPU fallback")
    # Rehearsal on a one-GPU box only (tools/rehearse_ranks.sh): DFE_BENCH_DEVICE puts every rank on that device, DFE_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one GPU).  The driver's launch sets neither: one rank per GPU over RCCL.
    dev_index = int(os.environ.get("DFE_BENCH_DEVICE", local_rank))
    backend = os.environ.get("DFE_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    local_rank = dev_index
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.workload in PYRAMIDS:
        return main_pyramid(args, world, rank, local_rank, dev, torch, dist, d, rp)
    if args.workload in RADIALS:
        return main_radial(args, world, rank, local_rank, dev, torch, dist, d, rp)
    if args.workload in BAND_WORKLOADS:
        return main_bands(args, world, rank, local_rank, dev, torch, dist, d, rp)
    if args.workload in LEARNED_SINGLE:
        return main_learned_single(args, world, rank, local_rank, dev, torch, dist, d, rp)
    f16 = args.workload in F16_WORKLOADS
    H, W, Cc, k, hWin, wWin = (F16_WORKLOADS if f16 else WORKLOADS)[args.workload]
    (pair_id,) = shard_pairs(world, world, rank)                              # a batch of `world` pairs, pair p on rank p
    f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=Cc, seed=pair_id, max_flow=12)  # one seeded pair per rank
    t0, t1 = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
    flow = torch.empty((2, H, W), device=dev)
    scores = torch.empty((H, W), device=dev)
    depth = torch.empty((H, W), device=dev)
    dconf = torch.empty((H, W), device=dev)
    ctx = d.get_ctx(local_rank)
    lib = d.lib()

    def step():
        if f16:
            ctx.check(lib.dfe_flow_depth_pair_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, hWin, wWin, cx, cy, 2.0 ** -8, None, None,
                                                  flow.data_ptr(), depth.data_ptr(), dconf.data_ptr()))
            return
        ctx.check(
            lib.dfe_flow_depth_pair_f32(
                ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, hWin, wWin, cx, cy, 0.21,
                flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), dconf.data_ptr(),
            )
        )

    kernel = [None]

    def before_timed():
        kernel[0] = ctx.last_kernel()
        ctx.check(lib.dfe_profile_enable(ctx.handle, 1))

    elapsed = timed_region(step, args.steps, args.warmup, world, dist, dev, torch.cuda.synchronize, before_timed=before_timed)
    kernel = kernel[0]
    cv_ms, cv_n = C.c_double(), C.c_int()
    each = (C.c_float * max(args.steps * 4, 1))()
    ctx.check(lib.dfe_profile_read_each(ctx.handle, C.byref(cv_ms), C.byref(cv_n), each, len(each)))
    ctx.check(lib.dfe_profile_enable(ctx.handle, 0))
    each = sorted(each[: min(cv_n.value, len(each))])

    # outside the timed region: the cost-volume build ALONE (dfe_ssd_cost_volume_f32, no flow epilogue), the kernel the
    # north-star roofline target is stated for; reported as `roofline_build_only` next to the step's dominant kernel
    build_ms = build_kernel = None
    if rank == 0:
        Ho, Wo = H - k + 1 - hWin + 1, W - k + 1 - wWin + 1
        try:
            # the caller-owned volume, from the library's allocator (include/dfe.h dfe_device_alloc: physically contiguous where granted)
            vol, vol_contig = C.c_void_p(), C.c_int()
            rc_alloc = lib.dfe_device_alloc(ctx.handle, Ho * Wo * hWin * wWin * (2 if f16 else 4), C.byref(vol), C.byref(vol_contig))
            if rc_alloc != 0:
                raise torch.OutOfMemoryError()

            def build():
                if f16:
                    ctx.check(lib.dfe_ssd_cost_volume_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, k, hWin, wWin, 2.0 ** -8, vol))
                else:
                    ctx.check(lib.dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, k, hWin, wWin, vol))

            for _ in range(5):
                build()
            torch.cuda.synchronize()
            ctx.check(lib.dfe_profile_enable(ctx.handle, 1))
            for _ in range(20):
                build()
            b_ms, b_n = C.c_double(), C.c_int()
            ctx.check(lib.dfe_profile_read(ctx.handle, C.byref(b_ms), C.byref(b_n)))
            ctx.check(lib.dfe_profile_enable(ctx.handle, 0))
            build_ms, build_kernel = b_ms.value / 20, ctx.last_kernel()   # (per build = per pair, whatever the number of band launches)
            ctx.check(lib.dfe_device_free(ctx.handle, vol))
        except torch.OutOfMemoryError:
            pass

    gather = None
    if world > 1 and not args.no_gather and not f16:
        # results of the batch to rank 0 (flow as int16 x 2 -- displacements are integers in +-16 --, scores and depth fp32)
        res, g_s, g_bytes = gather_results([flow.to(torch.int16), scores, depth], world, rank, dist, torch.cuda.synchronize)
        if rank == 0:
            assert len(res) == world and torch.equal(res[0][1], scores)
            gather = {"ms": round(g_s * 1e3, 3), "bytes_to_rank0": g_bytes, "note": "untimed: after the K steps, one gather of every pair's {flow int16 x2, scores, depth}"}

    # "H2D excluded and included, both stated".  `value` above has the frames resident in HBM; this leg starts every step
    # from PINNED HOST uint8 frames (what a camera pipeline holds): two async copies on the step's stream + dfe_flow_depth_pair_u8
    # (conversion on the device), nothing overlapped -- the serial cost of the boundary; with two streams the copy of pair i+1 would hide
    # behind the step of pair i.
    h2d = None
    if rank == 0 and not f16 and Cc == 3:
        hu0 = torch.from_numpy(f0.astype(np.uint8)).pin_memory()
        hu1 = torch.from_numpy(f1.astype(np.uint8)).pin_memory()
        du0, du1 = torch.empty_like(hu0, device=dev), torch.empty_like(hu1, device=dev)

        def step_h2d():
            du0.copy_(hu0, non_blocking=True)
            du1.copy_(hu1, non_blocking=True)
            ctx.check(lib.dfe_flow_depth_pair_u8(ctx.handle, du0.data_ptr(), du1.data_ptr(), Cc, H, W, k, hWin, wWin, cx, cy, 0.21, 1.0,
                                                 flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), dconf.data_ptr()))

        for _ in range(5):
            step_h2d()
        torch.cuda.synchronize()
        nh = max(20, min(args.steps, 100))
        th = time.perf_counter()
        for _ in range(nh):
            step_h2d()
        torch.cuda.synchronize()
        th = (time.perf_counter() - th) / nh
        h2d = {"value": round(H * W / th / 1e6, 3), "unit": "Mpixels/s", "ms_per_step": round(th * 1e3, 4), "host_bytes_per_step": int(2 * hu0.numel()),
               "note": "pinned host uint8 frames -> device per step (2 async copies + on-device conversion + the same pipeline), serial on one stream; "
                       "fp32 frames would move 4x the bytes"}
        # ... and PIPELINED (host staging is what bounds the multi-GPU scaling): dfe_ingest_submit_u8 uploads pair i+1 on the
        # ctx's copy stream into another of three device slots while dfe_flow_depth_pair_u8_slot computes pair i -- event-ordered, no host
        # synchronisation inside the loop; every step still starts from host memory
        slot = C.c_int()
        nb = int(hu0.numel())

        def submit():
            ctx.check(lib.dfe_ingest_submit_u8(ctx.handle, hu0.data_ptr(), hu1.data_ptr(), nb, C.byref(slot)))
            return slot.value

        def run_pipelined(n):
            cur = submit()
            for _ in range(n):
                nxt = submit()
                ctx.check(lib.dfe_flow_depth_pair_u8_slot(ctx.handle, cur, Cc, H, W, k, hWin, wWin, cx, cy, 0.21, 1.0,
                                                          flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), dconf.data_ptr()))
                cur = nxt

        run_pipelined(5)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        run_pipelined(nh)
        torch.cuda.synchronize()
        tp = (time.perf_counter() - tp) / nh
        h2d["pipelined"] = {"value": round(H * W / tp / 1e6, 3), "unit": "Mpixels/s", "ms_per_step": round(tp * 1e3, 4),
                            "of_resident": round((H * W / tp / 1e6) / (world * args.steps * H * W / elapsed / 1e6), 4),
                            "note": "copy-engine transfer of pair i+1 (copy stream, another of three device slots) under the step of pair i: dfe_ingest_submit_u8 + dfe_flow_depth_pair_u8_slot"}

    scale1080 = None
    if world > 1 and args.workload == "vga":
        # the same step on one 1080p pair per GPU (8.6 GB of volume each: fits the default 16-GiB arena in one band)
        H2, W2 = WORKLOADS["1080p"][:2]
        g0, g1, _, (cx2, cy2) = rp.synth_pair(H2, W2, C=Cc, seed=pair_id, max_flow=12)
        u0, u1 = torch.from_numpy(g0).to(dev), torch.from_numpy(g1).to(dev)
        fl2 = torch.empty((2, H2, W2), device=dev)
        sc2, de2, co2 = (torch.empty((H2, W2), device=dev) for _ in range(3))

        def step1080():
            ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, u0.data_ptr(), u1.data_ptr(), Cc, H2, W2, k, hWin, wWin, cx2, cy2, 0.21,
                                                  fl2.data_ptr(), sc2.data_ptr(), de2.data_ptr(), co2.data_ptr()))

        scale1080 = scale_leg(step1080, max(5, args.steps // 4), max(2, args.warmup // 4), world, dist, dev, torch.cuda.synchronize, H2, W2)
        del u0, u1, fl2, sc2, de2, co2

    if rank == 0:
        balg = algorithmic_bytes(H, W, Cc, k, hWin, wWin, 2 if f16 else 4)
        # (a frame whose volume exceeds the scratch limit is built in several bands = several launches per step: the
        #  algorithmic bytes are the pair's, so the time is the sum of the step's cost-volume launches)
        kern_s = cv_ms.value / 1e3 / max(args.steps, 1)
        achieved = balg / kern_s / 1e9 if kern_s > 0 else 0.0
        traffic = build_traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath):  # HBM bytes per launch from rocprofv3 --pmc passes (profiles/README.md)
            try:
                tj = json.load(open(tpath))
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from kernel_hash import kernel_source_hash

                same_src = tj.get("source_sha256") == kernel_source_hash()
                if same_src and tj.get("kernel_rev") == lib.dfe_kernel_revision().decode():   # counters of another kernel version say nothing
                    traffic = tj.get("hbm_bytes_per_launch")
                    build_traffic = tj.get("build_only", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = build_traffic = None
        out = {
            "metric": "Mpixels/s dense flow+depth, %dx%d pair, 7x7 patch +-16 search" % (W, H),
            "value": round(world * args.steps * H * W / elapsed / 1e6, 3),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 sums, f16 volume" if f16 else "f32",
            "data": "synthetic",
            "config": {
                "workload": "%dx%d C=%d single-scale dense SSD cost volume (7x7 patch, %dx%d window = +-16)%s "
                "+ arg-min/%sdecode + flow->depth, one pair per GPU per step" % (W, H, Cc, hWin, wWin, " stored as fp16 (cost * 2^-8)" if f16 else "",
                                                                               "" if f16 else "extractOutput/"),
                "pairs_per_step": world,
                "sharding": "pair-per-gpu" if world > 1 else "single",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": balg,
                # sliding-window flops of the build, (3C+4) per cell: shows the kernel is not compute-priced
                "gflops_sliding_window": round((3 * Cc + 4) * (H - k - hWin + 2) * (W - k - wWin + 2) * hWin * wWin / kern_s / 1e9, 1)
                if kern_s > 0 else None,
                "kernel_ms": round(kern_s * 1e3, 5),
                # per launch, over the timed region (the average above is what `frac` prices)
                "kernel_ms_min_median_max": [round(each[0], 5), round(each[len(each) // 2], 5), round(each[-1], 5)] if each else None,
                "launches_timed": cv_n.value,
                "launches_per_step": cv_n.value // max(args.steps, 1),
            },
        }
        if build_ms:
            out["roofline_build_only"] = {
                "bound": "hbm", "kernel": build_kernel, "achieved": round(balg / (build_ms / 1e3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(balg / (build_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": build_traffic,
                "kernel_ms": round(build_ms, 5),
                "launches_timed": 20, "note": "dfe_ssd_cost_volume_f32 alone, measured after the timed region",
            }
        if gather:
            out["gather"] = gather
        if h2d:
            out["h2d_inclusive"] = h2d
        if scale1080:
            out["scale_1080p"] = scale1080
        out["rccl"] = rccl_info(dist, world)
        if world == 1 and not args.no_cpu_baseline and not f16:
            out["cpu_baseline"] = cpu_baseline(f0, f1, k, hWin, wWin, cx, cy)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
