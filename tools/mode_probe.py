#!/usr/bin/env python3
"""Tuning only: is the per-process bimodal step time of a fused sweep (vga-luma: 0.212 / 0.242 ms kernels) tied to where the ctx's scratch
arena (the cost volume) lands?  Several contexts in ONE process, each with its own arena (debug_arena prints the address), timed in turn;
then the first one again.  usage: mode_probe.py [workload] [n_ctx]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d
from depth_estimation_amd.context import Context
from tests import refpath as rp
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "vga-luma"
n_ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H, W, Cc, k, hWin, wWin = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=Cc, seed=0, max_flow=12)
t0, t1 = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
flow = torch.empty((2, H, W), device=dev); scores = torch.empty((H, W), device=dev); depth = torch.empty((H, W), device=dev); dconf = torch.empty((H, W), device=dev)
lib = d.lib()


def timeit(ctx, n=int(os.environ.get('PROBE_N', '150'))):
    def step():
        ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, hWin, wWin, cx, cy, 0.21, flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), dconf.data_ptr()))
    for _ in range(int(os.environ.get('PROBE_WARM', '20'))): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


ctxs = []
pads = []
for i in range(n_ctx):
    ctx = Context(0)
    ctx.set_option("debug_arena", 1)
    if os.environ.get("PROBE_CONTIG") and i % 2 == 1: ctx.set_option("arena_contig", 1)
    ms = [timeit(ctx) for _ in range(3)]
    print("ctx %d: %s ms" % (i, " ".join("%.4f" % m for m in ms)), flush=True)
    ctxs.append(ctx)
    pads.append(torch.empty(((i + 1) * 3 << 20) + 4096 * (i + 1), dtype=torch.uint8, device=dev))   # shift what comes next
for i, ctx in enumerate(ctxs):
    print("again ctx %d: %.4f ms" % (i, timeit(ctx)), flush=True)
