#!/usr/bin/env python3
"""Tuning only: the flat feature matcher's three forms on one shape -- volume, first-minimum decode, soft-max + arg-max (no threshold) --
HIP-event time per call.  usage: prof_fm_forms.py [shape of tools/prof_fm.py] [n]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_fm import SHAPES

shape = sys.argv[1] if len(sys.argv) > 1 else "k10"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
K, H1, W1, win = SHAPES[shape]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
in1 = (torch.rand((K, H1, W1), generator=g) * 3).to(dev)
in2 = (torch.rand((K, H1 + win - 1, W1 + win - 1), generator=g) * 3).to(dev)
ctx = d.get_ctx(0); lib = d.lib()
vol = torch.empty((H1, W1, win, win), device=dev)
idx = torch.empty((H1, W1), dtype=torch.int64, device=dev); xf = torch.empty((H1, W1), device=dev); yf = torch.empty((H1, W1), device=dev)
full = torch.empty((2, H1 + win - 1, W1 + win - 1), device=dev); fc = torch.empty((H1 + win - 1, W1 + win - 1), device=dev)
# prefiltered single-scale call: feature maps of the full size, patch 1 narrowed inside
a_full = (torch.rand((K, H1 + win - 1, W1 + win - 1), generator=g) * 3).to(dev)


def timeit(f):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


t_vol = timeit(lambda: ctx.check(lib.dfe_spatial_matching_f32(ctx.handle, in1.data_ptr(), in2.data_ptr(), K, H1, W1, win, win, vol.data_ptr())))
k1 = ctx.last_kernel()
t_arg = timeit(lambda: ctx.check(lib.dfe_spatial_matching_argmin_f32(ctx.handle, in1.data_ptr(), in2.data_ptr(), K, H1, W1, win, win, idx.data_ptr(), xf.data_ptr(), yf.data_ptr())))
k2 = ctx.last_kernel()
for thr in (0, 1):
    t_soft = timeit(lambda: ctx.check(lib.dfe_flow_pair_filtered_f32(ctx.handle, a_full.data_ptr(), in2.data_ptr(), K, H1 + win - 1, W1 + win - 1, None, 0, win, win, thr, 0.3,
                                                                      H1 + win - 1, W1 + win - 1, full.data_ptr(), fc.data_ptr(), idx.data_ptr(), None)))
    print("%s K=%d %dx%d win %d: soft-max form (threshold %s) %.1f us (%s)" % (shape, K, W1, H1, win, "on" if thr else "off", t_soft, ctx.last_kernel()))
print("%s: volume %.1f us (%s) | first minimum %.1f us (%s)" % (shape, t_vol, k1, t_arg, k2))
