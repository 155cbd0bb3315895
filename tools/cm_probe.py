#!/usr/bin/env python3
"""Tuning only: the resident-weights matrix-core convolution (3 -> 32 planes, 17 x 17) on frames of growing size: launch + weight staging +
n tile rounds.  usage: cm_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d
dev = torch.device("cuda", 0)
ctx = d.get_ctx(0); lib = d.lib()
nIn, nOut, k = 3, 32, 17
w = torch.randn(nOut, nIn, k, k, device=dev); b = torch.randn(nOut, device=dev)
for H, W in ((20, 80), (20, 16 + 64 * 16), (16 + 4 * 16, 16 + 64 * 16), (16 + 4 * 64, 16 + 64 * 16), (16 + 4 * 128, 16 + 64 * 16), (480, 640), (16 + 4 * 256, 16 + 64 * 16)):
    src = torch.randn(nIn, H, W, device=dev); out = torch.empty(nOut, H - k + 1, W - k + 1, device=dev)
    def step():
        ctx.check(lib.dfe_spatial_convolution_mfma_f32(ctx.handle, src.data_ptr(), w.data_ptr(), b.data_ptr(), nIn, nOut, H, W, k, k, 0, out.data_ptr()))
    for _ in range(20): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 200
    for _ in range(n): step()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / n * 1e6
    tiles = ((H - k + 1 + 3) // 4) * ((W - k + 1 + 63) // 64)
    print("%4d x %4d: %5d tiles = %.2f rounds  %.1f us  (%s)" % (H, W, tiles, tiles / 256, us, ctx.last_kernel()), flush=True)
