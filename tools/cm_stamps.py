#!/usr/bin/env python3
"""Tuning only: phase stamps (s_memtime) of the matrix-core convolution's tile loop, block 3, first tiles -- needs a -DDFE_CM_STAMPS=1 side build
(DFE_LIB=tools/ubench/libdfe_cmstamps.so): the stamps overwrite the head of the output, which this script reads back."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import depth_estimation_amd as d
dev = torch.device("cuda", 0)
ctx = d.get_ctx(0); lib = d.lib()
nIn, nOut, k, H, W = 3, 32, 17, 528, 1040
w = torch.randn(nOut, nIn, k, k, device=dev); b = torch.randn(nOut, device=dev)
src = torch.randn(nIn, H, W, device=dev); out = torch.empty(nOut, H - k + 1, W - k + 1, device=dev)
for _ in range(5):
    ctx.check(lib.dfe_spatial_convolution_mfma_f32(ctx.handle, src.data_ptr(), w.data_ptr(), b.data_ptr(), nIn, nOut, H, W, k, k, 0, out.data_ptr()))
torch.cuda.synchronize()
st = out.view(torch.int32).reshape(-1)[: 8 * 16 * 5].cpu().numpy().astype(np.uint32).reshape(8, 16, 5)
t0 = st[0, :, 0].min()
print("tile wave  top->loop  loop  epilogue  barrier-wait | top (rel. to first)")
for it in range(1, 7):
    for wv in range(16):
        a = st[it, wv].astype(np.int64)
        d_ = np.diff(a) & 0xffffffff
        print("%3d %4d   %8d %8d %8d %8d | %d" % (it, wv, d_[0], d_[1], d_[2], d_[3], (a[0] - t0) & 0xffffffff))
    print()
