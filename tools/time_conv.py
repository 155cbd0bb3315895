"""Tuning / evidence: nn.SpatialConvolution as the direct (bit-exact) kernel and as the MFMA implicit GEMM, on the layer shapes the
reference uses: tests/time_matching.lua:13 ({3,5,5,4},{4,5,5,4},{4,5,5,10} on 320x180), version2/network.lua (3 -> 32, 17x17)
and the radial separable stack (1x17, 17x1) at 720p.  usage: python tools/time_conv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d

dev = torch.device("cuda:0")
ctx = d.get_ctx(0)
lib = d.lib()
cases = [("time_matching L1 3->4 5x5 320x180", 3, 4, 5, 5, 180, 320), ("time_matching L2 4->4 5x5", 4, 4, 5, 5, 176, 316),
         ("time_matching L3 4->10 5x5", 4, 10, 5, 5, 172, 312), ("version2 3->32 17x17 720p", 3, 32, 17, 17, 720, 1280),
         ("version2 3->32 17x17 VGA", 3, 32, 17, 17, 480, 640), ("radial 3->5 1x17 720p", 3, 5, 1, 17, 720, 1296), ("radial 5->10 17x1 720p", 5, 10, 17, 1, 720, 1280)]
for name, nIn, nOut, kH, kW, H, W in cases:
    x = torch.randn((nIn, H, W), device=dev)
    w = torch.randn((nOut, nIn, kH, kW), device=dev) / (nIn * kH * kW) ** 0.5
    b = torch.randn((nOut,), device=dev)
    out = torch.empty((nOut, H - kH + 1, W - kW + 1), device=dev)
    flops = 2.0 * nIn * kH * kW * out.numel()
    res = []
    for fast in (0, 1):
        def run():
            if fast:
                ctx.check(lib.dfe_spatial_convolution_mfma_f32(ctx.handle, x.data_ptr(), w.data_ptr(), b.data_ptr(), nIn, nOut, H, W, kH, kW, 0, out.data_ptr()))
            else:
                ctx.check(lib.dfe_spatial_convolution_f32(ctx.handle, x.data_ptr(), w.data_ptr(), b.data_ptr(), nIn, nOut, H, W, kH, kW, out.data_ptr()))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    print("%-34s direct %8.3f ms (%6.2f TFLOP/s)   mfma %8.3f ms (%6.2f TFLOP/s)   x%.1f" % (name, res[0], flops / res[0] / 1e9, res[1], flops / res[1] / 1e9, res[0] / res[1]), flush=True)
