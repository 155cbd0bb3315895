"""Tuning only: the single-scale learned model of version2/network.lua at VGA -- contrastive normalisation, 17 x 17 x 32 convolution (direct
and MFMA implicit GEMM), nn.SpatialMatching(17, 17) on the 32-plane features -- per-stage times with torch.cuda events."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import depth_estimation_amd as dfe
from depth_estimation_amd import network as net, nn as dnn

dev = torch.device("cuda:0")
H, W = 480, 640
g = torch.Generator().manual_seed(0)
x0 = torch.rand((3, H, W), generator=g).to(dev)
x1 = torch.rand((3, H, W), generator=g).to(dev)
scn = net.SpatialContrastiveNormalization(3, net.gaussian1D(17))
conv = net.SpatialConvolution(3, 32, 17, 17, device=dev, generator=g)
match = dnn.SpatialMatching(17, 17, False)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

print("contrastive normalisation %.3f ms" % timed(lambda: scn.forward(x0)))
n0 = scn.forward(x0).clone(); n1 = scn.forward(x1).clone()
for kern in ("exact", "mfma"):
    conv.kernel = kern
    print("conv 3->32 17x17 (%s) %.3f ms" % (kern, timed(lambda: conv.forward(n0))))
f0 = conv.forward(n0).clone(); f1 = conv.forward(n1).clone()
hc, wc = f0.shape[1] - 16, f0.shape[2] - 16
f0c = f0[:, 8:8 + hc, 8:8 + wc].contiguous()
print("SpatialMatching(17,17) on 32 planes %dx%d: %.3f ms" % (hc, wc, timed(lambda: match.forward([f0c, f1]))))

# the rows matcher's own shape: 16 x 16 window, K = 32 (and tests/time_matching.lua's K = 10)
m16 = dnn.SpatialMatching(16, 16, False)
for K in (32, 10):
    a = torch.rand((K, H - 15, W - 15), generator=g).to(dev)
    b = torch.rand((K, H, W), generator=g).to(dev)
    print("SpatialMatching(16,16) on %d planes %dx%d: %.3f ms (%s)" % (K, H - 15, W - 15, timed(lambda: m16.forward([a, b])), dfe.get_ctx(0).last_kernel()))
