"""Times the multiscale pipeline (cfg2b: 640x480, ratios {1,2,4}, 8x8 window per scale, 7x7 patch) through the host mirror:
per-scale cost volume (A2) -> softmin (A3) -> cascade + ring extraction (A4/A5) -> processOutput (A6/A10/A11)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d
from tests import refpath as rp
H, W = 480, 640
geo = dict(maxh=8, maxw=8, ratios=[1, 2, 4], multiscale=True, hKernel=7, wKernel=7, hImg=H, wImg=W, output_extraction_method="max")
f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=0)
dev = torch.device("cuda:0")
t0, t1 = torch.from_numpy(f0 / 64).to(dev), torch.from_numpy(f1 / 64).to(dev)
model = d.getModelMultiscale(geo)
for _ in range(5):
    out = model.forward([t0, t1]); ret = d.processOutput(geo, out, True)
torch.cuda.synchronize()
n = 50
a = time.perf_counter()
for _ in range(n):
    out = model.forward([t0, t1])
torch.cuda.synchronize()
b = time.perf_counter()
for _ in range(n):
    ret = d.processOutput(geo, out, True)
torch.cuda.synchronize()
c = time.perf_counter()
for _ in range(5):
    ret2 = model.forwardFlow([t0, t1], True)
torch.cuda.synchronize()
e = time.perf_counter()
for _ in range(n):
    ret2 = model.forwardFlow([t0, t1], True)
torch.cuda.synchronize()
f = time.perf_counter()
print("forwardFlow (one C call) %.3f ms -> %.1f Mpixels/s" % ((f - e) / n * 1e3, H * W / ((f - e) / n) / 1e6))
for _ in range(5):
    ret3 = model.forwardFlow([t0, t1], True, one_call=False)
torch.cuda.synchronize()
e = time.perf_counter()
for _ in range(n):
    ret3 = model.forwardFlow([t0, t1], True, one_call=False)
torch.cuda.synchronize()
f = time.perf_counter()
print("forwardFlow (staged) %.3f ms -> %.1f Mpixels/s" % ((f - e) / n * 1e3, H * W / ((f - e) / n) / 1e6))
print("multiscale forward %.3f ms, processOutput %.3f ms -> %.1f Mpixels/s (host-driven, includes per-call allocation)" %
      ((b - a) / n * 1e3, (c - b) / n * 1e3, H * W / ((c - a) / n) / 1e6))
