"""Tuning only: does an asynchronous host-to-device copy on a second stream block the host / overlap with the sweep?"""
import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import depth_estimation_amd as d
from tests import refpath as rp
dev = torch.device("cuda:0"); ctx = d.get_ctx(0); lib = d.lib()
H, W = 480, 640
f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=3, seed=0, max_flow=12)
hu0 = torch.from_numpy(f0.astype(np.uint8)).pin_memory(); hu1 = torch.from_numpy(f1.astype(np.uint8)).pin_memory()
du = [[torch.empty_like(hu0, device=dev), torch.empty_like(hu1, device=dev)] for _ in range(2)]
flow = torch.empty((2, H, W), device=dev); scores, depth, dconf = (torch.empty((H, W), device=dev) for _ in range(3))
s2 = torch.cuda.Stream()
def comp(b):
    ctx.check(lib.dfe_flow_depth_pair_u8(ctx.handle, du[b][0].data_ptr(), du[b][1].data_ptr(), 3, H, W, 7, 33, 33, cx, cy, 0.21, 1.0, flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), dconf.data_ptr()))
def copy(b):
    with torch.cuda.stream(s2):
        du[b][0].copy_(hu0, non_blocking=True); du[b][1].copy_(hu1, non_blocking=True)
for b in (0, 1): copy(b)
torch.cuda.synchronize()
for _ in range(20): comp(0)
torch.cuda.synchronize()
n = 200
for mode in ("no events", "events"):
    tcp = tcm = 0.0
    t = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        if mode == "events": s2.wait_stream(torch.cuda.current_stream())
        copy((i + 1) & 1)
        b = time.perf_counter()
        if mode == "events": torch.cuda.current_stream().wait_stream(s2)
        comp(i & 1)
        c = time.perf_counter()
        tcp += b - a; tcm += c - b
    torch.cuda.synchronize()
    print("%s: wall %.1f us per step; host: copy %.1f us, compute call %.1f us" % (mode, (time.perf_counter() - t) / n * 1e6, tcp / n * 1e6, tcm / n * 1e6))
