import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import depth_estimation_amd as d
from tests import refpath as rp
dev = torch.device("cuda:0"); ctx = d.get_ctx(0); lib = d.lib()
H, W = 480, 640
f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=3, seed=0, max_flow=12)
hu0 = torch.from_numpy(f0.astype(np.uint8)).pin_memory(); hu1 = torch.from_numpy(f1.astype(np.uint8)).pin_memory()
flow = torch.empty((2, H, W), device=dev); scores, depth, dconf = (torch.empty((H, W), device=dev) for _ in range(3))
slot = C.c_int(); nb = hu0.numel()
def submit():
    ctx.check(lib.dfe_ingest_submit_u8(ctx.handle, hu0.data_ptr(), hu1.data_ptr(), nb, C.byref(slot))); return slot.value
def comp(s):
    ctx.check(lib.dfe_flow_depth_pair_u8_slot(ctx.handle, s, 3, H, W, 7, 33, 33, cx, cy, 0.21, 1.0, flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), dconf.data_ptr()))
s0 = submit(); s1 = submit(); torch.cuda.synchronize()
for _ in range(20): comp(s0)
torch.cuda.synchronize()
n = 200
t = time.perf_counter()
for _ in range(n): comp(s0)
torch.cuda.synchronize(); print("compute only (slot resident): %.1f us" % ((time.perf_counter() - t) / n * 1e6))
t = time.perf_counter()
for _ in range(n): submit()
th = time.perf_counter() - t
torch.cuda.synchronize(); print("submit only: host %.1f us, wall %.1f us" % (th / n * 1e6, (time.perf_counter() - t) / n * 1e6))
t = time.perf_counter()
cur = submit()
for _ in range(n):
    nxt = submit(); comp(cur); cur = nxt
th = time.perf_counter() - t
torch.cuda.synchronize(); print("pipelined: host %.1f us, wall %.1f us" % (th / n * 1e6, (time.perf_counter() - t) / n * 1e6))
ts = tc = 0.0
cur = submit()
for _ in range(n):
    a = time.perf_counter(); nxt = submit(); b = time.perf_counter(); comp(cur); c = time.perf_counter()
    ts += b - a; tc += c - b; cur = nxt
torch.cuda.synchronize(); print("pipelined host split: submit %.1f us, compute call %.1f us" % (ts / n * 1e6, tc / n * 1e6))
