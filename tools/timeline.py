"""Tuning only: phase timeline of the row-image kernel from a side build with -DDFE_TIMELINE=1 (tools/mklib.sh tl -DDFE_TIMELINE=1).
usage: DFE_LIB=tools/ubench/libdfe_tl.so python tools/timeline.py [vga] [build|fused]
Prints, for blocks 0 and 100, the mean cycles (s_memtime) each wave spends per row between the stamps:
  main task | ring step, next scalars, deposit | quarter / mini task | wait + barrier | refill / scan | copy-out | loop"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import depth_estimation_amd as d
from bench import WORKLOADS
from tests import refpath as rp

wl = sys.argv[1] if len(sys.argv) > 1 else "vga"
what = sys.argv[2] if len(sys.argv) > 2 else "build"
dev = torch.device("cuda:0")
ctx = d.get_ctx(0)
lib = d.lib()
H, W, Cc, k, hW, wW = WORKLOADS[wl]
f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=Cc, seed=0)
t0, t1 = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
if what == "build":
    out = torch.empty((H - k - hW + 2, W - k - wW + 2, hW, wW), device=dev)
    ctx.set_cost_volume_kernel(3)
    step = lambda: ctx.check(lib.dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, k, hW, wW, out.data_ptr()))
else:
    flow = torch.empty((2, H, W), device=dev)
    scores, depth, dconf = (torch.empty((H, W), device=dev) for _ in range(3))
    step = lambda: ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, hW, wW, cx, cy, 0.21,
                                                         flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), dconf.data_ptr()))
for _ in range(20):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ctx.check(lib.dfe_profile_enable(ctx.handle, 1))
for _ in range(10):
    step()
ms, n = C.c_double(), C.c_int()
ctx.check(lib.dfe_profile_read(ctx.handle, C.byref(ms), C.byref(n)))
ctx.check(lib.dfe_profile_enable(ctx.handle, 0))
print("kernel time %.1f us" % (ms.value / n.value * 1e3))
mask = int(os.environ.get("TL_MASK", "127"))
tl = np.zeros((2, 16, 256, 8), np.uint64)
raw = C.CDLL(os.environ["DFE_LIB"])
rc = raw.dfe_debug_timeline(tl.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
tl = tl.astype(np.int64)
names = ["main", "dep", "q/m", "bar", "rf/scan", "copy", "loop"]
if mask != 127:   # two stamps only: point k relative to the row start, and the row period
    k = [i for i in range(1, 7) if (mask >> i) & 1][0]
    for b in range(2):
        t = tl[b]
        per = (t[:, 21:121, 0] - t[:, 20:120, 0]).mean(axis=1)
        off = (t[:, 20:120, k] - t[:, 20:120, 0]).mean(axis=1)
        span = t[0, 120, 0] - t[0, 20, 0]
        print("block %d point %d: period " % (b * 100, k) + " ".join("%5.0f" % v for v in per))
        print("block %d point %d: offset " % (b * 100, k) + " ".join("%5.0f" % v for v in off))
    sys.exit(0)
for b in range(2):
    t = tl[b]
    rows = slice(20, 120)
    per = (t[:, 21:121, 0] - t[:, 20:120, 0]).mean(axis=1)
    print("block %d: row period %.0f cycles (wave 0)" % (b * 100, per[0]))
    print("  wave " + " ".join("%8s" % n for n in names))
    for w in range(16):
        ph = [(t[w, rows, i + 1] - t[w, rows, i]).mean() for i in range(6)]
        ph.append((t[w, 21:121, 0] - t[w, 20:120, 6]).mean())
        print("  %4d " % w + " ".join("%8.0f" % v for v in ph))
    # skew: when does each wave reach the barrier relative to the last one
    arr = t[:, rows, 3]
    last = arr.max(axis=0)
    print("  barrier arrival before the last wave (cycles): " + " ".join("%d" % v for v in (last[None, :] - arr).mean(axis=1)))
