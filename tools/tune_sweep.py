"""Tuning only: interleaved timing of cost-volume kernel configurations at sustained clocks.
usage: tune_sweep.py WORKLOAD fused|build  cfg [cfg...]     cfg = tile[:ENV=VAL[,ENV=VAL]]   (tile = dfe_set_cost_volume_tile code)
Each configuration is timed over ROUNDS (env, default 4) rounds of 20 launches with HIP events on the kernel's stream
(dfe_profile_*), configurations interleaved, after an 0.3 s spin; prints min / median us and the fraction of 8 TB/s."""
import ctypes as C
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import depth_estimation_amd as d
from bench import WORKLOADS, algorithmic_bytes
from tests import refpath as rp

wl, what = sys.argv[1], sys.argv[2]
cfgs = sys.argv[3:] or ["0"]
rounds = int(os.environ.get("ROUNDS", "4"))
dev = torch.device("cuda:0")
ctx = d.get_ctx(0)
lib = d.lib()
H, W, Cc, k, hW, wW = WORKLOADS[wl]
f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=Cc, seed=0)
t0, t1 = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
balg = algorithmic_bytes(H, W, Cc, k, hW, wW)
if what == "build":
    out = torch.empty((H - k - hW + 2, W - k - wW + 2, hW, wW), device=dev)
    ctx.set_cost_volume_kernel(3)

    def step():
        ctx.check(lib.dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, k, hW, wW, out.data_ptr()))
else:
    flow = torch.empty((2, H, W), device=dev)
    scores, depth, dconf = (torch.empty((H, W), device=dev) for _ in range(3))

    def step():
        ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, hW, wW, cx, cy, 0.21,
                                              flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), dconf.data_ptr()))


def apply(cfg):
    tile, _, envs = cfg.partition(":")
    for kv in filter(None, envs.split(",")):
        key, _, val = kv.partition("=")
        os.environ[key] = val
    ctx.set_cost_volume_tile(int(tile))
    return [kv.partition("=")[0] for kv in filter(None, envs.split(","))]


t_end = time.time() + 0.3
while time.time() < t_end:
    step()
torch.cuda.synchronize()
res = {c: [] for c in cfgs}
for r in range(rounds):
    for c in cfgs:
        keys = apply(c)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ctx.check(lib.dfe_profile_enable(ctx.handle, 1))
        for _ in range(20):
            step()
        ms, n = C.c_double(), C.c_int()
        ctx.check(lib.dfe_profile_read(ctx.handle, C.byref(ms), C.byref(n)))
        ctx.check(lib.dfe_profile_enable(ctx.handle, 0))
        res[c].append(ms.value / n.value * 1e3)
        for key in keys:
            os.environ.pop(key, None)
for c in cfgs:
    v = sorted(res[c])
    med = statistics.median(v)
    print("%s %s %-28s min %.1f med %.1f us  -> %.3f of 8 TB/s (kernel %s)" % (wl, what, c, v[0], med, balg / (med * 1e-6) / 8e12, ctx.last_kernel()), flush=True)
