"""Times the tiled cost-volume kernel per forced tile height: python tools/tune_cv.py [workloads...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d
from tests import refpath as rp
from bench import WORKLOADS, algorithmic_bytes
dev = torch.device("cuda:0")
ctx = d.get_ctx(0)
lib = d.lib()
for wl in (sys.argv[1:] or ["vga"]):
    H, W, Cc, k, hW, wW = WORKLOADS[wl]
    f0, f1, _, _ = rp.synth_pair(H, W, C=Cc, seed=0)
    t0, t1 = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
    out = torch.empty((H - k - hW + 2, W - k - wW + 2, hW, wW), device=dev)
    balg = algorithmic_bytes(H, W, Cc, k, hW, wW)
    mode = int(os.environ.get("CV_MODE", "2"))
    ctx.set_cost_volume_kernel(mode)
    codes = [int(c) for c in os.environ["CV_TILES"].split(",")] if "CV_TILES" in os.environ else None
    for tyq in (codes or ((0, 0, 5, 1, 0, 5, 1) if mode == 3 else (4, 2, 3, 4, 5))):   # (first entry repeats: clocks ramp up during it)
        ctx.set_cost_volume_tile(tyq)
        for _ in range(3):
            ctx.check(lib.dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, k, hW, wW, out.data_ptr()))
        torch.cuda.synchronize()
        ctx.check(lib.dfe_profile_enable(ctx.handle, 1))
        for _ in range(20):
            ctx.check(lib.dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, k, hW, wW, out.data_ptr()))
        ms, n = C.c_double(), C.c_int()
        ctx.check(lib.dfe_profile_read(ctx.handle, C.byref(ms), C.byref(n)))
        ctx.check(lib.dfe_profile_enable(ctx.handle, 0))
        t = ms.value / n.value
        print("%s mode=%d tyq=%d  %.1f us  %.0f GB/s  %.1f%% of 8TB/s" % (wl, mode, tyq, t * 1e3, balg / t / 1e6, balg / t / 1e6 / 80))
    ctx.set_cost_volume_tile(0)
    del out
