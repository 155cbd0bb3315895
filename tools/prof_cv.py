"""Runs only the cost-volume kernel (N launches) for profiling: python tools/prof_cv.py [workload] [n] [mode]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d
from tests import refpath as rp
from bench import WORKLOADS
wl = sys.argv[1] if len(sys.argv) > 1 else "vga"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = int(sys.argv[3]) if len(sys.argv) > 3 else int(os.environ.get("CV_MODE", "2"))
H, W, C, k, hW, wW = WORKLOADS[wl]
f0, f1, _, _ = rp.synth_pair(H, W, C=C, seed=0)
dev = torch.device("cuda:0")
t0, t1 = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
out = torch.empty((H - k - hW + 2, W - k - wW + 2, hW, wW), device=dev)
ctx = d.get_ctx(0)
ctx.set_cost_volume_kernel(mode)
for _ in range(n):
    ctx.check(d.lib().dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, k, hW, wW, out.data_ptr()))
torch.cuda.synchronize()
print("ran", n, ctx.last_kernel())
