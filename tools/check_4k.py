"""One 3840x2160 pair through dfe_flow_depth_pair_f32 (the volume is 35 GB: banded by the 16 GiB scratch limit); checks the
planted flow is recovered on a central crop and prints the time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import depth_estimation_amd as d
from tests import refpath as rp
H, W, k, win = 2160, 3840, 7, 33
f0, f1, flow, (cx, cy) = rp.synth_pair(H, W, C=3, seed=1, max_flow=12, noise_sigma=0)
dev = torch.device("cuda:0")
t0, t1 = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
out = torch.empty((2, H, W), device=dev); sc = torch.empty((H, W), device=dev); dp = torch.empty((H, W), device=dev); cf = torch.empty((H, W), device=dev)
ctx = d.get_ctx(0)
def step():
    ctx.check(d.lib().dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, win, win, cx, cy, 0.21,
                                             out.data_ptr(), sc.data_ptr(), dp.data_ptr(), cf.data_ptr()))
step(); torch.cuda.synchronize()
t = time.perf_counter(); step(); step(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 2
g = out.cpu().numpy()
inner = (slice(40, H - 40), slice(40, W - 40))
ok = ((g[0] == flow[0]) & (g[1] == flow[1]))[inner].mean()
print("4K pair: %.2f ms, %.0f Mpixels/s, kernel %s, planted flow recovered on %.1f %% of the interior" % (dt * 1e3, H * W / dt / 1e6, ctx.last_kernel(), 100 * ok))
assert ok > 0.9
