"""Runs only the hot path (N launches) for profiling:
    python tools/prof_cv.py [workload] [n] [what]
what = "pair" (default): dfe_flow_depth_pair_f32, i.e. the fused build + finalize + border/depth pass of the bench step;
what = 0..3: dfe_ssd_cost_volume_f32 alone with that kernel mode (0 auto, 1 reference order, 2 tiled, 3 row-image)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d
from tests import refpath as rp
from bench import WORKLOADS, F16_WORKLOADS, PYRAMIDS
wl = sys.argv[1] if len(sys.argv) > 1 else "vga"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
what = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("CV_MODE", "pair")
spec = WORKLOADS.get(wl) or F16_WORKLOADS.get(wl) or PYRAMIDS[wl]
if what == "pyramid":        # dfe_multiscale_flow_pair_f32 of a *-pyramid workload (7-tuple: ..., ratios)
    from depth_estimation_amd._lib import ratios_array
    H, W, C, k, hW, wW, ratios = spec
    f0, f1, _, _ = rp.synth_pair(H, W, C=C, seed=0)
    dev = torch.device("cuda:0")
    t0, t1 = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
    ctx = d.get_ctx(0)
    rr, nr = ratios_array(list(ratios))
    flow = torch.empty((2, H, W), device=dev)
    idx = torch.empty((H, W), dtype=torch.int64, device=dev)
    for _ in range(n):
        ctx.check(d.lib().dfe_multiscale_flow_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, hW, wW, rr, nr, flow.data_ptr(), idx.data_ptr()))
    torch.cuda.synchronize()
    print("ran", n, what, ctx.last_kernel())
    sys.exit(0)
H, W, C, k, hW, wW = spec
f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=C, seed=0)
dev = torch.device("cuda:0")
t0, t1 = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
ctx = d.get_ctx(0)
if what == "pair":
    flow = torch.empty((2, H, W), device=dev)
    scores, depth, conf = (torch.empty((H, W), device=dev) for _ in range(3))
    for _ in range(n):
        ctx.check(d.lib().dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, hW, wW, cx, cy, 0.21,
                                                 flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), conf.data_ptr()))
elif what == "pair16":   # dfe_flow_depth_pair_f16: the fp16-volume step
    flow = torch.empty((2, H, W), device=dev)
    depth, conf = (torch.empty((H, W), device=dev) for _ in range(2))
    for _ in range(n):
        ctx.check(d.lib().dfe_flow_depth_pair_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, hW, wW, cx, cy, 2.0 ** -8, None, None,
                                                 flow.data_ptr(), depth.data_ptr(), conf.data_ptr()))
    print("scratch arena", ctx.scratch_info() if hasattr(ctx, "scratch_info") else "")
elif what == "build16":
    out = torch.empty((H - k - hW + 2, W - k - wW + 2, hW, wW), device=dev, dtype=torch.float16)
    for _ in range(n):
        ctx.check(d.lib().dfe_ssd_cost_volume_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, k, hW, wW, 2.0 ** -8, out.data_ptr()))
else:
    out = torch.empty((H - k - hW + 2, W - k - wW + 2, hW, wW), device=dev)
    ctx.set_cost_volume_kernel(int(what))
    for _ in range(n):
        ctx.check(d.lib().dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, k, hW, wW, out.data_ptr()))
torch.cuda.synchronize()
print("ran", n, what, ctx.last_kernel())
