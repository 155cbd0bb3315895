import sys, numpy as np, torch
sys.path.insert(0, '.')
import depth_estimation_amd as d
from tests import oracle as orc, refpath as rp
dev = torch.device('cuda:0')
cases = [(67,46,3,7,33,33),(67,50,3,7,33,33),(70,46,3,7,33,33),(80,100,3,7,33,33),(80,101,3,7,33,33),(80,102,3,7,33,33)]
ctx = d.get_ctx(0)
for (H,W,C,k,hW,wW) in cases:
    f0,f1,_,_ = rp.synth_pair(H,W,C=C,seed=H+W,max_flow=5)
    cpu = orc.ssd_cost_volume(f0,f1,k,k,hW,wW)
    ctx.set_cost_volume_kernel(2)
    out = torch.full(cpu.shape, -1.0, device=dev)
    t0=torch.from_numpy(f0).to(dev); t1=torch.from_numpy(f1).to(dev)
    ctx.check(d.lib().dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C,H,W,k,k,hW,wW,out.data_ptr()))
    torch.cuda.synchronize()
    g = out.cpu().numpy()
    bad = g != cpu
    print((H,W), 'shape', cpu.shape, 'bad frac', bad.mean(), 'unwritten', (g==-1).mean())
    if bad.any():
        by = bad.any(axis=(1,2,3)); bx = bad.any(axis=(0,2,3)); bdy = bad.any(axis=(0,1,3)); bdx = bad.any(axis=(0,1,2))
        print(' bad rows', np.nonzero(by)[0][:50].tolist(), ' bad cols', np.nonzero(bx)[0][:50].tolist(), ' bad dy', np.nonzero(bdy)[0].tolist(), ' bad dx', np.nonzero(bdx)[0].tolist())
        i = np.argwhere(bad)[0]; print(' first', i, g[tuple(i)], cpu[tuple(i)])
