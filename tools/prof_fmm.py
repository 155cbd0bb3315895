"""Tuning only: the matrix-core matcher's arg-min form (dfe_spatial_matching_argmin_f32 under fm_mfma = 1) at version2's VGA shape,
HIP-event time per call of the kernel alone (ctx profile).  usage: python tools/prof_fmm.py [n]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
K, H1, W1, win = 32, 448, 608, 17
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
in1 = torch.rand((K, H1, W1), generator=g).to(dev)
in2 = torch.rand((K, H1 + win - 1, W1 + win - 1), generator=g).to(dev)
ctx = d.get_ctx(0)
ctx.set_option("fm_mfma", 1)
idx = torch.empty((H1, W1), dtype=torch.int64, device=dev)
xf, yf = torch.empty((H1, W1), device=dev), torch.empty((H1, W1), device=dev)
lib = d.lib()
def call():
    ctx.check(lib.dfe_spatial_matching_argmin_f32(ctx.handle, in1.data_ptr(), in2.data_ptr(), K, H1, W1, win, win, idx.data_ptr(), xf.data_ptr(), yf.data_ptr()))
for _ in range(5): call()
torch.cuda.synchronize()
ctx.check(lib.dfe_profile_enable(ctx.handle, 1))
for _ in range(n): call()
torch.cuda.synchronize()
ms, k = C.c_double(), C.c_int()
ctx.check(lib.dfe_profile_read(ctx.handle, C.byref(ms), C.byref(k)))
print("%s: %s %.1f us per launch (%d launches)" % (os.environ.get("DFE_LIB", "product").split("/")[-1], ctx.last_kernel(), ms.value * 1e3 / max(k.value, 1), k.value))
