#!/usr/bin/env python3
"""Tuning only: the batched convolution (4 -> 10 planes, 5 x 5, 480 x 640) with its input / output in plain hipMalloc memory or in physically
contiguous memory (hipExtMallocWithFlags, hipDeviceMallocContiguous = 0x4), of the exact size or carved from a large allocation."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import depth_estimation_amd as d

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
ctx = d.get_ctx(0)
lib = d.lib()
nIn, nOut, H, W, k = 4, 10, 476, 636, 5
Ho, Wo = H - k + 1, W - k + 1
w = torch.randn(nOut, nIn, k, k, device=dev); b = torch.randn(nOut, device=dev)
src = torch.randn(nIn, H, W, device=dev)


def alloc(kind, nbytes):
    p = C.c_void_p()
    if kind == "contig":
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(0x4))
    else:
        rc = hip.hipMalloc(C.byref(p), C.c_size_t(nbytes))
    assert rc == 0, rc
    return p.value


def run(pin, pout, n=300):
    def step():
        ctx.check(lib.dfe_spatial_convolution_f32(ctx.handle, C.c_void_p(pin), w.data_ptr(), b.data_ptr(), nIn, nOut, H, W, k, k, C.c_void_p(pout)))
    for _ in range(20): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


in_b, out_b = nIn * H * W * 4, nOut * Ho * Wo * 4
for rep in range(2):
    for kin in ("plain", "contig"):
        for kout in ("plain", "contig"):
            for big in (0, 1):
                pi = alloc(kin, in_b if not big else 64 << 20)
                po = alloc(kout, out_b if not big else 256 << 20)
                hip.hipMemcpy(C.c_void_p(pi), C.c_void_p(src.data_ptr()), C.c_size_t(in_b), C.c_int(3))
                us = run(pi, po)
                print("in %-6s out %-6s %s  in %#x out %#x : %.1f us  (%s)" % (kin, kout, "big  " if big else "exact", pi, po, us, ctx.last_kernel()), flush=True)
                hip.hipFree(C.c_void_p(pi)); hip.hipFree(C.c_void_p(po))

# one allocation, input at 0, output behind it at several distances (the arena's layout: feature buffers side by side)
print("-- one allocation: in at 0, out at +delta")
al = lambda n, a: (n + a - 1) // a * a
for rep in range(2):
    for kind in ("plain", "contig"):
        base = alloc(kind, 512 << 20)
        hip.hipMemcpy(C.c_void_p(base), C.c_void_p(src.data_ptr()), C.c_size_t(in_b), C.c_int(3))
        for delta in (al(in_b, 256), al(in_b, 4096), al(in_b, 1 << 16), al(in_b, 1 << 21), al(in_b, 1 << 21) + (1 << 12), 10 * 480 * 640 * 4, 2 * 10 * 480 * 640 * 4, 64 << 20, (64 << 20) + 4352):
            us = run(base, base + delta)
            print("%-6s delta %10d (%#x): %.1f us" % (kind, delta, delta, us), flush=True)
        hip.hipFree(C.c_void_p(base))
