"""ctypes/numpy binding of the CPU oracle (oracle/dfe_oracle.c).  Test infrastructure: imported
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
# DFE_ORACLE_SO: another build of the same source (the AddressSanitizer build of `make -C oracle asan`, tests/test_sanitizer_cpu.py)
SO = os.environ.get("DFE_ORACLE_SO") or os.path.join(ODIR, "libdfe_oracle.so")

_lib = None
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build():
    if os.environ.get("DFE_ORACLE_SO"):
        return
    src = os.path.join(ODIR, "dfe_oracle.c")
    if (not os.path.exists(SO)) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ODIR, "dfe_oracle.h"))):
        subprocess.check_call(["make", "-C", ODIR, "libdfe_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(SO)
        sig = {
            "orc_set_num_threads": (None, [C.c_int]),
            "orc_get_max_threads": (C.c_int, []),
            "orc_unfold": (None, [f32p] + [C.c_int] * 5 + [f32p]),
            "orc_spatial_matching": (None, [f32p, f32p] + [C.c_int] * 5 + [f32p]),
            "orc_ssd_cost_volume": (None, [f32p, f32p] + [C.c_int] * 7 + [f32p, C.c_int, C.c_int]),
            "orc_radial_matching": (None, [f32p, f32p] + [C.c_int] * 4 + [f32p]),
            "orc_argbest_center": (None, [f32p, C.c_int64, C.c_int, C.c_int, C.c_int, i64p, C.c_void_p]),
            "orc_extract_output": (None, [f32p, C.c_int64, C.c_int, C.c_double, i64p, f32p]),
            "orc_extract_output_marginalized": (None, [f32p, C.c_int64, C.c_int, C.c_double, C.c_double, i64p, i64p]),
            "orc_x2yx": (None, [i64p, C.c_int64, C.c_int, C.c_int, i64p, i64p]),
            "orc_yx2x_multi": (C.c_int64, [C.c_int, C.c_int, i32p, C.c_int, C.c_double, C.c_double]),
            "orc_x2yx_multi_number": (C.c_int, [C.c_int, C.c_int, i32p, C.c_int, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
            "orc_x2yx_multi": (C.c_int, [C.c_int, C.c_int, i32p, C.c_int, i64p, C.c_int64, i64p, i64p]),
            "orc_multi_nclasses": (C.c_int64, [C.c_int, C.c_int, i32p, C.c_int]),
            "orc_x2yx_multi_compat_c": (None, [C.c_int, C.c_int, i32p, C.c_int, i64p, C.c_int64, i64p, i64p]),
            "orc_downsample_box": (None, [f32p] + [C.c_int] * 4 + [f32p]),
            "orc_zero_pad": (None, [f32p] + [C.c_int] * 7 + [f32p]),
            "orc_pyramid_scale_volume": (None, [f32p, f32p] + [C.c_int] * 8 + [f32p]),
            "orc_softmin": (None, [f32p, C.c_int64, C.c_int, f32p]),
            "orc_cascade_ring": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
            "orc_cascading_add": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, i32p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
            "orc_spatial_convolution": (None, [f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
            "orc_spatial_convolution_map": (None, [f32p, f32p, C.c_void_p, i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
            "orc_spatial_convolution_fma": (None, [f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
            "orc_contrastive_normalization": (None, [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_float, C.c_float, f32p]),
            "orc_tanh": (None, [f32p, C.c_int64, f32p]),
            "orc_rgb2y": (None, [f32p, C.c_int, C.c_int, f32p]),
            "orc_spatial_convolution_grad_input": (None, [f32p, f32p, C.c_void_p] + [C.c_int] * 7 + [f32p]),
            "orc_spatial_convolution_acc_grad": (None, [f32p, f32p, C.c_void_p] + [C.c_int] * 7 + [C.c_float, f32p, C.c_void_p]),
            "orc_tanh_backward": (None, [f32p, f32p, C.c_int64, f32p]),
            "orc_log_softmax": (None, [f32p, C.c_int64, C.c_int, f32p]),
            "orc_log_softmax_backward": (None, [f32p, f32p, C.c_int64, C.c_int, f32p]),
            "orc_softmax_backward": (None, [f32p, f32p, C.c_int64, C.c_int, f32p]),
            "orc_spatial_matching_backward": (None, [f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p]),
            "orc_radial_matching_backward": (None, [f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p]),
            "orc_cascading_add_backward": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, i32p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
            "orc_paste_center": (None, [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_int]),
            "orc_flow_to_depth_cartesian": (None, [f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, f32p, f32p]),
            "orc_flow_to_depth_radial": (None, [f32p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, f32p, f32p]),
            "orc_flow_to_depth_ardrone": (None, [f32p, f32p, C.c_int, C.c_int, C.c_float, f32p, f32p]),
            "orc_marginal_sum": (None, [f32p, C.c_int64, C.c_int, C.c_int, f32p]),
            "orc_polar_grid_c2p": (None, [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, f32p]),
            "orc_polar_grid_p2c": (None, [C.c_int] * 4 + [C.c_float] * 4 + [f32p]),
            "orc_warp_bilinear": (None, [f32p] + [C.c_int] * 3 + [f32p, C.c_int, C.c_int, f32p]),
            "orc_postprocess_image": (C.c_int, [f32p, f32p] + [C.c_int] * 4 + [f32p]),
            "orc_enlarge_mask": (None, [f32p] + [C.c_int] * 4),
            "orc_output_extractor": (None, [f32p, C.c_int64, C.c_int, C.c_int, f32p, f32p]),
            "orc_epipole": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]),
            "orc_remove_ego_motion": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, f32p, C.c_void_p]),
            "orc_undistort_image": (None, [f32p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), f32p]),
            "orc_foe_from_flow": (C.c_int, [f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
            "orc_ego_motion_from_points": (C.c_int, [f32p, f32p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_uint, C.POINTER(C.c_double),
                                                     C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
        }
        for n, (r, a) in sig.items():
            f = getattr(l, n)
            f.restype = r
            f.argtypes = a
        _lib = l
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _r(ratios):
    return np.ascontiguousarray(ratios, dtype=np.int32)


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def max_threads():
    return lib().orc_get_max_threads()


def unfold(img, kh, kw):
    img = _f(img)
    Cc, H, W = img.shape
    out = np.empty((Cc * kh * kw, H - kh + 1, W - kw + 1), np.float32)
    lib().orc_unfold(img, Cc, H, W, kh, kw, out)
    return out


def spatial_matching(in1, in2, maxh, maxw):
    in1, in2 = _f(in1), _f(in2)
    K, H1, W1 = in1.shape
    assert in2.shape == (K, H1 + maxh - 1, W1 + maxw - 1)
    out = np.empty((H1, W1, maxh, maxw), np.float32)
    lib().orc_spatial_matching(in1, in2, K, H1, W1, maxh, maxw, out)
    return out


def ssd_cost_volume(I0, I1, kh, kw, hWin, wWin, row0=0, row1=None, out=None):
    I0, I1 = _f(I0), _f(I1)
    Cc, H, W = I0.shape
    Ho, Wo = H - kh + 1 - hWin + 1, W - kw + 1 - wWin + 1
    if out is None:
        out = np.zeros((Ho, Wo, hWin, wWin), np.float32)
    lib().orc_ssd_cost_volume(I0, I1, Cc, H, W, kh, kw, hWin, wWin, out, row0, Ho if row1 is None else row1)
    return out


def radial_matching(in1, in2, hWin):
    in1, in2 = _f(in1), _f(in2)
    K, H1, W = in1.shape
    out = np.empty((H1, W, hWin), np.float32)
    lib().orc_radial_matching(in1, in2, K, H1, W, hWin, out)
    return out


def argbest_center(vol, middle, take_max):
    vol = _f(vol)
    N = vol.shape[-1]
    P = vol.size // N
    idx = np.empty(vol.shape[:-1], np.int64)
    best = np.empty(vol.shape[:-1], np.float32)
    lib().orc_argbest_center(vol.reshape(P, N), P, N, middle, int(take_max), idx.reshape(-1), best.ctypes.data_as(C.c_void_p))
    return idx, best


def extract_output(inp, threshold, imaxs, scores):
    """in place, like the reference"""
    inp = _f(inp)
    N = inp.shape[-1]
    P = inp.size // N
    lib().orc_extract_output(inp.reshape(P, N), P, N, float(threshold), imaxs.reshape(-1), scores.reshape(-1))


def extract_output_marginalized(inp, threshold, threshold_acc, ret, retgd):
    inp = _f(inp)
    N = inp.shape[-1]
    P = inp.size // N
    lib().orc_extract_output_marginalized(inp.reshape(P, N), P, N, float(threshold), float(threshold_acc), ret.reshape(-1), retgd.reshape(-1))


def x2yx(idx, maxh, maxw):
    idx = np.ascontiguousarray(idx, np.int64)
    y, x = np.empty_like(idx), np.empty_like(idx)
    lib().orc_x2yx(idx.reshape(-1), idx.size, maxh, maxw, y.reshape(-1), x.reshape(-1))
    return y, x


def yx2x_multi(maxh, maxw, ratios, y, x):
    r = _r(ratios)
    return int(lib().orc_yx2x_multi(maxh, maxw, r, len(r), float(y), float(x)))


def x2yx_multi_number(maxh, maxw, ratios, i):
    r = _r(ratios)
    y, x = C.c_int64(), C.c_int64()
    rc = lib().orc_x2yx_multi_number(maxh, maxw, r, len(r), int(i), C.byref(y), C.byref(x))
    return rc, int(y.value), int(x.value)


def x2yx_multi(maxh, maxw, ratios, idx):
    r = _r(ratios)
    idx = np.ascontiguousarray(idx, np.int64)
    y, x = np.empty_like(idx), np.empty_like(idx)
    rc = lib().orc_x2yx_multi(maxh, maxw, r, len(r), idx.reshape(-1), idx.size, y.reshape(-1), x.reshape(-1))
    return rc, y, x


def x2yx_multi_compat_c(maxh, maxw, ratios, idx, fill=0):
    r = _r(ratios)
    idx = np.ascontiguousarray(idx, np.int64)
    y, x = np.full_like(idx, fill), np.full_like(idx, fill)
    lib().orc_x2yx_multi_compat_c(maxh, maxw, r, len(r), idx.reshape(-1), idx.size, y.reshape(-1), x.reshape(-1))
    return y, x


def multi_nclasses(maxh, maxw, ratios):
    r = _r(ratios)
    return int(lib().orc_multi_nclasses(maxh, maxw, r, len(r)))


def downsample_box(img, r):
    img = _f(img)
    Cc, H, W = img.shape
    out = np.empty((Cc, H // r, W // r), np.float32)
    lib().orc_downsample_box(img, Cc, H, W, r, out)
    return out


def zero_pad(img, pl, pr, pt, pb):
    img = _f(img)
    Cc, H, W = img.shape
    out = np.empty((Cc, H + pt + pb, W + pl + pr), np.float32)
    lib().orc_zero_pad(img, Cc, H, W, pl, pr, pt, pb, out)
    return out


def pyramid_scale_volume(I0, I1, r, kh, kw, maxh, maxw):
    I0, I1 = _f(I0), _f(I1)
    Cc, H, W = I0.shape
    out = np.empty((H // r, W // r, maxh, maxw), np.float32)
    lib().orc_pyramid_scale_volume(I0, I1, Cc, H, W, r, kh, kw, maxh, maxw, out)
    return out


def softmin(cost):
    cost = _f(cost)
    N = cost.shape[-1] if cost.ndim == 2 else cost.shape[-1] * cost.shape[-2]
    P = cost.size // N
    out = np.empty_like(cost)
    lib().orc_softmin(cost.reshape(P, N), P, N, out.reshape(P, N))
    return out


def cascade_ring(probs, ratios, H, W, maxh, maxw):
    probs = [_f(p) for p in probs]
    r = _r(ratios)
    ncls = multi_nclasses(maxh, maxw, ratios)
    out = np.empty((H, W, ncls), np.float32)
    arr = (C.c_void_p * len(probs))(*[p.ctypes.data for p in probs])
    rc = lib().orc_cascade_ring(arr, len(probs), r, H, W, maxh, maxw, out)
    return rc, out


def cascading_add(ins, ratios, maxh, maxw):
    ins = [_f(p) for p in ins]
    P = ins[0].size // (maxh * maxw)
    outs = [np.empty_like(p) for p in ins]
    r = _r(ratios)
    a = (C.c_void_p * len(ins))(*[p.ctypes.data for p in ins])
    b = (C.c_void_p * len(ins))(*[p.ctypes.data for p in outs])
    rc = lib().orc_cascading_add(a, len(ins), r, P, maxh, maxw, b)
    return rc, outs


def spatial_convolution(inp, weight, bias):
    inp, weight = _f(inp), _f(weight)
    nOut, nIn, kH, kW = weight.shape
    _, H, W = inp.shape
    out = np.empty((nOut, H - kH + 1, W - kW + 1), np.float32)
    b = _f(bias) if bias is not None else None
    lib().orc_spatial_convolution(inp, weight, b.ctypes.data if b is not None else None, nIn, nOut, H, W, kH, kW, out)
    return out


def spatial_convolution_fma(inp, weight, bias):
    inp, weight = _f(inp), _f(weight)
    nOut, nIn, kH, kW = weight.shape
    _, H, W = inp.shape
    out = np.empty((nOut, H - kH + 1, W - kW + 1), np.float32)
    b = _f(bias) if bias is not None else None
    lib().orc_spatial_convolution_fma(inp, weight, b.ctypes.data if b is not None else None, nIn, nOut, H, W, kH, kW, out)
    return out


def spatial_convolution_map(inp, weight, bias, conn, nOut):
    inp, weight = _f(inp), _f(weight)
    nConn, kH, kW = weight.shape
    nIn, H, W = inp.shape
    conn = np.ascontiguousarray(conn, np.int32)
    out = np.empty((nOut, H - kH + 1, W - kW + 1), np.float32)
    b = _f(bias) if bias is not None else None
    lib().orc_spatial_convolution_map(inp, weight, b.ctypes.data if b is not None else None, conn.reshape(-1), nConn, nIn, nOut, H, W, kH, kW, out)
    return out


def gaussian1D(size, sigma=0.25, amplitude=1.0, normalize=False):
    """image.gaussian1D [3P-recall]: g[i] = amplitude * exp(-((i - center) / (sigma * size))^2 / 2), i = 1..size, center = size / 2 + 0.5"""
    i = np.arange(1, size + 1, dtype=np.float64)
    g = amplitude * np.exp(-(((i - (size / 2 + 0.5)) / (sigma * size)) ** 2) / 2)
    if normalize:
        g = g / g.sum()
    return g.astype(np.float32)


def rgb2y(rgb):
    rgb = _f(rgb)
    _, H, W = rgb.shape
    y = np.empty((1, H, W), np.float32)
    lib().orc_rgb2y(rgb, H, W, y)
    return y


def contrastive_normalization(inp, kernel, threshold=1e-4, thresval=1e-4):
    inp, kernel = _f(inp), _f(kernel)
    Cc, H, W = inp.shape
    out = np.empty_like(inp)
    lib().orc_contrastive_normalization(inp, Cc, H, W, kernel, kernel.size, threshold, thresval, out)
    return out


def tanh(inp):
    inp = _f(inp)
    out = np.empty_like(inp)
    lib().orc_tanh(inp.reshape(-1), inp.size, out.reshape(-1))
    return out


def spatial_convolution_backward(inp, weight, go, conn=None, nOut=None, scale=1.0, with_bias=True):
    """(gradInput, gradWeight, gradBias) of nn.SpatialConvolution (conn None, weight nOut x nIn x kH x kW) or
    nn.SpatialConvolutionMap (weight nConn x kH x kW, conn nConn x 2 (from, to) 1-based), gradients accumulated from zero."""
    inp, weight, go = _f(inp), _f(weight), _f(go)
    nIn, H, W = inp.shape
    if conn is None:
        nOut, _, kH, kW = weight.shape
        cp, nConn = None, 0
    else:
        nConn, kH, kW = weight.shape
        conn = np.ascontiguousarray(conn, np.int32)
        cp = conn.ctypes.data
    gi = np.empty_like(inp)
    gw = np.zeros_like(weight)
    gb = np.zeros(nOut, np.float32)
    lib().orc_spatial_convolution_grad_input(go, weight, cp, nConn, nIn, nOut, H, W, kH, kW, gi)
    lib().orc_spatial_convolution_acc_grad(inp, go, cp, nConn, nIn, nOut, H, W, kH, kW, scale, gw, gb.ctypes.data if with_bias else None)
    return gi, gw, gb


def tanh_backward(out, go):
    out, go = _f(out), _f(go)
    gi = np.empty_like(out)
    lib().orc_tanh_backward(out.reshape(-1), go.reshape(-1), out.size, gi.reshape(-1))
    return gi


def log_softmax(x):
    x = _f(x)
    out = np.empty_like(x)
    lib().orc_log_softmax(x.reshape(-1), x.size // x.shape[-1], x.shape[-1], out.reshape(-1))
    return out


def log_softmax_backward(out, go):
    out, go = _f(out), _f(go)
    gi = np.empty_like(out)
    lib().orc_log_softmax_backward(out.reshape(-1), go.reshape(-1), out.size // out.shape[-1], out.shape[-1], gi.reshape(-1))
    return gi


def softmax_backward(out, go):
    out, go = _f(out), _f(go)
    gi = np.empty_like(out)
    lib().orc_softmax_backward(out.reshape(-1), go.reshape(-1), out.size // out.shape[-1], out.shape[-1], gi.reshape(-1))
    return gi


def spatial_matching_backward(in1, in2, go, maxh, maxw):
    in1, in2, go = _f(in1), _f(in2), _f(go)
    K, H1, W1 = in1.shape
    g1, g2 = np.empty_like(in1), np.empty_like(in2)
    lib().orc_spatial_matching_backward(in1, in2, go, K, H1, W1, maxh, maxw, g1, g2)
    return g1, g2


def radial_matching_backward(in1, in2, go, hWin):
    in1, in2, go = _f(in1), _f(in2), _f(go)
    K, H1, W = in1.shape
    g1, g2 = np.empty_like(in1), np.empty_like(in2)
    lib().orc_radial_matching_backward(in1, in2, go, K, H1, W, hWin, g1, g2)
    return g1, g2


def cascading_add_backward(grad_outs, ratios, maxh, maxw):
    gos = [_f(p) for p in grad_outs]
    P = gos[0].size // (maxh * maxw)
    gis = [np.empty_like(p) for p in gos]
    r = _r(ratios)
    a = (C.c_void_p * len(gos))(*[p.ctypes.data for p in gos])
    b = (C.c_void_p * len(gos))(*[p.ctypes.data for p in gis])
    rc = lib().orc_cascading_add_backward(a, len(gos), r, P, maxh, maxw, b)
    return rc, gis


def flow_to_depth_cartesian(flow, cx, cy, fix_dot=False):
    flow = _f(flow)
    _, H, W = flow.shape
    d, c = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
    lib().orc_flow_to_depth_cartesian(flow, H, W, cx, cy, int(fix_dot), d, c)
    return d, c


def flow_to_depth_radial(rflow, cx, cy, infty):
    rflow = _f(rflow)
    H, W = rflow.shape
    d, c = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
    lib().orc_flow_to_depth_radial(rflow, None, H, W, cx, cy, infty, d, c)
    return d, c


def marginal_sum(inp, A, B):
    inp = _f(inp)
    P = inp.size // (A * B)
    out = np.empty((P, A), np.float32)
    lib().orc_marginal_sum(inp.reshape(-1), P, A, B, out)
    return out


def flow_to_depth_ardrone(xflow, mask, m):
    xflow, mask = _f(xflow), _f(mask)
    H, W = xflow.shape
    d, c = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
    lib().orc_flow_to_depth_ardrone(xflow, mask, H, W, m, d, c)
    return d, c


def polar_grid_c2p(wsrc, hsrc, wdst, hdst, xc, yc, lpad, rpad, rmax, alpha=1.0):
    m = np.empty((2, hdst, wdst + lpad + rpad), np.float32)
    lib().orc_polar_grid_c2p(wsrc, hsrc, wdst, hdst, xc, yc, lpad, rpad, rmax, alpha, m)
    return m


def polar_grid_p2c(wsrc, hsrc, wdst, hdst, xc, yc, rmax, alpha=1.0):
    m = np.empty((2, hdst, wdst), np.float32)
    lib().orc_polar_grid_p2c(wsrc, hsrc, wdst, hdst, xc, yc, rmax, alpha, m)
    return m


def warp_bilinear(img, mask):
    img, mask = _f(img), _f(mask)
    Cc, H, W = img.shape
    _, Hd, Wd = mask.shape
    out = np.empty((Cc, Hd, Wd), np.float32)
    lib().orc_warp_bilinear(img, Cc, H, W, mask, Hd, Wd, out)
    return out


def postprocess_image(flow, mask, k, method):
    flow, mask = _f(flow), _f(mask)
    _, H, W = flow.shape
    out = np.empty_like(flow)
    rc = lib().orc_postprocess_image(flow, mask, H, W, k, 0 if method == "max" else 1, out)
    return rc, out


def enlarge_mask(mask, ix, iy):
    m = _f(mask).copy()
    lib().orc_enlarge_mask(m, m.shape[0], m.shape[1], ix, iy)
    return m


def output_extractor(inp, maxh, maxw):
    inp = _f(inp)
    P = inp.size // (maxh * maxw)
    x, y = np.empty(inp.shape[:-1], np.float32), np.empty(inp.shape[:-1], np.float32)
    lib().orc_output_extractor(inp.reshape(P, maxh * maxw), P, maxh, maxw, x.reshape(-1), y.reshape(-1))
    return x, y


# ---- next-row N4 (sfm2 call sites; parity unpinned) ------------------------------------------------------------------
def _dv(a, n):
    v = np.asarray(a, np.float64).reshape(-1)
    assert v.size == n
    return (C.c_double * n)(*v.tolist())


def epipole(K, T, scale=1.0):
    e = (C.c_double * 2)()
    rc = lib().orc_epipole(_dv(K, 9), _dv(T, 3), float(scale), e)
    return rc, (e[0], e[1])


def remove_ego_motion(img, K, R, inverse=False):
    img = _f(img)
    Cc, H, W = img.shape
    out, mask = np.empty_like(img), np.empty((H, W), np.float32)
    rc = lib().orc_remove_ego_motion(img, Cc, H, W, _dv(K, 9), _dv(R, 9), int(inverse), out, mask.ctypes.data)
    assert rc == 0
    return out, mask


def undistort_image(img, K, dist):
    img = _f(img)
    Cc, H, W = img.shape
    out = np.empty_like(img)
    lib().orc_undistort_image(img, Cc, H, W, _dv(K, 9), _dv(dist, 5), out)
    return out


def foe_from_flow(flow, conf=None, min_flow=0.5, iterations=2):
    fy, fx = _f(flow[0]), _f(flow[1])
    H, W = fy.shape
    c = _f(conf) if conf is not None else None
    out, n = (C.c_double * 2)(), C.c_double()
    rc = lib().orc_foe_from_flow(fy, fx, c.ctypes.data if c is not None else None, H, W, float(min_flow), int(iterations), out, C.byref(n))
    return rc, (out[0], out[1]), n.value


def ego_motion_from_points(p1, p2, K, max_dist, iterations, seed, weights=None):
    p1, p2 = _f(p1), _f(p2)
    N = p1.shape[0]
    w = _f(weights) if weights is not None else None
    R, T, F, ni = (C.c_double * 9)(), (C.c_double * 3)(), (C.c_double * 9)(), C.c_int()
    rc = lib().orc_ego_motion_from_points(p1, p2, w.ctypes.data if w is not None else None, N, _dv(K, 9), float(max_dist), int(iterations), int(seed), R, T,
                                          C.byref(ni), F)
    return rc, np.array(R[:]).reshape(3, 3), np.array(T[:]), ni.value, np.array(F[:]).reshape(3, 3)
