"""The reference's dense single-scale path composed from oracle calls (numpy), plus the
synthetic inputs shared by the CPU and GPU tests.  Test infrastructure."""
import math

import numpy as np

from tests import oracle as orc


def middle_index(hWin, wWin):
    # radial/radial_opticalflow_groundtruth.lua:91
    return math.ceil(wWin / 2) + wWin * (math.ceil(hWin / 2) - 1)


def pad_output(a, wWin, hWin, wKer, hKer):
    # radial/radial_opticalflow_groundtruth.lua:23-35
    l = (wWin - 1) // 2 + (wKer - 1) // 2
    r = math.ceil((wWin - 1) / 2) + math.ceil((wKer - 1) / 2)
    t = (hWin - 1) // 2 + (hKer - 1) // 2
    b = math.ceil((hWin - 1) / 2) + math.ceil((hKer - 1) / 2)
    pad = [(0, 0)] * (a.ndim - 2) + [(t, b), (l, r)]
    return np.pad(a, pad)


def adapt_mask(hWin, wWin, hKer, wKer, mask):
    # radial/radial_opticalflow_groundtruth.lua:37-63
    h, w = mask.shape
    new = np.zeros_like(mask)
    ls = (wWin - 1) // 2 + (wKer - 1) // 2
    if ls > 0:
        new[:, ls:] += mask[:, : w - ls]
    rs = math.ceil((wWin - 1) / 2) + math.ceil((wKer - 1) / 2)
    if rs > 0:
        new[:, : w - rs] += mask[:, rs:]
    ts = (hWin - 1) // 2 + (hKer - 1) // 2
    if ts > 0:
        new[ts:, :] += mask[: h - ts, :]
    bs = math.ceil((hWin - 1) / 2) + math.ceil((hKer - 1) / 2)
    if bs > 0:
        new[: h - bs, :] += mask[bs:, :]
    return (new > 3.9).astype(np.float32)


def dense_flow_oracle(img1, img2, hWin, wWin, hKer, wKer, thr=0.21, via_unfold=False):
    """compute_cartesian_groundtruth_cross_correlation (radial/radial_opticalflow_groundtruth.lua:66-112)
    on the oracle. Returns dict(cost, idx, best, fy, fx, scores, imaxs, flowp)."""
    if via_unfold:  # the literal reference composition: unfold -> crop -> SpatialMatching
        u1, u2 = orc.unfold(img1, hKer, wKer), orc.unfold(img2, hKer, wKer)
        t, b = (hWin - 1) // 2, math.ceil((hWin - 1) / 2)
        l, r = (wWin - 1) // 2, math.ceil((wWin - 1) / 2)
        u1c = np.ascontiguousarray(u1[:, t : u1.shape[1] - b, l : u1.shape[2] - r])
        cost = orc.spatial_matching(u1c, u2, hWin, wWin)
    else:
        cost = orc.ssd_cost_volume(img1, img2, hKer, wKer, hWin, wWin)
    Ho, Wo = cost.shape[:2]
    vol = cost.reshape(Ho, Wo, hWin * wWin)
    idx, best = orc.argbest_center(vol, middle_index(hWin, wWin), take_max=False)
    y, x = orc.x2yx(idx, hWin, wWin)
    scores = np.zeros((Ho, Wo), np.float32)
    imaxs = np.zeros((Ho, Wo), np.int64)
    orc.extract_output(vol, thr, imaxs, scores)
    H, W = img1.shape[1:]
    mask = adapt_mask(hWin, wWin, hKer, wKer, np.ones((H, W), np.float32))
    flow = np.stack([y.astype(np.float32), x.astype(np.float32), np.ones((Ho, Wo), np.float32), scores])
    flowp = pad_output(flow, wWin, hWin, wKer, hKer)
    flowp[2] *= mask
    return dict(cost=cost, idx=idx, best=best, fy=y, fx=x, scores=scores, imaxs=imaxs, flowp=flowp)


def warp_nearest_offset(im2, flow):
    """image.warp(im2, flow, 'nearest', true): im1(y,x) = im2(y+flow[0], x+flow[1])
    (radial/radial_opticalflow_groundtruth.lua:172-175); out-of-frame samples clamp."""
    C, H, W = im2.shape
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    sy = np.clip(yy + flow[0].astype(np.int64), 0, H - 1)
    sx = np.clip(xx + flow[1].astype(np.int64), 0, W - 1)
    return im2[:, sy, sx]


def kat_case(case, seed=0, C=30, h=32, w=42):
    """The four cases of cartesian_groundtruth_cc_testme (radial/radial_opticalflow_groundtruth.lua:170-210).
    Returns (im1, im2, flowbase, (hKer, wKer, hWin, wWin))."""
    rng = np.random.default_rng(seed + 101 * case)
    im2 = rng.random((C, h, w), dtype=np.float32)
    if case == 0:
        geo = (1, 1, 12, 15)
        fb = np.floor(rng.random((2, h, w)) * 12 - 5)
        noise = 0.0
    elif case == 1:
        geo = (1, 1, 17, 15)
        fb = np.floor(rng.random((2, h, w)) * 15 - 7)
        noise = 0.0
    elif case == 2:
        geo = (3, 3, 17, 17)
        fb = np.empty((2, h, w))
        fb[0] = math.floor(rng.random() * 17 - 8 + 0.5)
        fb[1] = math.floor(rng.random() * 17 - 8 + 0.5)
        noise = 0.5
    else:
        geo = (5, 5, 17, 17)
        fb = np.empty((2, h, w))
        fb[0] = math.floor(rng.random() * 17 - 8 + 0.5)
        fb[1] = math.floor(rng.random() * 17 - 8 + 0.5)
        noise = 1.0
    fb = fb.astype(np.float32)
    im1 = warp_nearest_offset(im2, fb)
    if noise:
        im1 = im1 + (rng.standard_normal(im1.shape) * noise).astype(np.float32)
    return np.ascontiguousarray(im1, np.float32), im2, fb, geo


def synth_pair(H, W, C=3, seed=0, max_flow=12, integer=True, noise_sigma=2.0):
    """Synthetic frame pair of SURVEY 8(d): box-smoothed uint8 noise, planted radial integer flow
    from an off-centre focus of expansion, re-quantised noise.  frame0(p) = frame1(p + flow(p))."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(C, H + 4, W + 4)).astype(np.float64)
    sm = np.zeros((C, H, W))
    for i in range(5):
        for j in range(5):
            sm += base[:, i : i + H, j : j + W]
    frame1 = np.round(sm / 25.0)
    cx, cy = W / 2 + 17, H / 2 - 9
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    s = max_flow / max(cx, W - cx, cy, H - cy)
    flow = np.stack([np.round(s * (yy - cy)), np.round(s * (xx - cx))]).astype(np.float32)
    frame0 = warp_nearest_offset(frame1, flow)
    if noise_sigma:
        frame0 = np.clip(np.round(frame0 + rng.standard_normal(frame0.shape) * noise_sigma), 0, 255)
    f0 = np.ascontiguousarray(frame0, np.float32)
    f1 = np.ascontiguousarray(frame1, np.float32)
    if not integer:
        f0, f1 = f0 / np.float32(255), f1 / np.float32(255)
    return f0, f1, flow, (cx, cy)


def multiscale_flow_oracle(f0, f1, k, maxh, maxw, ratios, f16_scale=None):
    """getModelMultiscale(...):forward + processOutput ('max', no threshold) for the identity patch filter, on the oracle:
    per ratio pyramid_scale_volume (A2) -> softmin over the window (A3) -> cascade + ring extraction (A4 + A5) -> arg-max
    with the centre tie-break (A6) -> x2yxMulti decode (A10).  opticalflow_model_multiscale.lua:134-333,
    opticalflow_model.lua:153-161,201-252.  Returns dict(joined, idx, y, x, middle)."""
    from tests import oracle as orc

    H, W = f0.shape[1:]
    vols = [orc.pyramid_scale_volume(f0, f1, r, k, k, maxh, maxw) for r in ratios]
    if f16_scale:   # fp16 volumes (BASELINE configs[4]): the fp32 chain with half() applied where the volume is stored
        sc, inv = np.float32(f16_scale), np.float32(1.0) / np.float32(f16_scale)
        vols = [(v * sc).astype(np.float16).astype(np.float32) * inv for v in vols]
    probs = [orc.softmin(v.reshape(-1, maxh * maxw)).reshape(v.shape) for v in vols]
    rc, joined = orc.cascade_ring(probs, ratios, H, W, maxh, maxw)
    assert rc == 0
    middle = orc.yx2x_multi(maxh, maxw, ratios, 0, 0)
    idx, best = orc.argbest_center(joined, middle, True)
    rc, y, x = orc.x2yx_multi(maxh, maxw, ratios, idx)
    assert rc == 0
    return dict(joined=joined, idx=idx, best=best, y=y, x=x, middle=middle, vols=vols)


def filter_stack_oracle(x, layers):
    """getFilter(geometry):forward (opticalflow_model.lua:45-79) on the oracle.  layers: list of dicts
    {weight, bias, conn (None or [nConn][2] int32, 1-based), nOut, tanh}."""
    from tests import oracle as orc

    for L in layers:
        if L.get("conn") is not None:
            x = orc.spatial_convolution_map(x, L["weight"], L["bias"], L["conn"], L["nOut"])
        else:
            x = orc.spatial_convolution(x, L["weight"], L["bias"])
        if L.get("tanh"):
            x = orc.tanh(x)
    return x


def multiscale_filtered_oracle(f0, f1, stacks, maxh, maxw, ratios, f16_scale=None):
    """getModelMultiscale(geometry, true, false):forward({I0, I1}) + processOutput ('max') WITH the learned filters, on the oracle
    (opticalflow_model_multiscale.lua:134-173,196-229): per ratio r: SpatialDownSampling(r) -> SpatialZeroPadding(hPatch2-1 /
    wPatch2-1 split floor / ceil) of both frames; frame 0 cropped by maxh-1 / maxw-1 (filter1, :198-202); getFilter on both;
    SpatialMatching(maxh, maxw); softmin; cascade + ring; arg-max with the centre tie-break; decode.
    stacks: one list of layer dicts per ratio (the same list object when the filters are shared)."""
    from tests import oracle as orc

    H, W = f0.shape[1:]
    hk = 1 + sum(L["weight"].shape[-2] - 1 for L in stacks[0])
    wk = 1 + sum(L["weight"].shape[-1] - 1 for L in stacks[0])
    hp, wp = maxh - 1 + hk - 1, maxw - 1 + wk - 1
    ct, cl = (maxh - 1) // 2, (maxw - 1) // 2
    vols, feats = [], []
    for r, st in zip(ratios, stacks):
        d0, d1 = (orc.downsample_box(f, r) if r > 1 else np.ascontiguousarray(f, np.float32) for f in (f0, f1))
        p0 = orc.zero_pad(d0, wp // 2, wp - wp // 2, hp // 2, hp - hp // 2)
        p1 = orc.zero_pad(d1, wp // 2, wp - wp // 2, hp // 2, hp - hp // 2)
        p0c = np.ascontiguousarray(p0[:, ct : p0.shape[1] - (maxh - 1 - ct), cl : p0.shape[2] - (maxw - 1 - cl)])
        a, b = filter_stack_oracle(p0c, st), filter_stack_oracle(p1, st)
        assert a.shape[1:] == (H // r, W // r) and b.shape[1:] == (H // r + maxh - 1, W // r + maxw - 1)
        feats.append((a, b))
        vols.append(orc.spatial_matching(a, b, maxh, maxw))
    if f16_scale:
        sc, inv = np.float32(f16_scale), np.float32(1.0) / np.float32(f16_scale)
        vols = [(v * sc).astype(np.float16).astype(np.float32) * inv for v in vols]
    probs = [orc.softmin(v.reshape(-1, maxh * maxw)).reshape(v.shape) for v in vols]
    rc, joined = orc.cascade_ring(probs, ratios, H, W, maxh, maxw)
    assert rc == 0
    middle = orc.yx2x_multi(maxh, maxw, ratios, 0, 0)
    idx, best = orc.argbest_center(joined, middle, True)
    rc, y, x = orc.x2yx_multi(maxh, maxw, ratios, idx)
    assert rc == 0
    return dict(joined=joined, idx=idx, best=best, y=y, x=x, middle=middle, vols=vols, feats=feats)


def radial_path_oracle(prev_img, img, e2, networkp, w1, b1, w2, b2, tanh_between=False, kinfty=0.65, alpha=1.0, zero_last_row=False):
    """radial/test_radial_opticalflow.lua:186-225 on the oracle, for the separable filter stack conv(1 x kW) [tanh] conv(kH x 1):
    getC2PMask(+ wrap columns) -> cartesian2polar of both frames -> getTesterNetwork (crop hWin-1 rows of the previous frame,
    shared filter, SpatialRadialMatching) -> min(3) - 1 (test_radial:204-207; the last row zeroed only with zero_last_row, which is
    what the trainer's display code does, train_radial:178-180) -> getP2CMaskOF ->
    cartesian2polar of the flow -> flow2depth(center = e2 * getKOutput).  Returns dict(polar_prev, polar_img, feat1, feat2,
    output, polar_flow, flow, depth, confs)."""
    import math

    import numpy as np

    from tests import oracle as orc

    Cc, hImg, wImg = img.shape
    hIn, wIn, hWin = networkp["hInput"], networkp["wInput"], networkp["hWin"]
    kW, kH = w1.shape[3], w2.shape[2]
    ex, ey = float(e2[0]), float(e2[1])
    rmax = math.floor(math.sqrt(max(max(ex * ex + ey * ey, (wImg - ex) ** 2 + ey * ey), max(ex * ex + (hImg - ey) ** 2, (wImg - ex) ** 2 + (hImg - ey) ** 2))))
    mask = orc.polar_grid_c2p(wImg, hImg, wIn, hIn, ex, ey, (kW - 1) // 2, -(-(kW - 1) // 2), rmax, alpha)
    pp, pi = orc.warp_bilinear(prev_img, mask), orc.warp_bilinear(img, mask)

    def filt(x):
        t = orc.spatial_convolution(x, w1, b1)
        if tanh_between:
            t = orc.tanh(t)
        return orc.spatial_convolution(t, w2, b2)

    f1, f2 = filt(np.ascontiguousarray(pp[:, : hIn - hWin + 1])), filt(pi)      # SpatialPadding(0,0,0,-hWin+1) on the previous frame
    out = orc.radial_matching(f1, f2, hWin)
    pf = out.argmin(2).astype(np.float32)                                       # numpy argmin: first minimum
    if zero_last_row:
        pf[-1] = 0
    hPolar = hIn - kH - hWin + 2
    assert hPolar == out.shape[0]
    kOut = hPolar / hIn
    hOut, wOut = int(hImg * kOut), int(wImg * kOut)
    p2c = orc.polar_grid_p2c(wIn, hPolar, wOut, hOut, ex * kOut, ey * kOut, rmax * kOut, alpha)
    cart = orc.warp_bilinear(pf[None], p2c)[0]
    kOut2 = (hIn - (kH - 1) // 2 - hWin + 1) / hIn
    cx, cy = ex * kOut2, ey * kOut2
    infty = math.floor(math.sqrt(max(max(cx * cx + cy * cy, (wImg - cx) ** 2 + cy * cy), max(cx * cx + (hImg - cy) ** 2, (wImg - cx) ** 2 + (hImg - cy) ** 2)))) * kinfty
    depth, conf = orc.flow_to_depth_radial(cart, cx, cy, infty)
    return dict(polar_prev=pp, polar_img=pi, feat1=f1, feat2=f2, output=out, polar_flow=pf, flow=cart, depth=depth, confs=conf)


def version2_flow_oracle(prev, cur, datap, weights, biases, threshold=1e-4, thresval=1e-4):
    """version2/test.lua:43-51 on the oracle: getNetwork(datap):forward({prev, cur}) (version2/network.lua:5-39: contrastive
    normalisation of both frames with gaussian1D(normalization_k), the first branch cropped by the window, the shared convolution
    stack, SpatialMatching(hWin, wWin)) and the decode (first minimum over the window, yflow = idx // wWin - tWin,
    xflow = idx % wWin - lWin).  weights[i] [nOut][nIn][kH][kW], biases[i] [nOut]."""
    import math
    from tests import oracle as orc

    k = datap["normalization_k"]
    i = np.arange(1, k + 1, dtype=np.float64)
    g = np.exp(-(((i - (k / 2 + 0.5)) / (0.25 * k)) ** 2) / 2).astype(np.float32)      # image.gaussian1D(k): sigma 0.25, amplitude 1
    n0 = orc.contrastive_normalization(prev, g, threshold, thresval)
    n1 = orc.contrastive_normalization(cur, g, threshold, thresval)
    hWin, wWin = datap["hWin"], datap["wWin"]
    lWin, tWin = math.ceil(wWin / 2) - 1, math.ceil(hWin / 2) - 1
    H, W = prev.shape[1:]
    a = np.ascontiguousarray(n0[:, tWin : tWin + H - (hWin - 1), lWin : lWin + W - (wWin - 1)])
    b = n1
    for w, bb in zip(weights, biases):
        a = orc.spatial_convolution(a, w, bb)
        b = orc.spatial_convolution(b, w, bb)
    vol = orc.spatial_matching(a, b, hWin, wWin)
    H1, W1 = vol.shape[:2]
    idx0 = vol.reshape(H1, W1, -1).argmin(axis=2)               # numpy: the first minimum
    return {"volume": vol, "index": idx0 + 1, "yflow": (idx0 // wWin - tWin).astype(np.float32), "xflow": (idx0 % wWin - lWin).astype(np.float32)}


def version2_band_oracle(prev, cur, datap, weights, biases, y0, y1, threshold=1e-4, thresval=1e-4):
    """version2_flow_oracle for the output rows y0 .. y1-1 only: the normalisation on the whole frames (its estimators see the frame's
    borders), then the convolution stack and the matcher on the rows the band needs -- both are 'valid' operators, so the band's values
    are exactly the full frame's.  Returns the same keys, [y1 - y0][W1]..."""
    import math
    from tests import oracle as orc

    k = datap["normalization_k"]
    i = np.arange(1, k + 1, dtype=np.float64)
    g = np.exp(-(((i - (k / 2 + 0.5)) / (0.25 * k)) ** 2) / 2).astype(np.float32)
    n0 = orc.contrastive_normalization(prev, g, threshold, thresval)
    n1 = orc.contrastive_normalization(cur, g, threshold, thresval)
    hWin, wWin = datap["hWin"], datap["wWin"]
    lWin, tWin = math.ceil(wWin / 2) - 1, math.ceil(hWin / 2) - 1
    H, W = prev.shape[1:]
    hk = 1 + sum(w.shape[2] - 1 for w in weights)
    a = np.ascontiguousarray(n0[:, tWin + y0 : tWin + y1 + hk - 1, lWin : lWin + W - (wWin - 1)])
    b = np.ascontiguousarray(n1[:, y0 : y1 + hk - 1 + hWin - 1])
    for w, bb in zip(weights, biases):
        a = orc.spatial_convolution(a, w, bb)
        b = orc.spatial_convolution(b, w, bb)
    assert a.shape[1] == y1 - y0 and b.shape[1] == y1 - y0 + hWin - 1
    vol = orc.spatial_matching(a, b, hWin, wWin)
    idx0 = vol.reshape(vol.shape[0], vol.shape[1], -1).argmin(axis=2)
    return {"volume": vol, "index": idx0 + 1, "yflow": (idx0 // wWin - tWin).astype(np.float32), "xflow": (idx0 % wWin - lWin).astype(np.float32)}


def single_scale_flow_oracle(f0, f1, layers, maxh, maxw, hImg, wImg, threshold=None):
    """depth_estimation_opticalflow.lua:66-116 for a single-scale model, on the oracle: filter:forward of both frames (layers: list of
    dicts as filter_stack_oracle takes them; empty = the inputs are feature maps already), prepareInput's narrow of patch 1
    (opticalflow_model.lua:147-149), getModel(geometry, true, true):forward = SpatialMatching -> Minus -> SoftMax over the window
    (:81-129), processOutput(geometry, out, true, threshold) (:201-252): the first maximum with the centre tie-break, or
    extractOutput(p, 0.11) and scores > threshold (imaxs = centre, scores = 0 where nothing exceeds 0.11: SURVEY appendix A); the
    decode x2yx - centered2onebased(0, 0); the centre paste.  Returns dict(volume, prob, index, scores, y, x, confidences, full,
    full_confidences)."""
    import math
    from tests import oracle as orc

    a = filter_stack_oracle(f0, layers) if layers else np.ascontiguousarray(f0, np.float32)
    b = filter_stack_oracle(f1, layers) if layers else np.ascontiguousarray(f1, np.float32)
    y0, x0 = math.ceil(maxh / 2) - 1, math.ceil(maxw / 2) - 1
    a = np.ascontiguousarray(a[:, y0 : y0 + a.shape[1] - maxh + 1, x0 : x0 + a.shape[2] - maxw + 1])
    vol = orc.spatial_matching(a, b, maxh, maxw)
    H1, W1 = vol.shape[:2]
    N = maxh * maxw
    prob = orc.softmin(vol.reshape(-1, N)).reshape(H1, W1, N)
    middle = math.ceil(maxw / 2) + maxw * (math.ceil(maxh / 2) - 1)
    scores = np.zeros((H1, W1), np.float32)
    if threshold is None:
        idx, _ = orc.argbest_center(prob, middle, True)
        conf = np.ones((H1, W1), np.float32)
    else:
        idx = np.full((H1, W1), middle, np.int64)
        orc.extract_output(prob, 0.11, idx, scores)
        conf = (scores > np.float32(threshold)).astype(np.float32)
    y = (idx - 1) // maxw + 1 - math.ceil(maxh / 2)
    x = (idx - 1) % maxw + 1 - math.ceil(maxw / 2)
    ho, wo = (hImg - H1) // 2, (wImg - W1) // 2
    full = np.zeros((2, hImg, wImg), np.float32)
    full[0, ho : ho + H1, wo : wo + W1] = y
    full[1, ho : ho + H1, wo : wo + W1] = x
    fc = np.zeros((hImg, wImg), np.float32)
    fc[ho : ho + H1, wo : wo + W1] = conf
    return dict(volume=vol, prob=prob, index=idx, scores=scores, y=y, x=x, confidences=conf, full=full, full_confidences=fc)
