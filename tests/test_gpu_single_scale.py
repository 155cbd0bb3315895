"""GPU suite (-m gpu), part 4: the single-scale trained model as a pipeline -- what depth_estimation_opticalflow.lua:66-116 runs per frame
pair for a model that is not multiscale: filter:forward of both frames (getFilter, opticalflow_model.lua:45-79), prepareInput's narrow
(:131-151), getModel(geometry, true, prefiltered):forward = SpatialMatching -> Minus -> SoftMax over the window (:81-129) and
processOutput(geometry, moutput, true, threshold) (:201-252).

One call (dfe_flow_pair_filtered_f32; with 16- / 17-wide windows the matcher's soft-max epilogue, no volume) == the module path bit for
bit (index, scores, y, x, confidences, the centre-pasted full planes); both against the ORACLE composition (tests/refpath.
single_scale_flow_oracle on the device's feature maps: costs bit-exact, probabilities within N * 2^-24 -- the oracle adds a window's N
exponentials one after the other in fp32, the device as 16 partial sums and a tree, and either sum is exact to N / 2 roundings at most; the
device's exp differs from glibc's in the last place -- indices tie-aware, extractOutput scores within the probabilities' tolerance)."""
import math

import numpy as np
import pytest
import torch

from tests import oracle as orc
from tests import refpath as rp

pytestmark = pytest.mark.gpu

SOFT_ATOL = 1e-6            # windows of 64 cells (the multiscale model); windows of N > 64 cells: N * 2^-24, see soft_tol


def soft_tol(N):
    return max(SOFT_ATOL, N * 2.0 ** -24)
TM_LAYERS = [(3, 5, 5, 4), (4, 5, 5, 4), (4, 5, 5, 10)]          # tests/time_matching.lua:13


def T(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _pair(H, W, seed, gain, flat=True):
    """a smooth random frame and its shifted, slightly noisy copy; `gain` scales the contrast (small gain -> costs of the order of one ->
    several window cells above extractOutput's 0.11); a constant block (flat) makes every cost of its pixels equal: the centre tie-break"""
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=seed, max_flow=5, noise_sigma=1.0)
    f0, f1 = f0 / np.float32(255) * np.float32(gain), f1 / np.float32(255) * np.float32(gain)
    if flat:
        f0[:, H // 3 : H // 3 + 30, W // 4 : W // 4 + 60] = 0.25 * gain
        f1[:, max(H // 3 - 20, 0) : H // 3 + 50, W // 4 - 20 : W // 4 + 80] = 0.25 * gain
    return f0, f1


def _same(one, stg, threshold):
    for k in ("index", "y", "x", "full", "full_confidences"):
        assert torch.equal(one[k], stg[k]), k
    assert torch.equal(one["confidences"].to(torch.float32), stg["confidences"].to(torch.float32))
    if threshold is not None and "scores" in stg:
        assert torch.equal(one["scores"], stg["scores"])


@pytest.mark.parametrize("threshold", [None, 0.3, 3.0])
@pytest.mark.parametrize("H,W,layers,mh,mw,gain,wgain", [
    (64, 300, TM_LAYERS, 16, 16, 1.0, 100.0),         # time_matching.lua's stack and window: the flat matcher's soft-max epilogue, 16 x 16
    (64, 300, TM_LAYERS, 16, 16, 1.0, 1.0),           # ... with the initial (small) weights: probabilities all close to 1 / 256
    (50, 330, [(3, 9, 9, 8)], 17, 17, 30.0, 1.0),     # 17 x 17 (the extra-row task), one layer
    (44, 301, [(3, 5, 5, 6)], 12, 17, 30.0, 1.0),     # 12 rows of a 17-wide window, W1 % 4 = 1
    (40, 290, [(3, 3, 3, 5)], 6, 16, 30.0, 1.0),      # a window of fewer rows than a DPP row has lanes
    (48, 64, [(3, 5, 5, 4), (4, 3, 3, 6)], 9, 9, 1.0, 30.0),   # other windows: the stand-alone ops, same results
])
def test_single_scale_one_call_equals_staged_and_oracle(dfe, cuda, H, W, layers, mh, mw, gain, wgain, threshold):
    """(gain / wgain: contrast of the frames / a factor on the last layer's weights, chosen so that 30 .. 50 % of the pixels have SEVERAL
    probabilities above extractOutput's 0.11 and the scores spread over 0 .. 8: freshly initialised stacks give near-uniform windows)"""
    gen = torch.Generator().manual_seed(H + mh)
    geo = dict(layers=[list(l) for l in layers], maxh=mh, maxw=mw, multiscale=False, output_extraction_method="max", hImg=H, wImg=W)
    model = dfe.getModel(geo, True, False, device=cuda, generator=gen)
    [m for m in model.modules[0].modules[0].modules if getattr(m, "weight", None) is not None][-1].weight.mul_(wgain)
    f0, f1 = _pair(H, W, H + W, gain)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    ctx = dfe.get_ctx(0)
    one = model.forwardFlow([t0, t1], threshold, one_call=True)
    hk, wk = 1 + sum(l[2] - 1 for l in layers), 1 + sum(l[1] - 1 for l in layers)
    W1 = W - wk + 1 - mw + 1
    if mw in (16, 17) and W1 >= 253:
        assert ctx.last_kernel() == "feat_matching_flat_kernel+softmax", ctx.last_kernel()
    stg = model.forwardFlow([t0, t1], threshold, one_call=False)
    if threshold is not None:
        # (the staged table carries no scores: recompute them the way getOutputConfidences does)
        prob = model.modules[-1].output
        sc = torch.zeros(prob.shape[:2], device=cuda)
        im = torch.full(prob.shape[:2], dfe.getMiddleIndex(geo), dtype=torch.int64, device=cuda)
        dfe.extractoutput.extractOutput(prob, sc, 0.11, im)
        stg["scores"] = sc
        assert torch.equal(im, stg["index"])
    _same(one, stg, threshold)
    # ---- the oracle, on the device's feature maps
    feat0, feat1 = model.modules[0].modules[0].output.cpu().numpy(), model.modules[0].modules[1].output.cpu().numpy()
    ref = rp.single_scale_flow_oracle(feat0, feat1, [], mh, mw, H, W, threshold)
    N = mh * mw
    TOL = soft_tol(N)
    prob = model.modules[-1].output.cpu().numpy()
    assert prob.shape == ref["prob"].shape
    assert np.array_equal(model.modules[1].output.cpu().numpy(), ref["volume"])
    assert np.abs(prob - ref["prob"]).max() <= TOL
    gi = one["index"].cpu().numpy()
    if threshold is None:
        srt = np.sort(ref["prob"], axis=2)
        near = srt[..., -1] - srt[..., -2] <= 4 * TOL               # the two best within the tolerance: either may win
        mid = dfe.getMiddleIndex(geo)
        centre_near = np.abs(ref["prob"][..., mid - 1] - srt[..., -1]) <= 4 * TOL
        ok = (gi == ref["index"]) | near | centre_near
        assert ok.all(), "%d pixels differ from the oracle outside ties" % int((~ok).sum())
        assert (gi != ref["index"]).mean() < 0.02
        # the flat block: every cost equal -> every probability equal -> the first maximum is cell 1, the tie-break makes it the centre class
        flatpix = ref["volume"].reshape(gi.shape + (-1,)).max(axis=2) == ref["volume"].reshape(gi.shape + (-1,)).min(axis=2)
        if mw in (16, 17) and mh >= 12:
            assert flatpix.sum() >= 20
        assert (gi[flatpix] == mid).all()
        assert float(one["full_confidences"].sum()) == gi.size
    else:
        gs = one["scores"].cpu().numpy()
        # scores are sums of up to 36 probabilities (8 prefix sums of 8 values); a probability within TOL of 0.11 may fall on either side
        edge = (np.abs(ref["prob"] - 0.11) <= 2 * TOL).any(axis=2)
        assert np.abs(gs - ref["scores"])[~edge].max() <= 40 * TOL
        srt = np.sort(ref["prob"], axis=2)
        near = srt[..., -1] - srt[..., -2] <= 4 * TOL
        ok = (gi == ref["index"]) | near | edge
        assert ok.all(), "%d pixels differ from the oracle outside ties" % int((~ok).sum())
        conf_edge = np.abs(ref["scores"] - np.float32(threshold)) <= 40 * TOL
        gc = one["confidences"].cpu().numpy()
        assert ((gc == ref["confidences"]) | conf_edge | edge).all()
        if gain * wgain > 1:
            assert ((ref["prob"] > 0.11).sum(axis=2) >= 2).mean() > 0.2, "the case is meant to have several candidates per pixel"
            assert 0.02 < gc.mean() < 0.98
    # decode and paste are exact functions of the index
    y = (gi - 1) // mw + 1 - math.ceil(mh / 2)
    x = (gi - 1) % mw + 1 - math.ceil(mw / 2)
    assert np.array_equal(one["y"].cpu().numpy(), y) and np.array_equal(one["x"].cpu().numpy(), x)
    H1 = gi.shape[0]
    ho, wo = (H - H1) // 2, (W - gi.shape[1]) // 2
    full = one["full"].cpu().numpy()
    assert np.array_equal(full[0, ho : ho + H1, wo : wo + gi.shape[1]], y.astype(np.float32))
    full[:, ho : ho + H1, wo : wo + gi.shape[1]] = 0
    assert not full.any(), "the planes are zero outside the pasted region"


@pytest.mark.parametrize("threshold", [None, 0.4])
def test_single_scale_prefiltered_pair_reads_the_narrow_in_place(dfe, cuda, threshold):
    """geometry.prefilter (the script's own mode: loadModel(path, true, true), each frame filtered once and kept for the next pair):
    dfe_flow_pair_filtered_f32 with no layers takes the two FEATURE maps and reads patch 1's narrow as a view; the module path gets the
    same view from prepareInput and nn.SpatialMatching reads it in place (dfe_spatial_matching_strided_f32) -- equal to the contiguous
    copy's result and to the one call, bit for bit."""
    gen = torch.Generator().manual_seed(3)
    H, W, mh, mw = 70, 310, 16, 16
    geo = dict(layers=[list(l) for l in TM_LAYERS], maxh=mh, maxw=mw, multiscale=False, output_extraction_method="max", hImg=H, wImg=W, prefilter=True)
    filt = dfe.getFilter(geo, device=cuda, generator=gen)
    [m for m in filt.modules if getattr(m, "weight", None) is not None][-1].weight.mul_(100.0)
    f0, f1 = _pair(H, W, 11, 1.0)
    a, b = filt.forward(T(f0, cuda)).clone(), filt.forward(T(f1, cuda)).clone()
    model = dfe.getModel(geo, True, True)
    ctx = dfe.get_ctx(0)
    p1, p2 = dfe.prepareInput(geo, a, b)
    assert not p1.is_contiguous() and p1.data_ptr() != a.data_ptr()
    vol_view = dfe.nn.SpatialMatching(mh, mw, False).forward([p1, p2]).clone()
    assert ctx.last_kernel() == "feat_matching_flat_kernel"
    vol_copy = dfe.nn.SpatialMatching(mh, mw, False).forward([p1.contiguous(), p2])
    assert torch.equal(vol_view, vol_copy)
    with ctx.options(fm_flat=0):                                          # the kernels that need a contiguous map: the entry copies
        vol_old = dfe.nn.SpatialMatching(mh, mw, False).forward([p1, p2])
        assert ctx.last_kernel() != "feat_matching_flat_kernel"
    assert torch.equal(vol_old, vol_copy)
    assert np.array_equal(vol_view.cpu().numpy(), orc.spatial_matching(p1.contiguous().cpu().numpy(), p2.cpu().numpy(), mh, mw))
    one = model.forwardFlow([a, b], threshold, one_call=True)
    assert ctx.last_kernel() == "feat_matching_flat_kernel+softmax"
    stg = model.forwardFlow([a, b], threshold, one_call=False)
    _same(one, stg, threshold)
    # a narrow that ends at the end of its parent's allocation (maxh = 1-row margin cases are the clamped staging path's): the last rows
    last = one["index"][-2:].cpu().numpy()
    ref = rp.single_scale_flow_oracle(a.cpu().numpy(), b.cpu().numpy(), [], mh, mw, H, W, threshold)
    srt = np.sort(ref["prob"], axis=2)
    near = (srt[..., -1] - srt[..., -2] <= 4 * soft_tol(mh * mw)) | (np.abs(ref["prob"] - 0.11) <= 2 * soft_tol(mh * mw)).any(axis=2)
    assert ((last == ref["index"][-2:]) | near[-2:]).all()


def test_single_scale_vga_learned_workload_one_call_equals_staged(dfe, cuda):
    """The bench workload `vga-learned` as it runs: 640 x 480 frames, time_matching.lua's stack, 16 x 16 window (465 x 625 outputs, 1141
    matcher tiles = 4.5 rounds), without and with a threshold: one call == the module path bit for bit; the planted flow is found; the
    oracle composition on the last output rows (the narrow's last rows, the last tiles of the last round)."""
    H, W, mh, mw = 480, 640, 16, 16
    gen = torch.Generator().manual_seed(1)
    geo = dict(layers=[list(l) for l in TM_LAYERS], maxh=mh, maxw=mw, multiscale=False, output_extraction_method="max", hImg=H, wImg=W)
    model = dfe.getModel(geo, True, False, device=cuda, generator=gen)
    f0, f1, flow, _ = rp.synth_pair(H, W, C=3, seed=2, max_flow=6, noise_sigma=0)
    t0, t1 = T(f0 / np.float32(255), cuda), T(f1 / np.float32(255), cuda)
    ctx = dfe.get_ctx(0)
    for threshold in (None, 0.11):
        one = model.forwardFlow([t0, t1], threshold, one_call=True)
        assert ctx.last_kernel() == "feat_matching_flat_kernel+softmax"
        stg = model.forwardFlow([t0, t1], threshold, one_call=False)
        _same(one, stg, threshold)
    assert tuple(one["index"].shape) == (465 - 12, 625 - 12)
    feat0, feat1 = model.modules[0].modules[0].output, model.modules[0].modules[1].output
    rows = 6
    a = feat0[:, -(rows + mh - 1) :].cpu().numpy()                       # the rows of patch 1 whose narrow gives the last `rows` output rows
    b = feat1[:, -(rows + mh - 1) :].cpu().numpy()
    ref = rp.single_scale_flow_oracle(a, b, [], mh, mw, rows, one["index"].shape[1], 0.11)
    gi = one["index"][-rows:].cpu().numpy()
    srt = np.sort(ref["prob"], axis=2)
    near = (srt[..., -1] - srt[..., -2] <= 4 * soft_tol(mh * mw)) | (np.abs(ref["prob"] - 0.11) <= 2 * soft_tol(mh * mw)).any(axis=2)
    assert ((gi == ref["index"]) | near).all()


def test_sequential_runs_convolution_and_tanh_as_one_launch(dfe, cuda):
    """getFilter's nn.SpatialConvolution -> nn.Tanh pairs (opticalflow_model.lua:48-64) are one launch in nn.Sequential
    (dfe_spatial_convolution_tanh_f32: the convolution kernel's epilogue applies the tanh): same features and same gradients as the
    module-by-module evaluation (fuse = False), bit for bit; the fused pair leaves the convolution's .output unset."""
    gen = torch.Generator().manual_seed(5)
    geo = dict(layers=[list(l) for l in TM_LAYERS], maxh=16, maxw=16, multiscale=False)
    filt = dfe.getFilter(geo, device=cuda, generator=gen)
    x = torch.randn((3, 50, 70), generator=gen).to(cuda)
    go = torch.randn((10, 38, 58), generator=gen).to(cuda)
    res = {}
    for fuse in (True, False):
        filt.fuse = fuse
        filt.zeroGradParameters()
        out = filt.forward(x)
        assert (filt.modules[0].output is None) == fuse and filt.modules[1].output is not None
        gi = filt.backward(x, go)
        res[fuse] = (out.clone(), gi.clone(), [g.clone() for g in filt.parameters()[1]])
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert all(torch.equal(a, b) for a, b in zip(res[True][2], res[False][2]))
    layers = [dict(weight=m.weight.cpu().numpy(), bias=m.bias.cpu().numpy(), tanh=i < 2) for i, m in enumerate(filt.modules[::2])]
    assert np.allclose(res[True][0].cpu().numpy(), rp.filter_stack_oracle(x.cpu().numpy(), layers), rtol=0, atol=2e-6)


def test_single_scale_unique_minimum_fast_path_and_its_ties(dfe, cuda):
    """The soft-max epilogue without a threshold skips the exponentials for windows whose minimum is alone within 1e-6 (FF_TIE): planted
    windows with TWO exactly equal minima (a feature patch repeated at two displacements), with a second cell a few 1e-7 above the minimum
    (inside the band: the full arithmetic decides) and a few 1e-5 above it (outside: the fast path), all inside waves whose other windows
    are ordinary -- index, flow and full planes equal the staged modules' bit for bit, and the volume says which case each pixel is."""
    rng = np.random.default_rng(17)
    K, mh, mw, H1, W1 = 6, 16, 16, 40, 300
    H2, W2 = H1 + mh - 1, W1 + mw - 1
    geo = dict(layers=[[3, 1, 1, K]], maxh=mh, maxw=mw, multiscale=False, output_extraction_method="max", hImg=H2, wImg=W2, prefilter=True)
    b = (rng.standard_normal((K, H2, W2)) * 2).astype(np.float32)
    a_full = (rng.standard_normal((K, H2, W2)) * 2).astype(np.float32)
    ny, nx = (mh + 1) // 2 - 1, (mw + 1) // 2 - 1                  # prepareInput's narrow: patch 1 = a_full[:, ny : ny + H1, nx : nx + W1]
    a = a_full[:, ny : ny + H1, nx : nx + W1]
    # rows 5..9: patch 1 is a copy of in2 shifted by (3, 4) -> cost 0 at one cell; rows 12..16: in2 carries that patch at (3, 4) AND (9, 11)
    a[:, 5:10] = b[:, 5 + 3 : 10 + 3, 4 : 4 + W1]
    b[:, 12 + 9 : 17 + 9, 11 : 11 + 200] = b[:, 12 + 3 : 17 + 3, 4 : 4 + 200]
    a[:, 12:17, :200] = b[:, 12 + 3 : 17 + 3, 4 : 4 + 200]
    # rows 20..24: the second copy perturbed so that its cost is a few 1e-7; rows 28..32: a few 1e-5
    for r0, eps in ((20, 1.5e-4), (28, 1.5e-3)):
        b[:, r0 + 9 : r0 + 5 + 9, 11 : 11 + 200] = b[:, r0 + 3 : r0 + 5 + 3, 4 : 4 + 200]
        b[0, r0 + 9 : r0 + 5 + 9, 11 : 11 + 200] += np.float32(eps)
        a[:, r0 : r0 + 5, :200] = b[:, r0 + 3 : r0 + 5 + 3, 4 : 4 + 200]
    model = dfe.getModel(geo, True, True)
    ctx = dfe.get_ctx(0)
    ta, tb = T(a_full, cuda), T(b, cuda)
    one = model.forwardFlow([ta, tb], None, one_call=True)
    assert ctx.last_kernel() == "feat_matching_flat_kernel+softmax"
    stg = model.forwardFlow([ta, tb], None, one_call=False)
    _same(one, stg, None)
    vol = orc.spatial_matching(np.ascontiguousarray(a), b, mh, mw).reshape(H1, W1, -1)
    srt = np.sort(vol, axis=2)
    gap = srt[..., 1] - srt[..., 0]
    assert (gap[12:17, :200] == 0).all() and (gap[5:10] > 1e-3).all()                       # exact ties / a lone minimum
    assert ((gap[20:25, :200] > 0) & (gap[20:25, :200] < 1e-6)).mean() > 0.9                # inside the band
    assert (gap[28:33, :200] > 1e-6).all() and (gap[28:33, :200] < 1e-4).mean() > 0.9       # just outside it
    idx = one["index"].cpu().numpy()
    assert (idx[5:10] == 3 * mw + 4 + 1).all()
    assert (idx[12:17, :200] == 3 * mw + 4 + 1).all()                                       # the first of two equal maxima
    assert (idx[28:33, :200] == 3 * mw + 4 + 1).all()
